/* prototype: exact fixed-point average linkage, sequential (lex ties) vs round-based mutual-NN (lex ties) */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
typedef struct { uint64_t key; uint32_t idx; } keyed;
static int cmp_keyed(const void *a, const void *b) {
    const keyed *x = (const keyed *)a, *y = (const keyed *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
static inline uint64_t absdiff(uint64_t a, uint64_t b) { return a > b ? a - b : b - a; }
#define NMAX 128
#define QCAP (1ull << 41)
#define QONE (1ull << 26)
static uint64_t q0[NMAX][NMAX], S[NMAX][NMAX];
/* round 6: what a nearest neighbour KEPT across rounds would save -- per size class: [0] rounds, [1] sum over
 * rounds of clusters alive (= iterations of the lane-per-cluster scan: every lane walks all of them), [2] sum of merges, [3] sum of
 * clusters that would have to look again (the merged ones, and those whose kept neighbour was in a merge), [4] rounds in which NO
 * cluster has to look again */
uint64_t nn_stats[5][8];
/* ... and what cross-group sums restricted to one threshold-graph component would save (item 1b) -- per class: [0] partitions with open
 * rows, [1] pairs of open rows (what the sums' pass evaluates, before the same-group pairs are taken out), [2] of them inside one component */
uint64_t comp_stats[5][8];
static int nn_cls;
typedef unsigned __int128 u128;
static void seq(uint32_t n, int mergeable, uint32_t *root)
{
    uint32_t size[NMAX];
    for (uint32_t i = 0; i < n; ++i) { root[i] = i; size[i] = 1; for (uint32_t j = 0; j < n; ++j) S[i][j] = q0[i][j]; }
    if (!mergeable) return;
    for (;;) {
        int ba = -1, bb = -1; uint64_t bs = 0, bn = 1;
        for (uint32_t a = 0; a < n; ++a) { if (root[a] != a) continue;
            for (uint32_t b = a + 1; b < n; ++b) { if (root[b] != b) continue;
                const uint64_t nn = (uint64_t)size[a] * size[b];
                if (ba < 0 || (u128)S[a][b] * bn < (u128)bs * nn) { bs = S[a][b]; bn = nn; ba = a; bb = b; } } }
        if (ba < 0 || bs > QONE * bn) break;
        for (uint32_t k = 0; k < n; ++k) { if (root[k] != k || (int)k == ba || (int)k == bb) continue; S[ba][k] += S[bb][k]; S[k][ba] = S[ba][k]; }
        size[ba] += size[bb];
        for (uint32_t k = 0; k < n; ++k) if (root[k] == (uint32_t)bb) root[k] = ba;
    }
}
static int rnn(uint32_t n, int mergeable, uint32_t *root, uint64_t *sum_alive)
{
    uint32_t size[NMAX], nn[NMAX], alive[NMAX], rep[NMAX], nn_prev[NMAX], touched[NMAX];
    int have_prev = 0;
    for (uint32_t i = 0; i < n; ++i) touched[i] = 0;
    for (uint32_t i = 0; i < n; ++i) { root[i] = i; size[i] = 1; alive[i] = 1; for (uint32_t j = 0; j < n; ++j) S[i][j] = q0[i][j]; }
    if (!mergeable) return 0;
    /* identical marks */
    for (uint32_t i = 0; i < n; ++i) { rep[i] = i; for (uint32_t j = 0; j < i; ++j) if (q0[i][j] == 0) { rep[i] = rep[j]; break; } }
    int rounds = 0, first = 1;
    for (;;) {
        if (!first) {
            ++rounds;
            uint32_t na = 0; for (uint32_t i = 0; i < n; ++i) na += alive[i]; *sum_alive += na;
            for (uint32_t a = 0; a < n; ++a) { nn[a] = NMAX; if (!alive[a]) continue;
                uint64_t bs = 0, bn = 1;
                for (uint32_t k = 0; k < n; ++k) { if (!alive[k] || k == a) continue;
                    if (nn[a] == NMAX || S[a][k] * bn < bs * size[k]) { bs = S[a][k]; bn = size[k]; nn[a] = k; } }
                if (nn[a] != NMAX && bs > QONE * bn * size[a]) nn[a] = NMAX;      /* beyond the threshold */
            }
            {   /* instrumentation: who would have had to look again this round */
                uint32_t dirty = 0;
                if (have_prev) for (uint32_t a = 0; a < n; ++a) if (alive[a] && (touched[a] || (nn_prev[a] != NMAX && (touched[nn_prev[a]] || !alive[nn_prev[a]])))) ++dirty;
                if (!have_prev) dirty = na;
                nn_stats[nn_cls][0]++; nn_stats[nn_cls][1] += na; nn_stats[nn_cls][3] += dirty; if (dirty == 0) nn_stats[nn_cls][4]++;
                for (uint32_t a = 0; a < n; ++a) { nn_prev[a] = nn[a]; touched[a] = 0; }
                have_prev = 1;
            }
            for (uint32_t a = 0; a < n; ++a) rep[a] = a;
            for (uint32_t a = 0; a < n; ++a) if (alive[a] && nn[a] != NMAX && nn[a] > a && nn[nn[a]] == a) { rep[nn[a]] = a; touched[a] = 1; touched[nn[a]] = 1; nn_stats[nn_cls][2]++; }
        }
        first = 0;
        int merged = 0;
        for (uint32_t b = 0; b < n; ++b) if (alive[b] && rep[b] != b) for (uint32_t k = 0; k < n; ++k) S[rep[b]][k] += S[b][k];
        for (uint32_t b = 0; b < n; ++b) if (alive[b] && rep[b] != b) { alive[b] = 0; size[rep[b]] += size[b]; merged = 1; for (uint32_t k = 0; k < n; ++k) if (root[k] == b) root[k] = rep[b]; }
        for (uint32_t r = 0; r < n; ++r) if (alive[r]) for (uint32_t b = 0; b < n; ++b) if (!alive[b] && rep[b] != b) { S[r][rep[b]] += S[r][b]; }
        for (uint32_t b = 0; b < n; ++b) if (!alive[b]) rep[b] = b;   /* consumed */
        if (!merged && rounds > 0) return rounds;
    }
}
/* st: per class [0] parts [1] mismatches [2] sum rounds [3] max rounds [4] sum alive [5] sum n */
int int_proto(uint32_t M, const uint16_t *contig, const uint8_t *type, const uint32_t *pos, const uint32_t *span,
              double T, uint32_t part_gap, uint32_t part_max, double normalizer, uint64_t *stats)
{
    keyed *ks = (keyed *)malloc(sizeof(keyed) * (M ? M : 1));
    for (uint32_t i = 0; i < M; ++i) { ks[i].key = ((uint64_t)contig[i] << 42) | ((uint64_t)type[i] << 34) | ((uint64_t)pos[i] + span[i] / 2); ks[i].idx = i; }
    qsort(ks, M, sizeof(keyed), cmp_keyed);
    const double invn = 1.0 / normalizer, scale = 67108864.0 / T;
    const int mergeable = T >= 0;
    uint32_t p0 = 0;
    while (p0 < M) {
        uint32_t p1 = p0 + 1;
        while (p1 < M && p1 - p0 < part_max) {
            const uint32_t a = ks[p1 - 1].idx, b = ks[p1].idx;
            if (contig[a] != contig[b] || type[a] != type[b]) break;
            if (((uint64_t)pos[b] + span[b] / 2) - ((uint64_t)pos[a] + span[a] / 2) > part_gap) break;
            ++p1;
        }
        const uint32_t n = p1 - p0;
        const int cls = n <= 8 ? 0 : n <= 16 ? 1 : n <= 32 ? 2 : n <= 64 ? 3 : 4;
        uint64_t *st = stats + 8 * cls;
        st[0]++;
        for (uint32_t i = 0; i < n; ++i) { const uint32_t a = ks[p0 + i].idx;
            for (uint32_t j = 0; j < n; ++j) { const uint32_t b = ks[p0 + j].idx;
                uint64_t m = absdiff(pos[a], pos[b]);
                const uint64_t m2 = absdiff((uint64_t)pos[a] + span[a], (uint64_t)pos[b] + span[b]);
                const uint64_t m3 = absdiff((uint64_t)pos[a] + span[a] / 2, (uint64_t)pos[b] + span[b] / 2);
                if (m2 < m) m = m2; if (m3 < m) m = m3;
                const uint32_t smax = span[a] > span[b] ? span[a] : span[b];
                const double inv = smax ? 1.0 / (double)smax : 0.0;
                const double d = (double)m * invn + (double)absdiff(span[a], span[b]) * inv;
                double t = rint(d * scale); if (!(t < (double)QCAP)) t = (double)QCAP; if (t < 1.0) t = 1.0;
                q0[i][j] = d == 0.0 ? 0 : (uint64_t)t;
            } }
        uint32_t r0[NMAX], r1[NMAX]; uint64_t sa = 0;
        nn_cls = cls;
        if (mergeable) {   /* components of {q <= threshold}; a component is settled when it is a clique */
            uint32_t comp[NMAX];
            for (uint32_t i = 0; i < n; ++i) comp[i] = i;
            for (int ch = 1; ch;) { ch = 0;
                for (uint32_t i = 0; i < n; ++i) for (uint32_t j = 0; j < n; ++j)
                    if (q0[i][j] <= QONE && comp[j] < comp[i]) { comp[i] = comp[j]; ch = 1; } }
            uint32_t open_n = 0, csize[NMAX] = {0}, cedges[NMAX] = {0};
            for (uint32_t i = 0; i < n; ++i) { csize[comp[i]]++; for (uint32_t j = i + 1; j < n; ++j) if (comp[i] == comp[j] && q0[i][j] <= QONE) cedges[comp[i]]++; }
            uint64_t same = 0;
            for (uint32_t c = 0; c < n; ++c) if (csize[c] && cedges[c] != csize[c] * (csize[c] - 1) / 2) { open_n += csize[c]; same += (uint64_t)csize[c] * (csize[c] - 1) / 2; }
            if (open_n) { comp_stats[cls][0]++; comp_stats[cls][1] += (uint64_t)open_n * (open_n - 1) / 2; comp_stats[cls][2] += same; }
        }
        seq(n, mergeable, r0);
        int rr = rnn(n, mergeable, r1, &sa);
        if (memcmp(r0, r1, 4 * n)) st[1]++;
        st[2] += rr; if ((uint64_t)rr > st[3]) st[3] = rr; st[4] += sa; st[5] += n;
        p0 = p1;
    }
    free(ks);
    return 0;
}
