/*
 * Design prototype (CPU, not product, not oracle): would a round-based "merge every certain mutual nearest-neighbour
 * pair at once" average linkage settle stage A0's partitions, and in how many rounds?  Compares its final clusters with
 * the sequential rule of oracle/cluster_oracle.c partition by partition and counts where it has to give up.
 *
 *   gcc -O2 -shared -fPIC -o /tmp/librnn_proto.so tools/rnn_proto.c
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
int g_dump = 0, g_dump_min = 0;

typedef struct { uint64_t key; uint32_t idx; } keyed;
static int cmp_keyed(const void *a, const void *b) {
    const keyed *x = (const keyed *)a, *y = (const keyed *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
static inline uint64_t absdiff(uint64_t a, uint64_t b) { return a > b ? a - b : b - a; }

#define NMAX 128
static double d0[NMAX][NMAX];      /* pair distances */
static double S[NMAX][NMAX];       /* sums of pair distances between clusters */
static double D[NMAX][NMAX];
int g_mode = 1, g_mode2 = 3, g_levels = 2;

/* sequential rule -> root[] (smallest member) */
static void upgma(uint32_t n, double T, uint32_t *root)
{
    uint32_t size[NMAX];
    for (uint32_t i = 0; i < n; ++i) { root[i] = i; size[i] = 1; for (uint32_t j = 0; j < n; ++j) D[i][j] = d0[i][j]; }
    for (;;) {
        double best = 0; int ba = -1, bb = -1;
        for (uint32_t a = 0; a < n; ++a) {
            if (root[a] != a) continue;
            for (uint32_t b = a + 1; b < n; ++b) {
                if (root[b] != b) continue;
                if (ba < 0 || D[a][b] < best) { best = D[a][b]; ba = (int)a; bb = (int)b; }
            }
        }
        if (ba < 0 || !(best <= T)) break;
        const double na = size[ba], nb = size[bb];
        for (uint32_t k = 0; k < n; ++k) {
            if (root[k] != k || (int)k == ba || (int)k == bb) continue;
            const double v = (na * D[ba][k] + nb * D[bb][k]) / (na + nb);
            D[ba][k] = v; D[k][ba] = v;
        }
        size[ba] += size[bb];
        for (uint32_t k = 0; k < n; ++k) if (root[k] == (uint32_t)bb) root[k] = (uint32_t)ba;
    }
}

/* round-based; returns rounds (> 0) or -rounds when it gets stuck; sum_alive: work measure.
 * mode bit 0: tie rule (<= 3 equidistant singletons: smallest index), bit 1: safe groups */
static int rnn(uint32_t n, double T, uint32_t *root, int mode, uint64_t *sum_alive)
{
    const double delta = (mode & 4) ? 1e-4 : 1e-9;
    const int atoms = (mode & 12) != 0;
    uint32_t size[NMAX], nn[NMAX], alive[NMAX], single[NMAX];
    double rinv[NMAX], m1s[NMAX];
    int cert[NMAX];
    for (uint32_t i = 0; i < n; ++i) { root[i] = i; size[i] = 1; alive[i] = 1; single[i] = 1; rinv[i] = 1.0; for (uint32_t j = 0; j < n; ++j) S[i][j] = d0[i][j]; }
    if (T < 0) return 1;
    {
        uint32_t rep[NMAX], cnt[NMAX];
        for (uint32_t i = 0; i < n; ++i) { rep[i] = i; for (uint32_t j = 0; j < i; ++j) if (d0[i][j] == 0.0) { rep[i] = rep[j]; break; } }
        for (uint32_t i = 0; i < n; ++i) cnt[i] = 0;
        for (uint32_t i = 0; i < n; ++i) cnt[rep[i]]++;
        for (uint32_t i = 0; i < n; ++i) {
            if (rep[i] != i) { alive[i] = 0; root[i] = rep[i]; continue; }
            size[i] = cnt[i]; rinv[i] = 1.0 / cnt[i]; single[i] = cnt[i] == 1;
        }
        for (uint32_t i = 0; i < n; ++i) if (alive[i]) for (uint32_t j = 0; j < n; ++j) if (alive[j]) S[i][j] = d0[i][j] * (double)cnt[i] * (double)cnt[j];
    }
    if (atoms) {
        /* binary32 distances; atoms = clique components of the T/2 graph, then of the T/4 graph (guard band 1e-5) */
        for (uint32_t i = 0; i < n; ++i) { alive[i] = 1; root[i] = i; size[i] = 1; rinv[i] = 1.0; single[i] = 1; for (uint32_t j = 0; j < n; ++j) S[i][j] = (mode & 4) ? (double)(float)d0[i][j] : d0[i][j]; }
        uint32_t atom[NMAX];
        for (uint32_t i = 0; i < n; ++i) atom[i] = NMAX;
        for (int lvl = 1; lvl <= g_levels; ++lvl) {
            const double r = T / (1 << lvl), rlo = r * (1 - 1e-5), rhi = r * (1 + 1e-5);
            int pass[NMAX];
            for (uint32_t i = 0; i < n; ++i) {
                uint32_t f = i; int amb = 0;
                for (uint32_t j = 0; j < n; ++j) { const int hi = S[i][j] <= rhi || i == j, lo = S[i][j] <= rlo || i == j; if (hi != lo) amb = 1; if (hi && j < f) f = j; }
                int same = !amb;
                for (uint32_t j = 0; j < n && same; ++j) if ((S[i][j] <= rhi || i == j) != (S[f][j] <= rhi || f == j)) same = 0;
                pass[i] = same;
            }
            for (uint32_t i = 0; i < n; ++i) {
                if (atom[i] != NMAX) continue;
                int clean = pass[i]; uint32_t f = i;
                for (uint32_t j = 0; j < n; ++j) if (S[i][j] <= rhi || i == j) { if (!pass[j] || atom[j] != NMAX && atom[j] != NMAX + 1) clean = clean && pass[j]; if (j < f) f = j; }
                if (clean) atom[i] = NMAX + 1 + f + 1000 * lvl;      /* provisional: resolved below */
            }
            for (uint32_t i = 0; i < n; ++i) if (atom[i] >= NMAX + 1 + 1000 * lvl && atom[i] < NMAX + 1 + 1000 * (lvl + 1)) atom[i] = atom[i] - (NMAX + 1 + 1000 * lvl);
        }
        /* contract */
        for (uint32_t i = 0; i < n; ++i) if (atom[i] != NMAX && atom[i] != i) { const uint32_t a = atom[i]; for (uint32_t k = 0; k < n; ++k) S[a][k] += S[i][k]; }
        for (uint32_t i = 0; i < n; ++i) if (atom[i] != NMAX && atom[i] != i) alive[i] = 0;
        for (uint32_t r = 0; r < n; ++r) if (alive[r]) for (uint32_t i = 0; i < n; ++i) if (!alive[i]) S[r][atom[i]] += S[r][i];
        for (uint32_t i = 0; i < n; ++i) if (!alive[i]) { size[atom[i]]++; root[i] = atom[i]; single[atom[i]] = 0; }
        for (uint32_t i = 0; i < n; ++i) if (alive[i]) rinv[i] = 1.0 / size[i];
    }
    int rounds = 0;
    for (;;) {
        ++rounds;
        uint32_t na = 0;
        for (uint32_t i = 0; i < n; ++i) na += alive[i];
        *sum_alive += na;
        int open = 0;
        for (uint32_t a = 0; a < n; ++a) {
            if (!alive[a]) continue;
            double m1 = INFINITY; uint32_t k1 = NMAX;
            for (uint32_t k = 0; k < n; ++k) {
                if (!alive[k] || k == a) continue;
                const double v = S[a][k] * rinv[k];
                if (v < m1) { m1 = v; k1 = k; }
            }
            nn[a] = k1; cert[a] = 0; m1s[a] = m1;
            if (k1 == NMAX) continue;
            const double avg = m1 * rinv[a];
            if (avg <= T * (1 + delta)) open = 1;
            if (!(avg <= T * (1 - delta))) continue;
            /* the others: equal (bitwise, singletons) or clearly farther? */
            int near = 0, eq = 0, eq_single = single[a] && single[k1];
            for (uint32_t k = 0; k < n; ++k) {
                if (!alive[k] || k == a || k == k1) continue;
                const double v = S[a][k] * rinv[k];
                if (v > m1 * (1 + delta)) continue;
                ++near;
                if (v == m1) { ++eq; eq_single = eq_single && single[k]; }
            }
            if (near == 0 || m1 == 0.0) cert[a] = 1;
            else if ((mode & 1) && near == eq && eq <= 2 && eq_single) cert[a] = 1;     /* k1 is the smallest index by scan order */
        }
        uint32_t grp[NMAX];                         /* group id = its smallest member; NMAX = none */
        for (uint32_t i = 0; i < n; ++i) grp[i] = NMAX;
        int merged = 0;
        for (uint32_t a = 0; a < n; ++a) {
            if (!alive[a] || !cert[a]) continue;
            const uint32_t b = nn[a];
            if (b > a && cert[b] && nn[b] == a) { grp[a] = a; grp[b] = a; }
        }
        if (mode & 2) {
            for (uint32_t a = 0; a < n; ++a) {
                if (!alive[a] || grp[a] != NMAX || nn[a] == NMAX) continue;
                const double avg = m1s[a] * rinv[a];
                if (!(avg <= T * (1 - delta))) continue;
                /* Q = a + everything within the radius of its nearest (+ slack) */
                uint32_t Q[NMAX], nq = 0; int ok = 1;
                const double r = avg * (1 + 4 * delta);
                for (uint32_t k = 0; k < n; ++k) if (alive[k] && (k == a || S[a][k] * rinv[k] * rinv[a] <= r)) { Q[nq++] = k; if (grp[k] != NMAX) ok = 0; }
                if (!ok || nq < 2 || Q[0] != a) continue;
                double maxint = 0, minext = INFINITY;
                for (uint32_t x = 0; x < nq; ++x) for (uint32_t k = 0; k < n; ++k) {
                    if (!alive[k] || k == Q[x]) continue;
                    const double e = S[Q[x]][k] * rinv[k] * rinv[Q[x]];
                    int in = 0;
                    for (uint32_t y = 0; y < nq; ++y) if (Q[y] == k) in = 1;
                    if (in) { if (e > maxint) maxint = e; } else if (e < minext) minext = e;
                }
                if (maxint <= T * (1 - delta) && minext > maxint * (1 + delta)) for (uint32_t x = 0; x < nq; ++x) grp[Q[x]] = a;
            }
        }
        /* rows, then columns */
        for (uint32_t b = 0; b < n; ++b) if (alive[b] && grp[b] != NMAX && grp[b] != b) for (uint32_t k = 0; k < n; ++k) S[grp[b]][k] += S[b][k];
        for (uint32_t b = 0; b < n; ++b) if (alive[b] && grp[b] != NMAX && grp[b] != b) alive[b] = 0;
        for (uint32_t r = 0; r < n; ++r) if (alive[r]) for (uint32_t b = 0; b < n; ++b) if (!alive[b] && grp[b] != NMAX && grp[b] != b) { S[r][grp[b]] += S[r][b]; }
        for (uint32_t b = 0; b < n; ++b) if (grp[b] != NMAX && grp[b] != b) {
            const uint32_t a = grp[b];
            size[a] += size[b]; rinv[a] = 1.0 / size[a]; single[a] = 0;
            for (uint32_t k = 0; k < n; ++k) if (root[k] == b) root[k] = a;
            grp[b] = NMAX;                           /* (dead rows keep no group for the next round) */
            merged = 1;
        }
        if (!merged && open && g_dump > 0 && mode == g_mode2) {
            --g_dump;
            if (n <= (uint32_t)g_dump_min) { ++g_dump; goto nodump; }
            fprintf(stderr, "STUCK n=%u round %d\n", n, rounds);
            for (uint32_t a = 0; a < n; ++a) if (alive[a] && nn[a] != NMAX && m1s[a] * rinv[a] <= T * (1 + delta)) {
                fprintf(stderr, "  a=%u size=%u nn=%u avg=%.17g cert=%d | near:", a, size[a], nn[a], m1s[a] * rinv[a], cert[a]);
                for (uint32_t k = 0; k < n; ++k) if (alive[k] && k != a && S[a][k] * rinv[k] <= m1s[a] * (1 + 1e-6)) fprintf(stderr, " %u(sz%u,%.17g)", k, size[k], S[a][k] * rinv[k] * rinv[a]);
                fprintf(stderr, "\n");
            }
        }
        nodump:
        if (!merged) return open ? -rounds : rounds;
    }
}

/* stats layout: per size class c (0: <=8, 1: <=16, 2: <=32, 3: <=64, 4: >64), 12 words:
 * [0] partitions [1] box-settled [2] level-0 cliques (not box) [3] rnn needed [4] rnn ok [5] rnn stuck [6] MISMATCH
 * [7] sum of rounds [8] max rounds [9] sum over rounds of alive clusters [10] sum n of rnn partitions [11] stuck with tie rule too */
int rnn_proto(uint32_t M, const uint16_t *contig, const uint8_t *type, const uint32_t *pos, const uint32_t *span,
              double T, uint32_t part_gap, uint32_t part_max, double normalizer, uint64_t *stats)
{
    keyed *ks = (keyed *)malloc(sizeof(keyed) * (M ? M : 1));
    for (uint32_t i = 0; i < M; ++i) {
        ks[i].key = ((uint64_t)contig[i] << 42) | ((uint64_t)type[i] << 34) | ((uint64_t)pos[i] + span[i] / 2);
        ks[i].idx = i;
    }
    qsort(ks, M, sizeof(keyed), cmp_keyed);
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
        uint64_t *st = stats + 12 * cls;
        st[0]++;
        uint64_t lo[4] = {~0ull, ~0ull, ~0ull, ~0ull}, hi[4] = {0, 0, 0, 0};
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t a = ks[p0 + i].idx;
            const uint64_t q[4] = {pos[a], (uint64_t)pos[a] + span[a], (uint64_t)pos[a] + span[a] / 2, span[a]};
            for (int c = 0; c < 4; ++c) { if (q[c] < lo[c]) lo[c] = q[c]; if (q[c] > hi[c]) hi[c] = q[c]; }
            for (uint32_t j = 0; j < n; ++j) {
                const uint32_t b = ks[p0 + j].idx;
                uint64_t m = absdiff(pos[a], pos[b]);
                const uint64_t m2 = absdiff((uint64_t)pos[a] + span[a], (uint64_t)pos[b] + span[b]);
                const uint64_t m3 = absdiff((uint64_t)pos[a] + span[a] / 2, (uint64_t)pos[b] + span[b] / 2);
                if (m2 < m) m = m2;
                if (m3 < m) m = m3;
                const uint32_t smax = span[a] > span[b] ? span[a] : span[b];
                d0[i][j] = (double)m / normalizer + (smax ? (double)absdiff(span[a], span[b]) / (double)smax : 0.0);
            }
        }
        uint64_t r = hi[0] - lo[0];
        if (hi[1] - lo[1] < r) r = hi[1] - lo[1];
        if (hi[2] - lo[2] < r) r = hi[2] - lo[2];
        const double U = (double)r / normalizer + (hi[3] ? 1.0 - (double)lo[3] / (double)hi[3] : 0.0);
        if (n < 2 || U <= T * (1 - 1e-5)) { st[1]++; p0 = p1; continue; }
        /* level-0 cliques (guard band 1e-5) */
        int amb = 0, clq = 1;
        for (uint32_t i = 0; i < n && clq; ++i) {
            uint32_t f = i;
            for (uint32_t j = 0; j < n; ++j) {
                const int e_hi = d0[i][j] <= T * (1 + 1e-5), e_lo = d0[i][j] <= T * (1 - 1e-5);
                if (e_hi != e_lo) amb = 1;
                if (e_hi && j < f) f = j;
            }
            for (uint32_t j = 0; j < n; ++j) if ((d0[i][j] <= T * (1 + 1e-5)) != (d0[f][j] <= T * (1 + 1e-5))) { clq = 0; break; }
        }
        if (!amb && clq) { st[2]++; p0 = p1; continue; }
        st[3]++;
        st[10] += n;
        uint32_t r0[NMAX], r1[NMAX];
        upgma(n, T, r0);
        uint64_t sa = 0;
        int rr = rnn(n, T, r1, g_mode, &sa);
        if (rr < 0) {
            st[5]++;
            uint64_t sb = 0;
            int rr2 = rnn(n, T, r1, g_mode2, &sb);
            if (rr2 < 0) st[11]++;
            else if (memcmp(r0, r1, sizeof(uint32_t) * n)) st[6]++;
        } else {
            st[4]++;
            if (memcmp(r0, r1, sizeof(uint32_t) * n)) st[6]++;
            st[7] += (uint64_t)rr;
            if ((uint64_t)rr > st[8]) st[8] = (uint64_t)rr;
            st[9] += sa;
        }
        p0 = p1;
    }
    free(ks);
    return 0;
}
