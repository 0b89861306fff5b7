/*
 * TEST INFRASTRUCTURE ONLY -- scalar C statement of the A0 stage: span-position clustering of SV
 * marks into candidates (what `--cluster_max_distance` controls).
 *
 * Parity status: UNPINNED against the reference.  The reference delegates this stage to the external
 * `svim alignment` binary (src/duet/sv_calling.py:13-15; svim = 1.4.2 pinned in README.md:31,42); its
 * source is not under /root/reference and it is not installed, and the reference holds no test or golden
 * output for it.  This file is therefore the NORMATIVE statement of this repository's own deterministic
 * rule, modelled on the published SVIM 1.4.2 algorithm (SVIM_clustering.py: partition by a 1000 bp gap of
 * the sorted centres, at most 100 signatures per partition, span-position distance with normaliser 900,
 * scipy average linkage cut at cluster_max_distance); tests/test_cluster_oracle.py cross-checks the
 * agglomeration against scipy.cluster.hierarchy.linkage(method='average') + fcluster(criterion='distance').
 *
 * Rule:
 *  1. order marks by (contig, type, centre = pos + span/2), stable in the input order;
 *  2. a new partition starts where contig or type changes, where the centre gap exceeds part_gap, or when
 *     the current partition already holds part_max marks;
 *  3. d(i,j) = min(|pos_i-pos_j|, |end_i-end_j|, |centre_i-centre_j|) / normalizer
 *              + |span_i-span_j| / max(span_i, span_j)      (second term 0 when both spans are 0), binary64;
 *  4. repeatedly merge the closest pair of clusters (ties: smallest first index, then smallest second)
 *     while that distance <= max_dist; average linkage via the Lance-Williams update
 *     d(a,k) = (n_a d(a,k) + n_b d(b,k)) / (n_a + n_b), the merged cluster keeping the smaller index;
 *  5. clusters of a partition are emitted by smallest member, members in sorted order;
 *     candidate pos = floor(mean pos), span = floor(mean span), support = number of marks.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t key; uint32_t idx; } keyed;

static int cmp_keyed(const void *a, const void *b) {
    const keyed *x = (const keyed *)a, *y = (const keyed *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);          /* stable */
}

static inline uint64_t absdiff(uint64_t a, uint64_t b) { return a > b ? a - b : b - a; }

int duet_oracle_cluster(uint32_t M, const uint16_t *contig, const uint8_t *type, const uint32_t *pos,
                        const uint32_t *span, double max_dist, uint32_t part_gap, uint32_t part_max,
                        double normalizer,
                        uint32_t *order, uint32_t *cand_off, uint32_t *n_cands_out,
                        uint16_t *cand_contig, uint8_t *cand_type, uint32_t *cand_pos, uint32_t *cand_span)
{
    if (part_max < 1 || part_max > 1024) return -1;
    keyed *ks = (keyed *)malloc(sizeof(keyed) * (M ? M : 1));
    double *d = (double *)malloc(sizeof(double) * part_max * part_max);
    uint32_t *root = (uint32_t *)malloc(sizeof(uint32_t) * part_max);
    uint32_t *size = (uint32_t *)malloc(sizeof(uint32_t) * part_max);
    for (uint32_t i = 0; i < M; ++i) {
        const uint64_t centre = (uint64_t)pos[i] + span[i] / 2;     /* < 2^33 */
        ks[i].key = ((uint64_t)contig[i] << 42) | ((uint64_t)(type[i] & 0xFF) << 34) | centre;
        ks[i].idx = i;
    }
    qsort(ks, M, sizeof(keyed), cmp_keyed);

    uint32_t n_cands = 0, n_out = 0;
    cand_off[0] = 0;
    uint32_t p0 = 0;
    while (p0 < M) {
        /* partition [p0, p1) */
        uint32_t p1 = p0 + 1;
        while (p1 < M && p1 - p0 < part_max) {
            const uint32_t a = ks[p1 - 1].idx, b = ks[p1].idx;
            if (contig[a] != contig[b] || type[a] != type[b]) break;
            const uint64_t ca = (uint64_t)pos[a] + span[a] / 2, cb = (uint64_t)pos[b] + span[b] / 2;
            if (cb - ca > part_gap) break;
            ++p1;
        }
        const uint32_t n = p1 - p0;
        for (uint32_t i = 0; i < n; ++i) {
            root[i] = i;
            size[i] = 1;
            const uint32_t a = ks[p0 + i].idx;
            const uint64_t sa = pos[a], ea = (uint64_t)pos[a] + span[a], ca = (uint64_t)pos[a] + span[a] / 2;
            for (uint32_t j = 0; j < n; ++j) {
                const uint32_t b = ks[p0 + j].idx;
                const uint64_t sb = pos[b], eb = (uint64_t)pos[b] + span[b], cb = (uint64_t)pos[b] + span[b] / 2;
                uint64_t m = absdiff(sa, sb);
                const uint64_t m2 = absdiff(ea, eb), m3 = absdiff(ca, cb);
                if (m2 < m) m = m2;
                if (m3 < m) m = m3;
                const uint32_t smax = span[a] > span[b] ? span[a] : span[b];
                const double dp = (double)m / normalizer;
                const double ds = smax ? (double)absdiff(span[a], span[b]) / (double)smax : 0.0;
                d[i * n + j] = dp + ds;
            }
        }
        /* UPGMA on the active roots */
        for (;;) {
            double best = 0;
            int ba = -1, bb = -1;
            for (uint32_t a = 0; a < n; ++a) {
                if (root[a] != a) continue;
                for (uint32_t b = a + 1; b < n; ++b) {
                    if (root[b] != b) continue;
                    if (ba < 0 || d[a * n + b] < best) { best = d[a * n + b]; ba = (int)a; bb = (int)b; }
                }
            }
            if (ba < 0 || !(best <= max_dist)) break;
            const double na = (double)size[ba], nb = (double)size[bb];
            for (uint32_t k = 0; k < n; ++k) {
                if (root[k] != k || (int)k == ba || (int)k == bb) continue;
                const double v = (na * d[ba * n + k] + nb * d[bb * n + k]) / (na + nb);
                d[ba * n + k] = v;
                d[k * n + ba] = v;
            }
            size[ba] += size[bb];
            for (uint32_t k = 0; k < n; ++k) if (root[k] == (uint32_t)bb) root[k] = (uint32_t)ba;
        }
        for (uint32_t r = 0; r < n; ++r) {
            if (root[r] != r) continue;
            uint64_t sp = 0, ss = 0;
            uint32_t cnt = 0;
            for (uint32_t j = r; j < n; ++j) {
                if (root[j] != r) continue;
                const uint32_t a = ks[p0 + j].idx;
                order[n_out++] = a;
                sp += pos[a];
                ss += span[a];
                ++cnt;
            }
            const uint32_t a0 = ks[p0 + r].idx;
            cand_contig[n_cands] = contig[a0];
            cand_type[n_cands] = type[a0];
            cand_pos[n_cands] = (uint32_t)(sp / cnt);
            cand_span[n_cands] = (uint32_t)(ss / cnt);
            cand_off[++n_cands] = n_out;
        }
        p0 = p1;
    }
    *n_cands_out = n_cands;
    free(ks); free(d); free(root); free(size);
    return 0;
}
