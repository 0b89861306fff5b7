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
 * average linkage cut at cluster_max_distance); tests/test_cluster_oracle.py cross-checks the
 * agglomeration against scipy.cluster.hierarchy.linkage(method='average') + fcluster(criterion='distance').
 *
 * Rule:
 *  1. order marks by (contig, type, centre = pos + span/2), stable in the input order;
 *  2. a new partition starts where contig or type changes, where the centre gap exceeds part_gap, or when
 *     the current partition already holds part_max marks (part_max <= 128);
 *  3. pair distance, binary64, exactly these operations (no contraction):
 *         d(i,j) = m * (1 / normalizer) + |span_i - span_j| * (1 / max(span_i, span_j))
 *     with m = min(|pos_i-pos_j|, |end_i-end_j|, |centre_i-centre_j|) (second term 0 when both spans are 0),
 *     then FIXED POINT relative to the threshold: q(i,j) = 0 if d == 0, else rint(d * (2^26 / max_dist)) held
 *     to [1, 2^41]: the threshold itself is 2^26 and a step is 1.5e-8 of it; one pair at the cap of 2^41 = 2^15
 *     thresholds already keeps the mean of at most 64 x 64 member pairs above the threshold, so the cap never changes
 *     a decision (and every sum stays below 2^55);
 *  4. average linkage in EXACT arithmetic: the distance of two clusters is the exact mean of q over their
 *     member pairs, sum / (n_a n_b), compared as rationals; repeatedly merge the closest pair of clusters
 *     (ties: smallest first index, then smallest second; the merged cluster keeps the smaller index) while
 *     its mean <= 2^26; nothing merges when max_dist < 0 or is not a number;
 *  5. clusters of a partition are emitted by smallest member, members in sorted order;
 *     candidate pos = floor(mean pos), span = floor(mean span), support = number of marks.
 *
 * Why fixed point (round 3): sums of integers do not depend on the order of the merges, so ANY evaluation
 * order that merges provably-first pairs -- all mutual nearest neighbours of a round at once, whole cliques of the
 * threshold graph -- arrives at exactly these clusters (average linkage is reducible; DESIGN.md section 9 has
 * the argument), ties included.  The binary64 Lance-Williams recurrence of rounds 1-2 (what scipy evaluates) gives
 * the same clusters except where two cluster distances agree to ~1e-8 of the threshold: there its result depended on
 * the rounding of one particular merge order, which only a serial replay of that order could reproduce.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef struct { uint64_t key; uint32_t idx; } keyed;

static int cmp_keyed(const void *a, const void *b) {
    const keyed *x = (const keyed *)a, *y = (const keyed *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);          /* stable */
}

static inline uint64_t absdiff(uint64_t a, uint64_t b) { return a > b ? a - b : b - a; }

#define QONE (1ull << 26)                /* the threshold in fixed point */
#define QONE_D 67108864.0
#define QCAP_D 2199023255552.0           /* 2^41 */
/* rule 3: 0 for identical marks, else the nearest integer of d * scale (ties to even) held to [1, 2^41];
 * d * scale may be +inf (max_dist = 0 or denormal) or 0 (max_dist = +inf) */
static inline uint64_t quantise(double d, double scale) {
    if (d == 0.0) return 0;
    double t = rint(d * scale);
    if (!(t < QCAP_D)) t = QCAP_D;
    if (t < 1.0) t = 1.0;
    return (uint64_t)t;
}

int duet_oracle_cluster(uint32_t M, const uint16_t *contig, const uint8_t *type, const uint32_t *pos,
                        const uint32_t *span, double max_dist, uint32_t part_gap, uint32_t part_max,
                        double normalizer,
                        uint32_t *order, uint32_t *cand_off, uint32_t *n_cands_out,
                        uint16_t *cand_contig, uint8_t *cand_type, uint32_t *cand_pos, uint32_t *cand_span)
{
    if (part_max < 1 || part_max > 128) return -1;
    keyed *ks = (keyed *)malloc(sizeof(keyed) * (M ? M : 1));
    uint64_t *q = (uint64_t *)malloc(sizeof(uint64_t) * part_max * part_max);
    uint32_t *root = (uint32_t *)malloc(sizeof(uint32_t) * part_max);
    uint32_t *size = (uint32_t *)malloc(sizeof(uint32_t) * part_max);
    for (uint32_t i = 0; i < M; ++i) {
        const uint64_t centre = (uint64_t)pos[i] + span[i] / 2;     /* < 2^33 */
        ks[i].key = ((uint64_t)contig[i] << 42) | ((uint64_t)(type[i] & 0xFF) << 34) | centre;
        ks[i].idx = i;
    }
    qsort(ks, M, sizeof(keyed), cmp_keyed);
    const double invn = 1.0 / normalizer, scale = QONE_D / max_dist;
    const int mergeable = max_dist >= 0;                            /* false for negative and NaN */

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
                const double inv = smax ? 1.0 / (double)smax : 0.0;
                const double dp = (double)m * invn, ds = (double)absdiff(span[a], span[b]) * inv;
                q[i * n + j] = quantise(dp + ds, scale);
            }
        }
        /* average linkage on the active roots, exact means */
        while (mergeable) {
            int ba = -1, bb = -1;
            uint64_t bs = 0, bn = 1;
            for (uint32_t a = 0; a < n; ++a) {
                if (root[a] != a) continue;
                for (uint32_t b = a + 1; b < n; ++b) {
                    if (root[b] != b) continue;
                    const uint64_t nn = (uint64_t)size[a] * size[b];
                    if (ba < 0 || (unsigned __int128)q[a * n + b] * bn < (unsigned __int128)bs * nn) {
                        bs = q[a * n + b]; bn = nn; ba = (int)a; bb = (int)b;
                    }
                }
            }
            if (ba < 0 || bs > QONE * bn) break;
            for (uint32_t k = 0; k < n; ++k) {
                if (root[k] != k || (int)k == ba || (int)k == bb) continue;
                const uint64_t v = q[ba * n + k] + q[bb * n + k];             /* sums of member-pair distances */
                q[ba * n + k] = v;
                q[k * n + ba] = v;
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
    free(ks); free(q); free(root); free(size);
    return 0;
}
