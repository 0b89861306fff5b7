/*
 * TEST INFRASTRUCTURE ONLY -- scalar C restatement (oracle) of Duet's step E/F on the SoA layout of
 * include/duet_ef.h.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product path never does.
 *
 * One candidate at a time, one mark at a time, in list order -- the same order of evaluation as the
 * reference's Python (src/duet/sv_phasing_fn.py, cited per block), so every order-dependent rule
 * (first qualifying mark, first-seen phase set on ties, last voter's PS) falls out of the loop
 * order instead of being re-derived.  All floating point is IEEE binary64 with the operations
 * written exactly as upstream (compile with -ffp-contract=off; there is nothing to contract anyway).
 *
 * Parity status: PINNED through tests/test_c_oracle.py, which runs this code on the golden
 * fixtures (tests/golden/) and compares with the Python oracle's per-candidate trace, itself pinned
 * byte-for-byte to the imported reference.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PC_MAX 8100u                    /* sv_phasing_fn.py:76,88,201 */
#define ABSENT 0xFFFFFFFFu              /* mark whose read name is not in the contig's tag table */

static inline unsigned tag_hap(uint64_t t) { return (unsigned)(t >> 62); }
static inline uint32_t tag_pc(uint64_t t) { return (uint32_t)((t >> 32) & 0x3FFFFFFFu); }
static inline uint32_t tag_ps(uint64_t t) { return (uint32_t)t; }

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* lower_bound on ascending u32 */
static uint32_t lower_bound(const uint32_t *a, uint32_t n, int64_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        if ((int64_t)a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

static int member(const uint32_t *a, uint32_t n, uint32_t key) {
    uint32_t i = lower_bound(a, n, (int64_t)key);
    return i < n && a[i] == key;
}

/* sv_phasing_fn.py:107-111 -- nearest seed PS, ties to the larger one */
static uint32_t nearest(const uint32_t *a, uint32_t n, uint32_t pos) {
    uint32_t i = lower_bound(a, n, (int64_t)pos);
    uint32_t lo = i > 0 ? i - 1 : 0;
    uint32_t hi = i < n - 1 ? i : n - 1;
    int64_t dl = (int64_t)pos - (int64_t)a[lo]; if (dl < 0) dl = -dl;
    int64_t dh = (int64_t)pos - (int64_t)a[hi]; if (dh < 0) dh = -dh;
    return dl < dh ? a[lo] : a[hi];
}

/*
 * Returns 0, or -5 if a candidate that reaches the decision has svread + refread == 0 (the reference
 * raises ZeroDivisionError there, sv_phasing_fn.py:123).
 * out_pred[c] in 0..3; out_ps[c] = the PS predict_hp returns when it is called for c, else 0.
 */
int duet_oracle_ef(uint32_t K, uint32_t C,
                   const uint64_t *read_tag, const uint32_t *cand_ctg_off,
                   const uint32_t *cand_pos, const uint32_t *cand_svlen, const uint32_t *cand_svread,
                   const uint32_t *cand_refread, const uint8_t *cand_gt_ok,
                   const uint32_t *cand_off, const uint32_t *mark_read,
                   uint32_t svlen_thres, uint32_t suppread_thres,
                   uint8_t *out_pred, uint32_t *out_ps)
{
    uint8_t *cls = (uint8_t *)malloc(C ? C : 1);        /* 0,1,2 or 255 = filtered out */
    uint32_t *seeds = (uint32_t *)malloc(sizeof(uint32_t) * (C ? C : 1));
    int rc = 0;
    memset(out_pred, 0, C);
    memset(out_ps, 0, sizeof(uint32_t) * (size_t)C);

    for (uint32_t k = 0; k < K; ++k) {
        uint32_t c0 = cand_ctg_off[k], c1 = cand_ctg_off[k + 1];
        uint32_t n_seed = 0;
        /* E2 filter (:189-190), E3 class over ALL tagged marks (:191-194), E4 seeds (:195-203) */
        for (uint32_t c = c0; c < c1; ++c) {
            cls[c] = 255;
            if (!(cand_svlen[c] >= svlen_thres && cand_svread[c] >= suppread_thres && cand_gt_ok[c])) continue;
            int n_ps = 0; uint32_t first_ps = 0;
            for (uint32_t m = cand_off[c]; m < cand_off[c + 1]; ++m) {
                if (mark_read[m] == ABSENT) continue;
                uint32_t ps = tag_ps(read_tag[mark_read[m]]);
                if (n_ps == 0) { n_ps = 1; first_ps = ps; }
                else if (ps != first_ps) n_ps = 2;
            }
            cls[c] = (uint8_t)n_ps;
            if (n_ps == 1) {
                for (uint32_t m = cand_off[c]; m < cand_off[c + 1]; ++m) {
                    if (mark_read[m] == ABSENT) continue;
                    uint64_t t = read_tag[mark_read[m]];
                    if (tag_pc(t) <= PC_MAX) { seeds[c0 + n_seed++] = tag_ps(t); break; }
                }
            }
        }
        if (n_seed == 0) continue;                      /* :209-210 -- contig dropped */
        qsort(seeds + c0, n_seed, sizeof(uint32_t), cmp_u32);
        uint32_t n_one = 0;
        for (uint32_t i = 0; i < n_seed; ++i)
            if (i == 0 || seeds[c0 + i] != seeds[c0 + n_one - 1]) seeds[c0 + n_one++] = seeds[c0 + i];
        const uint32_t *one = seeds + c0;

        for (uint32_t c = c0; c < c1; ++c) {
            if (cls[c] == 255) continue;
            uint32_t b = cand_off[c], e = cand_off[c + 1];
            uint64_t hap1 = 0, hap2 = 0, hap0 = 0, allhap = 0, t1 = 0, t2 = 0;
            uint32_t ps = 0;
            if (cls[c] == 1) {                          /* :74-84 */
                for (uint32_t m = b; m < e; ++m) {
                    if (mark_read[m] == ABSENT) continue;
                    uint64_t t = read_tag[mark_read[m]];
                    if (tag_pc(t) > PC_MAX) continue;
                    ps = tag_ps(t);
                    if (tag_hap(t) == 1) { hap1++; t1 += tag_pc(t); }
                    else if (tag_hap(t) == 2) { hap2++; t2 += tag_pc(t); }
                }
                allhap = hap1 + hap2;
            } else if (cls[c] == 2) {                   /* :85-105 */
                uint64_t best = 0;
                for (uint32_t m = b; m < e; ++m) {
                    if (mark_read[m] == ABSENT) continue;
                    if (tag_pc(read_tag[mark_read[m]]) <= PC_MAX) allhap++;
                }
                for (uint32_t m = b; m < e; ++m) {      /* groups in first-seen order */
                    if (mark_read[m] == ABSENT) continue;
                    uint64_t t = read_tag[mark_read[m]];
                    if (tag_pc(t) > PC_MAX || !member(one, n_one, tag_ps(t))) continue;
                    int seen = 0;
                    for (uint32_t j = b; j < m && !seen; ++j) {
                        if (mark_read[j] == ABSENT) continue;
                        uint64_t u = read_tag[mark_read[j]];
                        if (tag_pc(u) <= PC_MAX && tag_ps(u) == tag_ps(t)) seen = 1;
                    }
                    if (seen) continue;
                    uint64_t n = 0, n1 = 0, n2 = 0, s1 = 0, s2 = 0;
                    for (uint32_t j = m; j < e; ++j) {
                        if (mark_read[j] == ABSENT) continue;
                        uint64_t u = read_tag[mark_read[j]];
                        if (tag_pc(u) > PC_MAX || tag_ps(u) != tag_ps(t)) continue;
                        n++;
                        if (tag_hap(u) == 1) { n1++; s1 += tag_pc(u); }
                        else if (tag_hap(u) == 2) { n2++; s2 += tag_pc(u); }
                    }
                    if (n > best) {                     /* strict: first-seen wins ties */
                        best = n; hap1 = n1; hap2 = n2; t1 = s1; t2 = s2; ps = tag_ps(t);
                        hap0 = allhap - hap1 - hap2;
                    }
                }
            }
            if (cls[c] == 0 || (hap1 == 0 && hap2 == 0)) ps = nearest(one, n_one, cand_pos[c]);   /* :106-111 */

            uint64_t svread = cand_svread[c], refread = cand_refread[c];
            if (svread + refread == 0) { rc = -5; continue; }
            double deg = (double)(e - b);
            double hapread_ratio = (double)allhap / deg;                               /* :112 */
            double a1 = hap1 > 0 ? (double)t1 / (double)hap1 : 0.0;                    /* :113-116 */
            double a2 = hap2 > 0 ? (double)t2 / (double)hap2 : 0.0;
            double sv_ratio = (double)svread / (double)(svread + refread);             /* :123 */
            uint64_t lo = t1 < t2 ? t1 : t2, hi = t1 < t2 ? t2 : t1;
            double totsc_ratio = lo > 0 ? (double)hi / (double)lo : 0.0;               /* :124-125 */
            uint64_t onehap = lo == 0 ? hi : 0;                                        /* :126-127 */
            double diff = a2 - a1; if (diff < 0) diff = -diff;                         /* :132 */
            int pred = 0;
            if (cls[c] == 0) {                                                         /* :145-147 */
                if (sv_ratio == 1.0 && svread >= 4) pred = 3;
            } else if (cls[c] == 2) {                                                  /* :148-155 */
                if (sv_ratio >= 0.72) {
                    if (diff <= 1369.50) { if (svread >= 3) pred = 3; }
                    else { if (hap0 >= 6) pred = 3; }
                }
            } else {                                                                   /* :156-182 */
                int gate = (hapread_ratio <= 0.75 && diff <= 2400.0) || hapread_ratio > 0.75;
                if (onehap != 0) {
                    if (sv_ratio <= 0.24) pred = 0;
                    else if (sv_ratio <= 0.9) { if (gate) pred = a1 > 0 ? 1 : 2; }
                    else { if (gate) pred = 3; }
                } else {
                    if (sv_ratio <= 0.3) pred = 0;
                    else if (sv_ratio <= 0.45) pred = refread > 10 ? 0 : (t1 > t2 ? 1 : 2);
                    else if (sv_ratio <= 0.75) pred = totsc_ratio <= 9.72 ? 3 : (t1 > t2 ? 1 : 2);
                    else pred = 3;
                }
            }
            out_pred[c] = (uint8_t)pred;
            out_ps[c] = ps;
        }
    }
    free(cls);
    free(seeds);
    return rc;
}
