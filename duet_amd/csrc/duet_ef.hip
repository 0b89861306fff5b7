// duet_ef.hip -- gfx950 (MI355X) kernels and C ABI for Duet's step E/F.
//
// Pipeline (three launches on one stream, no host round trip):
//
//   ef_classify   one 256-thread workgroup per 256 consecutive candidates.  The workgroup streams
//                 its slice of mark_read with 16-byte loads, gathers the 8-byte read tag of every
//                 mark (the join of sv_phasing_fn.py:46-48) into LDS, then each thread walks ITS
//                 candidate's marks in list order out of LDS: filter (:189-190), PS-class (:191-194),
//                 seed PS (:199-203), class-0/1 vote (:74-84) and decision (:142-183).  Seeds go
//                 into a per-contig open-addressing hash set in HBM (atomicCAS), deduplicated
//                 against the neighbouring candidate first.
//   ef_seed_sort  one workgroup per contig: collects the contig's distinct seeds, bitonic-sorts
//                 them in LDS (global memory for > 16K seeds) -> ascending `oneps` array (:107),
//                 and wipes the hash slots it consumed (the set is self-cleaning between runs).
//   ef_finalize   per candidate: contig drop (:209-210), nearest-PS (:106-111), and the class-2
//                 grouped vote (:85-105) + decision (:148-155) for the few multi-PS candidates.
//
// Everything order-dependent upstream (first qualifying mark, first-seen PS wins ties, last voter's
// PS) is reproduced by walking marks in list order; nothing relies on atomics ordering -- the only
// atomics build a SET, whose content does not depend on arrival order.
//
// Floating point: IEEE binary64, operations exactly as upstream, built with -ffp-contract=off.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "duet_ef.h"

#pragma clang fp contract(off)

namespace {

constexpr uint32_t kEmpty = 0xFFFFFFFFu;          // empty hash slot / "no seed"
constexpr uint64_t kUntagged = ~0ull;             // LDS tag word of an absent mark
constexpr uint32_t kPcMax = DUET_PC_MAX;

// provisional code written by ef_classify into out_pred; values < 4 are already final
constexpr uint8_t kNeedNearest = 4;               // ps must become nearest(oneps, pos)
constexpr uint8_t kClass2 = 8;                    // multi-PS candidate with a group summary (out_ps = slot)
constexpr uint8_t kClass2Slow = 16;               // multi-PS candidate without one: ef_finalize re-gathers
constexpr uint8_t kDivZero = 32;                  // kept candidate with svread + refread == 0

constexpr int kCandPerBlock = 256;
constexpr int kChunk = 4096;                      // marks staged in LDS per pass (32 KiB of tags)
constexpr int kSortThreads = 1024;
constexpr uint32_t kSortLds = 15360;              // seeds sorted in LDS (60 KiB)
constexpr uint32_t kOneLds = 4096;                // seed array staged in LDS by ef_finalize (16 KiB)
constexpr uint32_t kC2Quota = 32;                 // group-summary slots per classify block
constexpr int kC2Groups = 4;                      // voter groups kept per summary
constexpr int kC2Words = 2 + 6 * kC2Groups;       // allhap, ng, then {ps, n, n1, n2, t1, t2} per group

struct Vote {
    uint32_t hap1, hap2, hap0, allhap;
    uint64_t t1, t2;
};

__device__ __forceinline__ uint32_t tag_ps(uint64_t t) { return (uint32_t)t; }
__device__ __forceinline__ uint32_t tag_pc(uint64_t t) { return (uint32_t)(t >> 32) & 0x3FFFFFFFu; }
__device__ __forceinline__ uint32_t tag_hap(uint64_t t) { return (uint32_t)(t >> 62); }

// predict_hp, sv_phasing_fn.py:142-183, on the features of :112-139.  cls in {0,1,2}.
__device__ int decide(int cls, const Vote &v, uint32_t deg, uint32_t svread, uint32_t refread)
{
    const double hapread_ratio = (double)v.allhap / (double)deg;                      // :112
    const double a1 = v.hap1 > 0 ? (double)v.t1 / (double)v.hap1 : 0.0;               // :113-114
    const double a2 = v.hap2 > 0 ? (double)v.t2 / (double)v.hap2 : 0.0;               // :115-116
    const double sv_ratio = (double)svread / (double)((uint64_t)svread + (uint64_t)refread);   // :123
    const uint64_t lo = v.t1 < v.t2 ? v.t1 : v.t2;
    const uint64_t hi = v.t1 < v.t2 ? v.t2 : v.t1;
    const double totsc_ratio = lo > 0 ? (double)hi / (double)lo : 0.0;                // :124-125
    const uint64_t onehap = lo == 0 ? hi : 0;                                         // :126-127
    const double diff = fabs(a2 - a1);                                                // :132
    int pred = 0;
    if (cls == 0) {                                                                   // :145-147
        if (sv_ratio == 1.0 && svread >= 4) pred = 3;
    } else if (cls == 2) {                                                            // :148-155
        if (sv_ratio >= 0.72) {
            if (diff <= 1369.50) { if (svread >= 3) pred = 3; }
            else { if (v.hap0 >= 6) pred = 3; }
        }
    } else {                                                                          // :156-182
        const bool gate = (hapread_ratio <= 0.75 && diff <= 2400.0) || hapread_ratio > 0.75;
        if (onehap != 0) {
            if (sv_ratio <= 0.24) pred = 0;
            else if (sv_ratio <= 0.9) { if (gate) pred = a1 > 0 ? 1 : 2; }
            else { if (gate) pred = 3; }
        } else {
            if (sv_ratio <= 0.3) pred = 0;
            else if (sv_ratio <= 0.45) pred = refread > 10 ? 0 : (v.t1 > v.t2 ? 1 : 2);
            else if (sv_ratio <= 0.75) pred = totsc_ratio <= 9.72 ? 3 : (v.t1 > v.t2 ? 1 : 2);
            else pred = 3;
        }
    }
    return pred;
}

// index k of the contig owning candidate c: last k with ctg_off[k] <= c
__device__ __forceinline__ uint32_t find_contig(const uint32_t *__restrict__ ctg_off, uint32_t K, uint32_t c)
{
    uint32_t lo = 0, hi = K;          // invariant: ctg_off[lo] <= c < ctg_off[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ctg_off[mid] <= c) lo = mid; else hi = mid;
    }
    return lo;
}

struct Params {
    uint32_t K, C, M;
    const uint64_t *read_tag;
    const uint32_t *cand_pos, *cand_svlen, *cand_svread, *cand_refread;
    const uint8_t *cand_gt_ok;
    const uint32_t *cand_off, *mark_read;
    uint32_t svlen_thres, suppread_thres;
    // workspace (plan-time constants)
    const uint32_t *ctg_off;          // [K+1] device copy of cand_ctg_off
    const uint8_t *ctg_start;         // [C]   1 where a candidate is the first of its contig
    const uint32_t *blk_ctg;          // [B]   contig of candidate 256*b
    // workspace (per run)
    uint32_t *blk_cnt;                // [B]   seed entries emitted by classify block b
    uint64_t *seed_ent;               // [B*256] (candidate << 32 | seed PS), compacted per block
    uint32_t *onebuf;                 // [C]   per contig (at ctg_off[k]): ascending distinct seed PS
    uint32_t *tmpbuf;                 // [C]   scratch of the out-of-LDS seed sort
    uint32_t *n_one;                  // [K]   length of contig k's seed array
    uint32_t *c2rec;                  // [B*kC2Quota*kC2Words] group summaries of multi-PS candidates
    uint32_t *status;                 // [0] = div-zero flag
    uint8_t *out_pred;
    uint32_t *out_ps;
    uint32_t dbg;                     // diagnostic ablation bits (0 in production)
};

// ---------------------------------------------------------------------------------------------
// kernel 1: join + filter + PS-class + seeds + one-PS vote/decision + multi-PS group summaries
// ---------------------------------------------------------------------------------------------

template <bool VEC>
__global__ __launch_bounds__(kCandPerBlock) void ef_classify(const Params p)
{
    __shared__ uint64_t s_tag[kChunk];
    __shared__ uint32_t s_off[kCandPerBlock + 1];
    __shared__ uint32_t s_seed[kCandPerBlock];
    __shared__ uint32_t s_wcnt[kCandPerBlock / 64], s_wlast[kCandPerBlock / 64], s_wclean[kCandPerBlock / 64];
    __shared__ uint32_t s_c2n;

    const uint32_t tid = threadIdx.x;
    const uint32_t c0 = blockIdx.x * kCandPerBlock;
    const uint32_t nc = min((uint32_t)kCandPerBlock, p.C - c0);
    for (uint32_t i = tid; i <= nc; i += kCandPerBlock) s_off[i] = p.cand_off[c0 + i];
    if (tid == 0) s_c2n = 0;

    // candidate scalars, coalesced
    const bool live = tid < nc;
    const uint32_t c = c0 + tid;
    uint32_t svlen = 0, svread = 0, refread = 0, gt_ok = 0, is_start = 0;
    if (live) {
        svlen = p.cand_svlen[c];
        svread = p.cand_svread[c];
        refread = p.cand_refread[c];
        gt_ok = p.cand_gt_ok[c];
        is_start = p.ctg_start[c];
    }
    const bool kept = live && svlen >= p.svlen_thres && svread >= p.suppread_thres && gt_ok != 0;   // :189-190
    const bool divzero = kept && ((uint64_t)svread + (uint64_t)refread == 0);
    const bool active = kept && !divzero;
    __syncthreads();
    const uint32_t m_begin = s_off[0], m_end = s_off[nc];
    const uint32_t my_b = live ? s_off[tid] : 0, my_e = live ? s_off[tid + 1] : 0;

    // per-candidate running state over marks in list order
    uint32_t n_ps = 0, first_ps = 0;          // distinct PS among tagged marks: 0, 1, 2(=many)
    uint32_t seed = kEmpty;                   // PS of first voter (:199-203)
    uint32_t last_ps = 0;                     // PS of last voter (:77)
    uint32_t h1 = 0, h2 = 0, t1 = 0, t2 = 0;  // per-chunk partial sums (<= 4096 * 8100)
    uint64_t T1 = 0, T2 = 0;
    uint8_t code = 0;
    uint32_t ps_out = 0;
    bool c2_done = false;

    const uint32_t base = VEC ? (m_begin & ~3u) : m_begin;
    for (uint32_t cs = base; cs < m_end; cs += kChunk) {
        // ---- stage: LDS[i] = tag of mark cs+i -----------------------------------------------
        if (VEC) {
            uint4 r[4];
            uint64_t t[16];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const uint32_t m = cs + 4u * (tid + it * kCandPerBlock);
                r[it] = make_uint4(kEmpty, kEmpty, kEmpty, kEmpty);
                if (m < m_end) {
                    if (m + 3 < p.M) {
                        r[it] = *reinterpret_cast<const uint4 *>(p.mark_read + m);
                    } else {
                        r[it].x = p.mark_read[m];
                        if (m + 1 < p.M) r[it].y = p.mark_read[m + 1];
                        if (m + 2 < p.M) r[it].z = p.mark_read[m + 2];
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                t[4 * it + 0] = r[it].x == kEmpty ? kUntagged : p.read_tag[r[it].x];
                t[4 * it + 1] = r[it].y == kEmpty ? kUntagged : p.read_tag[r[it].y];
                t[4 * it + 2] = r[it].z == kEmpty ? kUntagged : p.read_tag[r[it].z];
                t[4 * it + 3] = r[it].w == kEmpty ? kUntagged : p.read_tag[r[it].w];
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const uint32_t i = 4u * (tid + it * kCandPerBlock);
                if (cs + i < m_end) {
                    s_tag[i + 0] = t[4 * it + 0];
                    s_tag[i + 1] = t[4 * it + 1];
                    s_tag[i + 2] = t[4 * it + 2];
                    s_tag[i + 3] = t[4 * it + 3];
                }
            }
        } else {
            for (uint32_t i = tid; i < kChunk && cs + i < m_end; i += kCandPerBlock) {
                const uint32_t r = p.mark_read[cs + i];
                s_tag[i] = r == kEmpty ? kUntagged : p.read_tag[r];
            }
        }
        __syncthreads();
        // ---- consume: each thread walks its candidate's part of this chunk (branch-light) ------
        uint32_t lo = max(my_b, cs);
        const uint32_t hi = min(my_e, cs + (uint32_t)kChunk);
        if (!active || (p.dbg & 4)) lo = hi;
        for (uint32_t m = lo; m < hi; ++m) {
            const uint64_t tag = s_tag[m - cs];
            const uint32_t ps = (uint32_t)tag, w = (uint32_t)(tag >> 32);
            const bool tagged = w != 0xFFFFFFFFu;           // an absent mark is all ones
            const uint32_t pc = w & 0x3FFFFFFFu, hap = w >> 30;
            const bool voter = pc <= kPcMax;                // absent marks have pc = 2^30-1: never voters
            const bool fresh = tagged && n_ps == 0;
            first_ps = fresh ? ps : first_ps;
            n_ps = tagged ? (ps != first_ps ? 2u : max(n_ps, 1u)) : n_ps;
            seed = (voter && seed == kEmpty) ? ps : seed;
            last_ps = voter ? ps : last_ps;
            const bool v1 = voter && hap == 1, v2 = voter && hap == 2;
            h1 += v1; t1 += v1 ? pc : 0u;
            h2 += v2; t2 += v2 ? pc : 0u;
        }
        T1 += t1; T2 += t2; t1 = 0; t2 = 0;
        // ---- multi-PS candidate that lies entirely in this chunk: summarise its voter groups -------
        // (first-seen order, sv_phasing_fn.py:85-98) so that ef_finalize needs no second gather
        if (active && n_ps == 2 && !c2_done && my_b >= cs && my_e <= cs + (uint32_t)kChunk) {
            c2_done = true;
            uint32_t g_ps[kC2Groups], g_n[kC2Groups], g_n1[kC2Groups], g_n2[kC2Groups], g_t1[kC2Groups], g_t2[kC2Groups];
#pragma unroll
            for (int k = 0; k < kC2Groups; ++k) { g_ps[k] = 0; g_n[k] = 0; g_n1[k] = 0; g_n2[k] = 0; g_t1[k] = 0; g_t2[k] = 0; }
            uint32_t ng = 0, allhap = 0;
            bool overflow = false;
            for (uint32_t m = my_b; m < my_e; ++m) {
                const uint64_t tag = s_tag[m - cs];
                const uint32_t ps = (uint32_t)tag, w = (uint32_t)(tag >> 32);
                const uint32_t pc = w & 0x3FFFFFFFu, hap = w >> 30;
                if (pc > kPcMax) continue;
                ++allhap;
                int idx = -1;
#pragma unroll
                for (int k = 0; k < kC2Groups; ++k) if ((uint32_t)k < ng && g_ps[k] == ps) idx = k;
                if (idx < 0) {
                    if (ng == kC2Groups) { overflow = true; continue; }
                    idx = (int)ng++;
                }
#pragma unroll
                for (int k = 0; k < kC2Groups; ++k) {
                    const bool hit = k == idx;
                    g_ps[k] = hit ? ps : g_ps[k];
                    g_n[k] += hit;
                    g_n1[k] += hit && hap == 1;
                    g_n2[k] += hit && hap == 2;
                    g_t1[k] += (hit && hap == 1) ? pc : 0u;
                    g_t2[k] += (hit && hap == 2) ? pc : 0u;
                }
            }
            const uint32_t rank = overflow ? kC2Quota : atomicAdd(&s_c2n, 1u);
            if (rank < kC2Quota) {
                const uint32_t slot = blockIdx.x * kC2Quota + rank;
                uint32_t *rec = p.c2rec + (size_t)slot * kC2Words;
                rec[0] = allhap;
                rec[1] = ng;
#pragma unroll
                for (int k = 0; k < kC2Groups; ++k) {
                    if ((uint32_t)k < ng) {
                        rec[2 + 6 * k + 0] = g_ps[k]; rec[2 + 6 * k + 1] = g_n[k];
                        rec[2 + 6 * k + 2] = g_n1[k]; rec[2 + 6 * k + 3] = g_n2[k];
                        rec[2 + 6 * k + 4] = g_t1[k]; rec[2 + 6 * k + 5] = g_t2[k];
                    }
                }
                code = kClass2;
                ps_out = slot;
            }
        }
        __syncthreads();
    }

    // ---- per-candidate result ------------------------------------------------------------------
    bool want_seed = false;
    if (live) {
        if (divzero) {
            code = kDivZero;
        } else if (active) {
            want_seed = (n_ps == 1) && seed != kEmpty;                                             // :198-203
            if (n_ps == 2) {
                if (code != kClass2) code = kClass2Slow;        // no summary: ef_finalize re-gathers
            } else {
                Vote v;
                v.hap1 = h1; v.hap2 = h2; v.hap0 = 0;
                v.allhap = h1 + h2;
                v.t1 = T1; v.t2 = T2;
                code = (p.dbg & 2) ? 0 : (uint8_t)decide((int)n_ps, v, my_e - my_b, svread, refread);
                ps_out = last_ps;
                if (n_ps == 0 || (h1 == 0 && h2 == 0)) code |= kNeedNearest;                       // :106
            }
        }
        p.out_pred[c] = code;
        p.out_ps[c] = ps_out;
    }

    // ---- seeds of this block, compacted.  A seed equal to the seed of the previous seed-bearing
    // candidate is dropped, unless a contig starts in between (seed sets are per contig).  Only the
    // set matters downstream, so dropping duplicates early just shortens ef_seed_sort's input.
    if (p.dbg & 1) want_seed = false;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const unsigned long long wmask = __ballot(want_seed);
    const unsigned long long smask = __ballot(is_start != 0);
    const unsigned long long upto = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);    // lanes 0..lane
    s_seed[tid] = want_seed ? seed : kEmpty;
    __syncthreads();
    if (lane == 0) {
        if (wmask) {
            const uint32_t li = 63u - (uint32_t)__clzll(wmask);
            s_wlast[wave] = s_seed[wave * 64 + li];
            s_wclean[wave] = (li == 63 || (smask >> (li + 1)) == 0) ? 1u : 0u;   // no contig start after it
        } else {
            s_wlast[wave] = kEmpty;
            s_wclean[wave] = 0;
        }
    }
    __syncthreads();
    bool keep = want_seed;
    if (want_seed) {
        const unsigned long long lower = wmask & (upto >> 1);                   // seed lanes below this one
        if (lower) {
            const uint32_t pl = 63u - (uint32_t)__clzll(lower);
            const unsigned long long between = smask & upto & ~((1ull << (pl + 1)) - 1ull);   // starts in (pl, lane]
            keep = !(between == 0 && s_seed[wave * 64 + pl] == seed);
        } else if (wave > 0) {
            keep = !((smask & upto) == 0 && s_wclean[wave - 1] && s_wlast[wave - 1] == seed);
        }
    }
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) s_wcnt[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kCandPerBlock / 64; ++w) {
        before += w < wave ? s_wcnt[w] : 0u;
        total += s_wcnt[w];
    }
    if (keep) {
        const uint32_t at = before + (uint32_t)__popcll(mask & (upto >> 1));
        p.seed_ent[(size_t)blockIdx.x * kCandPerBlock + at] = ((uint64_t)c << 32) | seed;
    }
    if (tid == 0) p.blk_cnt[blockIdx.x] = total;
}

// ---------------------------------------------------------------------------------------------
// kernel 2: per contig, distinct seeds -> ascending array (np.sort(list(oneps_set)), :107)
// ---------------------------------------------------------------------------------------------

// All-ascending bitonic network ("flip" form): every compare-exchange puts the smaller key at the
// lower index, so indices >= n behave as +inf padding without being stored.
__device__ void bitonic_sort(uint32_t *a, uint32_t n, uint32_t tid, uint32_t nthreads)
{
    uint32_t N = 1;
    while (N < n) N <<= 1;
    for (uint32_t k = 2; k <= N; k <<= 1) {
        const uint32_t half = k >> 1;
        for (uint32_t i = tid; i < (N >> 1); i += nthreads) {
            const uint32_t blk = i / half, off = i % half;
            const uint32_t x = blk * k + off, y = blk * k + (k - 1 - off);
            if (y < n) {
                const uint32_t ax = a[x], ay = a[y];
                if (ax > ay) { a[x] = ay; a[y] = ax; }
            }
        }
        __syncthreads();
        for (uint32_t j = half >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < (N >> 1); i += nthreads) {
                const uint32_t x = (i / j) * (j << 1) + (i % j), y = x + j;
                if (y < n) {
                    const uint32_t ax = a[x], ay = a[y];
                    if (ax > ay) { a[x] = ay; a[y] = ax; }
                }
            }
            __syncthreads();
        }
    }
}

// exclusive prefix sum of one value per thread over the block; *total gets the block sum
__device__ uint32_t block_exscan(uint32_t v, uint32_t tid, uint32_t *s_part /* [nthreads/64 + 1] */, uint32_t nthreads,
                                 uint32_t *total)
{
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if ((int)lane >= d) x += y;
    }
    if (lane == 63) s_part[wave] = x;
    __syncthreads();
    uint32_t before = 0, tot = 0;
    for (uint32_t w = 0; w < nthreads / 64; ++w) {
        const uint32_t pw = s_part[w];
        before += w < wave ? pw : 0u;
        tot += pw;
    }
    __syncthreads();
    *total = tot;
    return before + x - v;
}

// sorted keys src[0..n) -> distinct keys dst[...]; returns the number of distinct keys.
// Each thread owns a contiguous slice so the output order is the input order.
__device__ uint32_t unique_copy(const uint32_t *src, uint32_t n, uint32_t *dst, uint32_t tid, uint32_t nthreads,
                                uint32_t *s_part)
{
    const uint32_t per = (n + nthreads - 1) / nthreads;
    const uint32_t b = min(n, tid * per), e = min(n, b + per);
    uint32_t cnt = 0;
    for (uint32_t i = b; i < e; ++i) cnt += (i == 0 || src[i] != src[i - 1]);
    uint32_t total;
    uint32_t at = block_exscan(cnt, tid, s_part, nthreads, &total);
    for (uint32_t i = b; i < e; ++i)
        if (i == 0 || src[i] != src[i - 1]) dst[at++] = src[i];
    return total;
}

__global__ __launch_bounds__(kSortThreads) void ef_seed_sort(const Params p)
{
    __shared__ uint32_t s_key[kSortLds];
    __shared__ uint32_t s_part[kSortThreads / 64 + 1];
    __shared__ uint32_t s_n;
    const uint32_t k = blockIdx.x, tid = threadIdx.x;
    const uint32_t c_lo = p.ctg_off[k], c_hi = p.ctg_off[k + 1];
    if (c_lo == c_hi) {
        if (tid == 0) p.n_one[k] = 0;
        return;
    }
    if (tid == 0) s_n = 0;
    __syncthreads();
    // seed entries of the classify blocks that overlap this contig; keep those whose candidate is ours
    const uint32_t b_lo = c_lo / kCandPerBlock, b_hi = (c_hi - 1) / kCandPerBlock;
    uint32_t *glist = p.onebuf + c_lo;                       // capacity c_hi - c_lo >= number of entries
    for (uint32_t b = b_lo + tid; b <= b_hi; b += kSortThreads) {
        const uint32_t cnt = p.blk_cnt[b];
        const uint64_t *ent = p.seed_ent + (size_t)b * kCandPerBlock;
        // the last entry of the previous block, if it belongs to this contig, absorbs an equal first entry
        uint32_t prev_ps = kEmpty;
        if (b > b_lo && cnt) {
            const uint32_t pc = p.blk_cnt[b - 1];
            if (pc) {
                const uint64_t pe = p.seed_ent[(size_t)(b - 1) * kCandPerBlock + pc - 1];
                if ((uint32_t)(pe >> 32) >= c_lo) prev_ps = (uint32_t)pe;
            }
        }
        for (uint32_t j = 0; j < cnt; ++j) {
            const uint64_t e = ent[j];
            const uint32_t c = (uint32_t)(e >> 32);
            if (c < c_lo || c >= c_hi) continue;
            if (j == 0 && (uint32_t)e == prev_ps) continue;     // both candidates lie in [c_lo, c_hi)
            const uint32_t at = atomicAdd(&s_n, 1u);
            if (at < kSortLds) s_key[at] = (uint32_t)e;
            glist[at] = (uint32_t)e;                          // also kept in HBM for the large path
        }
    }
    __syncthreads();
    const uint32_t n = s_n;
    uint32_t n_one;
    if (n == 0) {
        n_one = 0;
    } else if (n <= kSortLds) {
        bitonic_sort(s_key, n, tid, kSortThreads);
        n_one = unique_copy(s_key, n, glist, tid, kSortThreads, s_part);
    } else {
        // more seeds than LDS holds (unsorted input with very many phase sets): sort in HBM
        __threadfence_block();
        bitonic_sort(glist, n, tid, kSortThreads);
        uint32_t *tmp = p.tmpbuf + c_lo;
        n_one = unique_copy(glist, n, tmp, tid, kSortThreads, s_part);
        __syncthreads();
        for (uint32_t i = tid; i < n_one; i += kSortThreads) glist[i] = tmp[i];
    }
    if (tid == 0) p.n_one[k] = n_one;
}

// ---------------------------------------------------------------------------------------------
// kernel 3: contig drop, nearest PS, multi-PS vote + decision
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *a, uint32_t n, uint32_t key)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ bool is_member(const uint32_t *a, uint32_t n, uint32_t key)
{
    const uint32_t at = lower_bound_u32(a, n, key);
    return at < n && a[at] == key;
}

// sv_phasing_fn.py:107-111 -- ties go to the larger seed
__device__ __forceinline__ uint32_t nearest_ps(const uint32_t *a, uint32_t n, uint32_t pos)
{
    const uint32_t i = lower_bound_u32(a, n, pos);
    const uint32_t lo = i > 0 ? i - 1 : 0;
    const uint32_t hi = i < n - 1 ? i : n - 1;
    const int64_t dl = llabs((int64_t)pos - (int64_t)a[lo]);
    const int64_t dh = llabs((int64_t)pos - (int64_t)a[hi]);
    return dl < dh ? a[lo] : a[hi];
}

__device__ __forceinline__ uint64_t fetch_tag(const Params &p, uint32_t m)
{
    const uint32_t r = p.mark_read[m];
    return r == kEmpty ? kUntagged : p.read_tag[r];
}

// multi-PS vote straight from the marks (:85-105): only for candidates without a group summary
__device__ void class2_from_marks(const Params &p, uint32_t c, const uint32_t *one, uint32_t n_one, Vote &v, uint32_t &ps)
{
    const uint32_t b = p.cand_off[c], e = p.cand_off[c + 1];
    uint32_t best = 0;
    for (uint32_t m = b; m < e; ++m) {
        const uint64_t t = fetch_tag(p, m);
        if (t == kUntagged || tag_pc(t) > kPcMax) continue;
        ++v.allhap;
    }
    uint32_t done_ps = kEmpty;                                 // last group evaluated (cheap duplicate skip)
    for (uint32_t m = b; m < e; ++m) {
        const uint64_t t = fetch_tag(p, m);
        if (t == kUntagged || tag_pc(t) > kPcMax) continue;
        const uint32_t g = tag_ps(t);
        if (g == done_ps || (g == ps && best)) continue;
        if (!is_member(one, n_one, g)) continue;               // :91
        // size and sums of g's group over the whole list; a later occurrence of an already
        // evaluated group reproduces the same n and cannot beat it (strict '>', :101)
        uint32_t n = 0, n1 = 0, n2 = 0;
        uint64_t s1 = 0, s2 = 0;
        for (uint32_t j = b; j < e; ++j) {
            const uint64_t u = fetch_tag(p, j);
            if (u == kUntagged || tag_pc(u) > kPcMax || tag_ps(u) != g) continue;
            ++n;
            const uint32_t hap = tag_hap(u);
            if (hap == 1) { ++n1; s1 += tag_pc(u); }
            else if (hap == 2) { ++n2; s2 += tag_pc(u); }
        }
        done_ps = g;
        if (n > best) {
            best = n; ps = g;
            v.hap1 = n1; v.hap2 = n2; v.t1 = s1; v.t2 = s2;
            v.hap0 = v.allhap - n1 - n2;                       // only with a winner (:105)
        }
    }
}

__global__ __launch_bounds__(256) void ef_finalize(const Params p)
{
    __shared__ uint32_t s_one[kOneLds];
    __shared__ uint32_t s_meta[4];                             // k0, n_one[k0] or ~0 (not LDS mode), any_empty, ctg_off[k0]
    const uint32_t tid = threadIdx.x;
    const uint32_t c0 = blockIdx.x * 256u;
    const uint32_t c = c0 + tid;
    const bool live = c < p.C;
    const uint8_t code = live ? p.out_pred[c] : 0;
    const uint32_t ps_in = live ? p.out_ps[c] : 0;
    if (tid == 0) {
        const uint32_t last = min(c0 + 255u, p.C - 1);
        const uint32_t k0 = p.blk_ctg[blockIdx.x];
        uint32_t k1 = k0;
        while (last >= p.ctg_off[k1 + 1]) ++k1;
        uint32_t any = 0;
        for (uint32_t k = k0; k <= k1; ++k) any |= (p.n_one[k] == 0);
        const uint32_t n0 = p.n_one[k0];
        s_meta[0] = k0;
        s_meta[1] = (k0 == k1 && n0 > 0 && n0 <= kOneLds) ? n0 : kEmpty;
        s_meta[2] = any;
        s_meta[3] = p.ctg_off[k0];
    }
    __syncthreads();
    const uint32_t k0 = s_meta[0], n_lds = s_meta[1], any_empty = s_meta[2];
    const bool lds_mode = n_lds != kEmpty;
    if (lds_mode) {
        const uint32_t *src = p.onebuf + s_meta[3];
        for (uint32_t i = tid; i < n_lds; i += 256u) s_one[i] = src[i];
        __syncthreads();
    }
    if (!live) return;
    if (code < 4 && !any_empty) return;                        // already final
    const uint32_t *one;
    uint32_t n_one;
    if (lds_mode) {
        one = s_one;
        n_one = n_lds;
    } else {
        uint32_t k = k0;
        while (c >= p.ctg_off[k + 1]) ++k;
        n_one = p.n_one[k];
        one = p.onebuf + p.ctg_off[k];
    }
    if (n_one == 0) {                                          // :209-210
        if (code != 0) p.out_pred[c] = 0;
        if (ps_in != 0) p.out_ps[c] = 0;
        return;
    }
    if (code < 4) return;
    if (code & kDivZero) {                                     // :123 would raise
        atomicOr(&p.status[0], 1u);
        p.out_pred[c] = 0;
        p.out_ps[c] = 0;
        return;
    }
    if (code & (kClass2 | kClass2Slow)) {                      // :85-105, :148-155
        Vote v = {0, 0, 0, 0, 0, 0};
        uint32_t ps = 0;
        if (code & kClass2) {
            const uint32_t *rec = p.c2rec + (size_t)ps_in * kC2Words;
            uint32_t w[kC2Words];
#pragma unroll
            for (int i = 0; i < kC2Words; ++i) w[i] = rec[i];  // slots beyond ng hold stale words, never used
            v.allhap = w[0];
            const uint32_t ng = w[1];
            uint32_t best = 0;
#pragma unroll
            for (int k = 0; k < kC2Groups; ++k) {
                if ((uint32_t)k < ng && w[2 + 6 * k + 1] > best && is_member(one, n_one, w[2 + 6 * k])) {
                    best = w[2 + 6 * k + 1];
                    ps = w[2 + 6 * k];
                    v.hap1 = w[2 + 6 * k + 2]; v.hap2 = w[2 + 6 * k + 3];
                    v.t1 = w[2 + 6 * k + 4]; v.t2 = w[2 + 6 * k + 5];
                    v.hap0 = v.allhap - v.hap1 - v.hap2;       // only with a winner (:105)
                }
            }
        } else {
            class2_from_marks(p, c, one, n_one, v, ps);
        }
        if (v.hap1 == 0 && v.hap2 == 0) ps = nearest_ps(one, n_one, p.cand_pos[c]);       // :106
        const uint32_t deg = p.cand_off[c + 1] - p.cand_off[c];
        p.out_pred[c] = (uint8_t)decide(2, v, deg, p.cand_svread[c], p.cand_refread[c]);
        p.out_ps[c] = ps;
        return;
    }
    // kNeedNearest
    p.out_pred[c] = code & 3;
    p.out_ps[c] = nearest_ps(one, n_one, p.cand_pos[c]);
}

// plan time: ctg_start[c] = 1 for the first candidate of every non-empty contig
__global__ void plan_mark_starts(const uint32_t *ctg_off, uint32_t K, uint8_t *ctg_start)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K && ctg_off[k] < ctg_off[k + 1]) ctg_start[ctg_off[k]] = 1;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------

thread_local std::string g_last_error;

struct DevBuf {
    void *ptr = nullptr;
    size_t cap = 0;
};

}  // namespace

struct duet_ctx {
    int device = 0;
    std::string err;
    bool profiling = false;
    uint32_t dbg = 0;
    hipStream_t own_stream = nullptr;
    // plan (workspace keyed by the contig layout)
    std::vector<uint32_t> plan_off;        // cached cand_ctg_off
    uint32_t plan_C = 0;
    DevBuf ws_small;                        // ctg_off | n_one | status | blk_ctg | blk_cnt
    DevBuf ws_start, ws_ent, ws_one, ws_tmp, ws_c2;
    uint32_t *d_ctg_off = nullptr, *d_n_one = nullptr, *d_status = nullptr, *d_blk_ctg = nullptr,
             *d_blk_cnt = nullptr;
    // host-run staging
    DevBuf h_in[9], h_out[2];
    // profiling events: 4 per run
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    bool pending_check = false;
};

namespace {

int fail(duet_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    g_last_error = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, e_ == hipErrorOutOfMemory ? DUET_ERR_OOM : DUET_ERR_HIP,           \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)

int reserve(duet_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return DUET_OK;
    if (b.ptr) HIP_TRY(ctx, hipFree(b.ptr));
    b.ptr = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    HIP_TRY(ctx, hipMalloc(&b.ptr, want));
    b.cap = want;
    return DUET_OK;
}

// (re)build the workspace for this contig layout; a no-op when it matches the cached plan
int ensure_plan(duet_ctx *ctx, const duet_ef_problem *pr, hipStream_t stream)
{
    const uint32_t K = pr->n_contigs, C = pr->n_cands;
    if (ctx->plan_C == C && ctx->plan_off.size() == (size_t)K + 1 &&
        memcmp(ctx->plan_off.data(), pr->cand_ctg_off, sizeof(uint32_t) * (K + 1)) == 0)
        return DUET_OK;
    const uint32_t B = (C + kCandPerBlock - 1) / kCandPerBlock;
    // contig of the first candidate of every 256-candidate block
    std::vector<uint32_t> blk_ctg(B);
    {
        uint32_t k = 0;
        for (uint32_t b = 0; b < B; ++b) {
            const uint32_t c = b * kCandPerBlock;
            while (c >= pr->cand_ctg_off[k + 1]) ++k;
            blk_ctg[b] = k;
        }
    }
    const size_t small_words = (size_t)(K + 1) + K + 8 + (size_t)B * 2;
    int rc;
    // the previous plan's buffers may still be in use by work queued on a stream
    HIP_TRY(ctx, hipDeviceSynchronize());
    if ((rc = reserve(ctx, ctx->ws_small, small_words * 4))) return rc;
    if ((rc = reserve(ctx, ctx->ws_start, C))) return rc;
    if ((rc = reserve(ctx, ctx->ws_ent, (size_t)B * kCandPerBlock * 8))) return rc;
    if ((rc = reserve(ctx, ctx->ws_one, (size_t)C * 4))) return rc;
    if ((rc = reserve(ctx, ctx->ws_tmp, (size_t)C * 4))) return rc;
    if ((rc = reserve(ctx, ctx->ws_c2, (size_t)B * kC2Quota * kC2Words * 4))) return rc;
    uint32_t *w = (uint32_t *)ctx->ws_small.ptr;
    ctx->d_ctg_off = w;            w += K + 1;
    ctx->d_n_one = w;              w += K;
    ctx->d_status = w;             w += 8;
    ctx->d_blk_ctg = w;            w += B;
    ctx->d_blk_cnt = w;
    HIP_TRY(ctx, hipMemcpy(ctx->d_ctg_off, pr->cand_ctg_off, sizeof(uint32_t) * (K + 1), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_blk_ctg, blk_ctg.data(), sizeof(uint32_t) * B, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemset(ctx->d_n_one, 0, sizeof(uint32_t) * ((size_t)K + 8)));
    HIP_TRY(ctx, hipMemset(ctx->ws_start.ptr, 0, C));
    hipLaunchKernelGGL(plan_mark_starts, dim3((K + 255) / 256), dim3(256), 0, 0, ctx->d_ctg_off, K,
                       (uint8_t *)ctx->ws_start.ptr);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipDeviceSynchronize());
    ctx->plan_off.assign(pr->cand_ctg_off, pr->cand_ctg_off + K + 1);
    ctx->plan_C = C;
    (void)stream;
    return DUET_OK;
}

int validate(duet_ctx *ctx, const duet_ef_problem *pr, const void *out_pred, const void *out_ps)
{
    if (!ctx) return fail(nullptr, DUET_ERR_INVALID, "null context");
    if (!pr) return fail(ctx, DUET_ERR_INVALID, "null problem");
    if (!pr->cand_ctg_off) return fail(ctx, DUET_ERR_INVALID, "cand_ctg_off is null");
    if (pr->n_contigs == 0 && pr->n_cands != 0) return fail(ctx, DUET_ERR_INVALID, "candidates without contigs");
    if (pr->cand_ctg_off[0] != 0 || pr->cand_ctg_off[pr->n_contigs] != pr->n_cands)
        return fail(ctx, DUET_ERR_INVALID, "cand_ctg_off must start at 0 and end at n_cands");
    for (uint32_t k = 0; k < pr->n_contigs; ++k)
        if (pr->cand_ctg_off[k] > pr->cand_ctg_off[k + 1])
            return fail(ctx, DUET_ERR_INVALID, "cand_ctg_off must be non-decreasing");
    if (pr->n_cands) {
        if (!pr->cand_pos || !pr->cand_svlen || !pr->cand_svread || !pr->cand_refread || !pr->cand_gt_ok ||
            !pr->cand_off || !pr->mark_read || !out_pred || !out_ps)
            return fail(ctx, DUET_ERR_INVALID, "null array");
        if (pr->n_reads && !pr->read_tag) return fail(ctx, DUET_ERR_INVALID, "read_tag is null");
    }
    return DUET_OK;
}

}  // namespace

extern "C" {

int duet_abi_version(void) { return DUET_ABI_VERSION; }

const char *duet_last_error(const duet_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

duet_ctx *duet_ctx_create(int device_id)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_last_error = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return nullptr;
    }
    if (device_id < 0 || device_id >= n) {
        g_last_error = "device id out of range";
        return nullptr;
    }
    if ((e = hipSetDevice(device_id)) != hipSuccess) {
        g_last_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return nullptr;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) {
        g_last_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return nullptr;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_last_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return nullptr;
    }
    duet_ctx *ctx = new duet_ctx();
    ctx->device = device_id;
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        g_last_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
        delete ctx;
        return nullptr;
    }
    return ctx;
}

void duet_ctx_destroy(duet_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (hipEvent_t ev : ctx->ev_pool) (void)hipEventDestroy(ev);
    DevBuf *all[] = {&ctx->ws_small, &ctx->ws_start, &ctx->ws_ent, &ctx->ws_one, &ctx->ws_tmp, &ctx->ws_c2};
    for (DevBuf *b : all) if (b->ptr) (void)hipFree(b->ptr);
    for (DevBuf &b : ctx->h_in) if (b.ptr) (void)hipFree(b.ptr);
    for (DevBuf &b : ctx->h_out) if (b.ptr) (void)hipFree(b.ptr);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int duet_ctx_set_profiling(duet_ctx *ctx, int enabled)
{
    if (!ctx) return fail(nullptr, DUET_ERR_INVALID, "null context");
    ctx->profiling = enabled != 0;
    return DUET_OK;
}

int duet_ctx_set_debug(duet_ctx *ctx, uint32_t flags)
{
    if (!ctx) return fail(nullptr, DUET_ERR_INVALID, "null context");
    ctx->dbg = flags;
    return DUET_OK;
}

int duet_ef_run_device(duet_ctx *ctx, const duet_ef_problem *pr, uint8_t *out_pred, uint32_t *out_ps, void *stream_)
{
    int rc = validate(ctx, pr, out_pred, out_ps);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (pr->n_cands == 0) return DUET_OK;
    if ((rc = ensure_plan(ctx, pr, stream))) return rc;

    Params p;
    p.K = pr->n_contigs; p.C = pr->n_cands; p.M = pr->n_marks;
    p.read_tag = pr->read_tag;
    p.cand_pos = pr->cand_pos; p.cand_svlen = pr->cand_svlen; p.cand_svread = pr->cand_svread;
    p.cand_refread = pr->cand_refread; p.cand_gt_ok = pr->cand_gt_ok;
    p.cand_off = pr->cand_off; p.mark_read = pr->mark_read;
    p.svlen_thres = pr->svlen_thres; p.suppread_thres = pr->suppread_thres;
    p.ctg_off = ctx->d_ctg_off; p.ctg_start = (const uint8_t *)ctx->ws_start.ptr; p.blk_ctg = ctx->d_blk_ctg;
    p.blk_cnt = ctx->d_blk_cnt; p.seed_ent = (uint64_t *)ctx->ws_ent.ptr;
    p.onebuf = (uint32_t *)ctx->ws_one.ptr; p.tmpbuf = (uint32_t *)ctx->ws_tmp.ptr;
    p.n_one = ctx->d_n_one; p.c2rec = (uint32_t *)ctx->ws_c2.ptr; p.status = ctx->d_status;
    p.out_pred = out_pred; p.out_ps = out_ps;
    p.dbg = ctx->dbg;

    hipEvent_t *ev = nullptr;
    if (ctx->profiling) {
        while (ctx->ev_pool.size() < ctx->ev_used + 4) {
            hipEvent_t e;
            HIP_TRY(ctx, hipEventCreate(&e));
            ctx->ev_pool.push_back(e);
        }
        ev = &ctx->ev_pool[ctx->ev_used];
        ctx->ev_used += 4;
        HIP_TRY(ctx, hipEventRecord(ev[0], stream));
    }
    const uint32_t blocks = (pr->n_cands + kCandPerBlock - 1) / kCandPerBlock;
    if (((uintptr_t)pr->mark_read & 15) == 0)
        hipLaunchKernelGGL(ef_classify<true>, dim3(blocks), dim3(kCandPerBlock), 0, stream, p);
    else
        hipLaunchKernelGGL(ef_classify<false>, dim3(blocks), dim3(kCandPerBlock), 0, stream, p);
    if (ev) HIP_TRY(ctx, hipEventRecord(ev[1], stream));
    hipLaunchKernelGGL(ef_seed_sort, dim3(pr->n_contigs), dim3(kSortThreads), 0, stream, p);
    if (ev) HIP_TRY(ctx, hipEventRecord(ev[2], stream));
    hipLaunchKernelGGL(ef_finalize, dim3((pr->n_cands + 255) / 256), dim3(256), 0, stream, p);
    if (ev) HIP_TRY(ctx, hipEventRecord(ev[3], stream));
    HIP_TRY(ctx, hipGetLastError());
    ctx->pending_check = true;
    return DUET_OK;
}

int duet_ef_check(duet_ctx *ctx, void *stream_)
{
    if (!ctx) return fail(nullptr, DUET_ERR_INVALID, "null context");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream_));
    if (!ctx->pending_check || !ctx->d_status) return DUET_OK;
    ctx->pending_check = false;
    uint32_t st = 0;
    HIP_TRY(ctx, hipMemcpy(&st, ctx->d_status, 4, hipMemcpyDeviceToHost));
    if (st & 1u) {
        HIP_TRY(ctx, hipMemset(ctx->d_status, 0, 4));
        return fail(ctx, DUET_ERR_DIV_ZERO, "division by zero: svread + refread == 0 for a candidate that reaches the decision");
    }
    return DUET_OK;
}

int duet_ef_profile_collect(duet_ctx *ctx, duet_ef_stats *stats)
{
    if (!ctx || !stats) return fail(ctx, DUET_ERR_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    memset(stats, 0, sizeof(*stats));
    const size_t runs = ctx->ev_used / 4;
    double acc[DUET_N_KERNELS] = {0, 0, 0}, tot = 0;
    for (size_t r = 0; r < runs; ++r) {
        hipEvent_t *ev = &ctx->ev_pool[4 * r];
        HIP_TRY(ctx, hipEventSynchronize(ev[3]));
        for (int i = 0; i < DUET_N_KERNELS; ++i) {
            float ms = 0;
            HIP_TRY(ctx, hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            acc[i] += ms;
        }
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ev[0], ev[3]));
        tot += ms;
    }
    if (runs) {
        for (int i = 0; i < DUET_N_KERNELS; ++i) stats->kernel_ms[i] = (float)(acc[i] / runs);
        stats->total_ms = (float)(tot / runs);
    }
    stats->n_profiled_runs = (uint32_t)runs;
    ctx->ev_used = 0;
    return DUET_OK;
}

int duet_ef_get_seed_ps(duet_ctx *ctx, uint32_t contig, uint32_t *out, uint32_t cap)
{
    if (!ctx) return fail(nullptr, DUET_ERR_INVALID, "null context");
    if (contig + 1 >= ctx->plan_off.size()) return fail(ctx, DUET_ERR_INVALID, "contig out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    uint32_t n = 0;
    HIP_TRY(ctx, hipMemcpy(&n, ctx->d_n_one + contig, 4, hipMemcpyDeviceToHost));
    const uint32_t take = n < cap ? n : cap;
    if (take && out)
        HIP_TRY(ctx, hipMemcpy(out, (uint32_t *)ctx->ws_one.ptr + ctx->plan_off[contig], (size_t)take * 4,
                               hipMemcpyDeviceToHost));
    return (int)n;
}

int duet_ef_run_host(duet_ctx *ctx, const duet_ef_problem *pr, uint8_t *out_pred, uint32_t *out_ps, duet_ef_stats *stats)
{
    int rc = validate(ctx, pr, out_pred, out_ps);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->algorithmic_bytes = 12ull * pr->n_marks + 27ull * pr->n_cands + 8ull * pr->n_reads;
    }
    const uint32_t C = pr->n_cands, M = pr->n_marks, R = pr->n_reads;
    if (C == 0) return DUET_OK;
    const void *src[9] = {pr->read_tag, pr->cand_pos, pr->cand_svlen, pr->cand_svread, pr->cand_refread,
                          pr->cand_gt_ok, pr->cand_off, pr->mark_read, nullptr};
    const size_t bytes[9] = {(size_t)R * 8, (size_t)C * 4, (size_t)C * 4, (size_t)C * 4, (size_t)C * 4,
                             (size_t)C, ((size_t)C + 1) * 4, (size_t)M * 4, 0};
    hipStream_t s = ctx->own_stream;
    for (int i = 0; i < 8; ++i) {
        if ((rc = reserve(ctx, ctx->h_in[i], bytes[i] ? bytes[i] : 16))) return rc;
        if (bytes[i]) HIP_TRY(ctx, hipMemcpyAsync(ctx->h_in[i].ptr, src[i], bytes[i], hipMemcpyHostToDevice, s));
    }
    if ((rc = reserve(ctx, ctx->h_out[0], C))) return rc;
    if ((rc = reserve(ctx, ctx->h_out[1], (size_t)C * 4))) return rc;
    duet_ef_problem d = *pr;
    d.read_tag = (const uint64_t *)ctx->h_in[0].ptr;
    d.cand_pos = (const uint32_t *)ctx->h_in[1].ptr;
    d.cand_svlen = (const uint32_t *)ctx->h_in[2].ptr;
    d.cand_svread = (const uint32_t *)ctx->h_in[3].ptr;
    d.cand_refread = (const uint32_t *)ctx->h_in[4].ptr;
    d.cand_gt_ok = (const uint8_t *)ctx->h_in[5].ptr;
    d.cand_off = (const uint32_t *)ctx->h_in[6].ptr;
    d.mark_read = (const uint32_t *)ctx->h_in[7].ptr;
    if ((rc = duet_ef_run_device(ctx, &d, (uint8_t *)ctx->h_out[0].ptr, (uint32_t *)ctx->h_out[1].ptr, s))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out_pred, ctx->h_out[0].ptr, C, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(out_ps, ctx->h_out[1].ptr, (size_t)C * 4, hipMemcpyDeviceToHost, s));
    if ((rc = duet_ef_check(ctx, s))) return rc;
    if (stats) {
        std::vector<uint32_t> n_one(pr->n_contigs);
        HIP_TRY(ctx, hipMemcpy(n_one.data(), ctx->d_n_one, sizeof(uint32_t) * pr->n_contigs, hipMemcpyDeviceToHost));
        uint32_t tot = 0;
        for (uint32_t v : n_one) tot += v;
        stats->n_seed_ps = tot;
        if (ctx->profiling) {
            duet_ef_stats prof;
            if ((rc = duet_ef_profile_collect(ctx, &prof))) return rc;
            memcpy(stats->kernel_ms, prof.kernel_ms, sizeof(prof.kernel_ms));
            stats->total_ms = prof.total_ms;
            stats->n_profiled_runs = prof.n_profiled_runs;
        }
    }
    return DUET_OK;
}

}  // extern "C"
