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
constexpr uint8_t kClass2 = 8;                    // multi-PS candidate: evaluated in ef_finalize
constexpr uint8_t kDivZero = 32;                  // kept candidate with svread + refread == 0

constexpr int kCandPerBlock = 256;
constexpr int kChunk = 4096;                      // marks staged in LDS per pass (32 KiB of tags)
constexpr int kSortThreads = 1024;
constexpr uint32_t kSortLds = 16384;              // seeds sorted in LDS (64 KiB)

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
    // workspace
    const uint32_t *ctg_off;          // [K+1] device copy of cand_ctg_off
    const uint32_t *tab_off;          // [K]   first slot of contig k's hash set
    const uint32_t *tab_mask;         // [K]   slots-1 (power of two)
    uint32_t *tab;                    // hash slots, kEmpty when free
    uint32_t *seed_cnt;               // [K]   distinct seeds inserted so far
    uint32_t *seedbuf;                // [C]   per contig (at ctg_off[k]): slot list, then sorted seeds
    uint32_t *n_one;                  // [K]   length of contig k's sorted seed array
    uint32_t *status;                 // [0] = div-zero flag
    uint8_t *out_pred;
    uint32_t *out_ps;
};

// ---------------------------------------------------------------------------------------------
// kernel 1
// ---------------------------------------------------------------------------------------------

template <bool VEC>
__global__ __launch_bounds__(kCandPerBlock) void ef_classify(const Params p)
{
    __shared__ uint64_t s_tag[kChunk];
    __shared__ uint32_t s_off[kCandPerBlock + 1];
    __shared__ uint32_t s_seed[kCandPerBlock];

    const uint32_t tid = threadIdx.x;
    const uint32_t c0 = blockIdx.x * kCandPerBlock;
    const uint32_t nc = min((uint32_t)kCandPerBlock, p.C - c0);
    for (uint32_t i = tid; i <= nc; i += kCandPerBlock) s_off[i] = p.cand_off[c0 + i];

    // candidate scalars, coalesced
    const bool live = tid < nc;
    const uint32_t c = c0 + tid;
    uint32_t svlen = 0, svread = 0, refread = 0, gt_ok = 0;
    if (live) {
        svlen = p.cand_svlen[c];
        svread = p.cand_svread[c];
        refread = p.cand_refread[c];
        gt_ok = p.cand_gt_ok[c];
    }
    __syncthreads();
    const uint32_t m_begin = s_off[0], m_end = s_off[nc];
    const uint32_t my_b = live ? s_off[tid] : 0, my_e = live ? s_off[tid + 1] : 0;

    // per-candidate running state over marks in list order
    uint32_t n_ps = 0, first_ps = 0;          // distinct PS among tagged marks: 0, 1, 2(=many)
    uint32_t seed = kEmpty;                   // PS of first voter
    uint32_t last_ps = 0;                     // PS of last voter (:77)
    uint32_t h1 = 0, h2 = 0;
    uint64_t t1 = 0, t2 = 0;

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
        // ---- consume: each thread walks its candidate's part of this chunk --------------------
        const uint32_t lo = max(my_b, cs);
        const uint32_t hi = min(my_e, cs + (uint32_t)kChunk);
        for (uint32_t m = lo; m < hi; ++m) {
            const uint64_t tag = s_tag[m - cs];
            if (tag == kUntagged) continue;
            const uint32_t ps = tag_ps(tag);
            if (n_ps == 0) { n_ps = 1; first_ps = ps; }
            else if (ps != first_ps) n_ps = 2;
            const uint32_t pc = tag_pc(tag);
            if (pc <= kPcMax) {
                if (seed == kEmpty) seed = ps;
                last_ps = ps;
                const uint32_t hap = tag_hap(tag);
                if (hap == 1) { ++h1; t1 += pc; }
                else if (hap == 2) { ++h2; t2 += pc; }
            }
        }
        __syncthreads();
    }

    // ---- per-candidate result ------------------------------------------------------------------
    uint8_t code = 0;
    uint32_t ps_out = 0;
    bool want_seed = false;
    if (live) {
        const bool kept = svlen >= p.svlen_thres && svread >= p.suppread_thres && gt_ok != 0;      // :189-190
        if (kept) {
            want_seed = (n_ps == 1) && seed != kEmpty;                                             // :198-203
            if ((uint64_t)svread + (uint64_t)refread == 0) {
                code = kDivZero;
            } else if (n_ps == 2) {
                code = kClass2;
            } else {
                Vote v;
                v.hap1 = n_ps ? h1 : 0; v.hap2 = n_ps ? h2 : 0; v.hap0 = 0;
                v.allhap = v.hap1 + v.hap2;
                v.t1 = n_ps ? t1 : 0; v.t2 = n_ps ? t2 : 0;
                code = (uint8_t)decide((int)n_ps, v, my_e - my_b, svread, refread);
                ps_out = last_ps;
                if (n_ps == 0 || (v.hap1 == 0 && v.hap2 == 0)) code |= kNeedNearest;               // :106
            }
        }
        p.out_pred[c] = code;
        p.out_ps[c] = ps_out;
    }

    // ---- seed set insertion (neighbour-deduplicated) -------------------------------------------
    s_seed[tid] = want_seed ? seed : kEmpty;
    __syncthreads();
    if (want_seed) {
        const uint32_t k = find_contig(p.ctg_off, p.K, c);
        const bool same_as_prev = tid > 0 && s_seed[tid - 1] == seed && (c - 1) >= p.ctg_off[k];
        if (!same_as_prev) {
            const uint32_t mask = p.tab_mask[k];
            uint32_t *tab = p.tab + p.tab_off[k];
            uint32_t slot = (seed * 2654435761u) & mask;
            for (;;) {
                const uint32_t old = atomicCAS(&tab[slot], kEmpty, seed);
                if (old == kEmpty) {
                    const uint32_t idx = atomicAdd(&p.seed_cnt[k], 1u);
                    p.seedbuf[p.ctg_off[k] + idx] = p.tab_off[k] + slot;
                    break;
                }
                if (old == seed) break;
                slot = (slot + 1) & mask;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// kernel 2
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

__global__ __launch_bounds__(kSortThreads) void ef_seed_sort(const Params p)
{
    __shared__ uint32_t s_key[kSortLds];
    const uint32_t k = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = p.seed_cnt[k];
    uint32_t *buf = p.seedbuf + p.ctg_off[k];
    __syncthreads();
    if (tid == 0) { p.n_one[k] = n; p.seed_cnt[k] = 0; }
    if (n == 0) return;
    if (n <= kSortLds) {
        for (uint32_t i = tid; i < n; i += kSortThreads) {
            const uint32_t slot = buf[i];
            s_key[i] = p.tab[slot];
            p.tab[slot] = kEmpty;
        }
        __syncthreads();
        bitonic_sort(s_key, n, tid, kSortThreads);
        for (uint32_t i = tid; i < n; i += kSortThreads) buf[i] = s_key[i];
    } else {
        for (uint32_t i = tid; i < n; i += kSortThreads) {
            const uint32_t slot = buf[i];
            buf[i] = p.tab[slot];
            p.tab[slot] = kEmpty;
        }
        __syncthreads();
        bitonic_sort(buf, n, tid, kSortThreads);
    }
}

// ---------------------------------------------------------------------------------------------
// kernel 3
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *__restrict__ a, uint32_t n, uint32_t key)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// sv_phasing_fn.py:107-111 -- ties go to the larger seed
__device__ __forceinline__ uint32_t nearest_ps(const uint32_t *__restrict__ a, uint32_t n, uint32_t pos)
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

__global__ __launch_bounds__(256) void ef_finalize(const Params p)
{
    __shared__ uint32_t s_k0, s_any_empty;
    const uint32_t tid = threadIdx.x;
    const uint32_t c0 = blockIdx.x * 256u;
    const uint32_t c = c0 + tid;
    if (tid == 0) {
        const uint32_t last = min(c0 + 255u, p.C - 1);
        const uint32_t k0 = find_contig(p.ctg_off, p.K, c0), k1 = find_contig(p.ctg_off, p.K, last);
        uint32_t any = 0;
        for (uint32_t k = k0; k <= k1; ++k) any |= (p.n_one[k] == 0);
        s_k0 = k0;
        s_any_empty = any;
    }
    __syncthreads();
    if (c >= p.C) return;
    const uint8_t code = p.out_pred[c];
    if (code < 4 && !s_any_empty) return;                      // already final
    uint32_t k = s_k0;
    while (c >= p.ctg_off[k + 1]) ++k;
    const uint32_t n_one = p.n_one[k];
    if (n_one == 0) {                                          // :209-210
        if (code != 0) p.out_pred[c] = 0;
        p.out_ps[c] = 0;
        return;
    }
    if (code < 4) return;
    if (code & kDivZero) {                                     // :123 would raise
        atomicOr(&p.status[0], 1u);
        p.out_pred[c] = 0;
        p.out_ps[c] = 0;
        return;
    }
    const uint32_t *one = p.seedbuf + p.ctg_off[k];
    if (code & kClass2) {                                      // :85-105, :148-155
        const uint32_t b = p.cand_off[c], e = p.cand_off[c + 1];
        Vote v = {0, 0, 0, 0, 0, 0};
        uint32_t ps = 0, best = 0;
        for (uint32_t m = b; m < e; ++m) {
            const uint64_t t = fetch_tag(p, m);
            if (t == kUntagged || tag_pc(t) > kPcMax) continue;
            ++v.allhap;
        }
        uint32_t done_ps = kEmpty;                             // last group evaluated (cheap duplicate skip)
        for (uint32_t m = b; m < e; ++m) {
            const uint64_t t = fetch_tag(p, m);
            if (t == kUntagged || tag_pc(t) > kPcMax) continue;
            const uint32_t g = tag_ps(t);
            if (g == done_ps || g == ps && best) continue;
            const uint32_t at = lower_bound_u32(one, n_one, g);
            if (at >= n_one || one[at] != g) continue;         // :91
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
                v.hap0 = v.allhap - n1 - n2;                   // only with a winner (:105)
            }
        }
        if (v.hap1 == 0 && v.hap2 == 0) ps = nearest_ps(one, n_one, p.cand_pos[c]);       // :106
        p.out_pred[c] = (uint8_t)decide(2, v, e - b, p.cand_svread[c], p.cand_refread[c]);
        p.out_ps[c] = ps;
        return;
    }
    // kNeedNearest
    p.out_pred[c] = code & 3;
    p.out_ps[c] = nearest_ps(one, n_one, p.cand_pos[c]);
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
    hipStream_t own_stream = nullptr;
    // plan (workspace keyed by the contig layout)
    std::vector<uint32_t> plan_off;        // cached cand_ctg_off
    uint32_t plan_C = 0;
    DevBuf ws_small;                        // ctg_off | tab_off | tab_mask | seed_cnt | n_one | status
    DevBuf ws_tab, ws_seed;
    uint32_t *d_ctg_off = nullptr, *d_tab_off = nullptr, *d_tab_mask = nullptr, *d_seed_cnt = nullptr,
             *d_n_one = nullptr, *d_status = nullptr;
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

uint32_t pow2_ceil(uint64_t x)
{
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return (uint32_t)p;
}

// (re)build the workspace for this contig layout; a no-op when it matches the cached plan
int ensure_plan(duet_ctx *ctx, const duet_ef_problem *pr, hipStream_t stream)
{
    const uint32_t K = pr->n_contigs, C = pr->n_cands;
    if (ctx->plan_C == C && ctx->plan_off.size() == (size_t)K + 1 &&
        memcmp(ctx->plan_off.data(), pr->cand_ctg_off, sizeof(uint32_t) * (K + 1)) == 0)
        return DUET_OK;
    std::vector<uint32_t> tab_off(K + 1), tab_mask(K + 1);
    uint64_t total = 0;
    for (uint32_t k = 0; k < K; ++k) {
        const uint64_t cnt = pr->cand_ctg_off[k + 1] - pr->cand_ctg_off[k];
        const uint32_t slots = pow2_ceil(cnt * 2 < 16 ? 16 : cnt * 2);
        tab_off[k] = (uint32_t)total;
        tab_mask[k] = slots - 1;
        total += slots;
    }
    if (total >= 0xFFFFFFF0ull) return fail(ctx, DUET_ERR_INVALID, "too many candidates for the seed hash set");
    const size_t small_words = (size_t)(K + 1) * 3 + (size_t)K * 2 + 8;
    int rc;
    // the previous plan's buffers may still be in use by work queued on a stream
    HIP_TRY(ctx, hipDeviceSynchronize());
    if ((rc = reserve(ctx, ctx->ws_small, small_words * 4))) return rc;
    if ((rc = reserve(ctx, ctx->ws_tab, (size_t)total * 4))) return rc;
    if ((rc = reserve(ctx, ctx->ws_seed, (size_t)(C ? C : 1) * 4))) return rc;
    uint32_t *w = (uint32_t *)ctx->ws_small.ptr;
    ctx->d_ctg_off = w;            w += K + 1;
    ctx->d_tab_off = w;            w += K + 1;
    ctx->d_tab_mask = w;           w += K + 1;
    ctx->d_seed_cnt = w;           w += K;
    ctx->d_n_one = w;              w += K;
    ctx->d_status = w;
    HIP_TRY(ctx, hipMemcpy(ctx->d_ctg_off, pr->cand_ctg_off, sizeof(uint32_t) * (K + 1), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_tab_off, tab_off.data(), sizeof(uint32_t) * (K + 1), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_tab_mask, tab_mask.data(), sizeof(uint32_t) * (K + 1), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemset(ctx->d_seed_cnt, 0, sizeof(uint32_t) * ((size_t)K * 2 + 8)));
    HIP_TRY(ctx, hipMemset(ctx->ws_tab.ptr, 0xFF, (size_t)total * 4));
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
    DevBuf *all[] = {&ctx->ws_small, &ctx->ws_tab, &ctx->ws_seed};
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
    p.ctg_off = ctx->d_ctg_off; p.tab_off = ctx->d_tab_off; p.tab_mask = ctx->d_tab_mask;
    p.tab = (uint32_t *)ctx->ws_tab.ptr; p.seed_cnt = ctx->d_seed_cnt;
    p.seedbuf = (uint32_t *)ctx->ws_seed.ptr; p.n_one = ctx->d_n_one; p.status = ctx->d_status;
    p.out_pred = out_pred; p.out_ps = out_ps;

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
        HIP_TRY(ctx, hipMemcpy(out, (uint32_t *)ctx->ws_seed.ptr + ctx->plan_off[contig], (size_t)take * 4,
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
