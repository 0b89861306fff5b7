// duet_cluster.hip -- gfx950 kernels and C ABI for stage A0: span-position clustering of SV marks into
// candidates (what `--cluster_max_distance` controls).
//
// The reference delegates this stage to the external `svim alignment` binary (src/duet/sv_calling.py:13-15),
// so there is no reference code to follow; the rule implemented here is this repository's own deterministic
// statement of the published SVIM 1.4.2 scheme, normative text in oracle/cluster_oracle.c / DESIGN.md section 9.
//
// Pipeline (all on one stream):
//   cl_keys        key = (contig, type, centre = pos + span/2) packed into the fewest bits, val = mark index
//   radix sort     stable LSD, 8-bit digits: rx_hist -> scan -> rx_scatter per pass (ballot-ranked, no atomics
//                  on the data path, so the order is deterministic)
//   cl_heads1/2    partition starts: contig/type change, centre gap > part_gap, or part_max marks reached
//   cl_parts       partition start list (from an exclusive scan of the head flags)
//   cl_cluster     16 / 32 / 64 lanes per partition (one lane for <= 8 marks): span-position distances into an LDS
//                  triangle (fp64), average linkage by repeated group-wide argmin + Lance-Williams update
//   cl_emit        per partition: clusters by smallest member, members in sorted order -> order[], cand_*[]
//
// Bit-exactness vs the oracle: distances and updates are the same binary64 expressions in the same order
// (-ffp-contract=off); the argmin breaks ties by the smallest (first, second) index pair.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>

#include "duet_ef.h"
#include "duet_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int kRxThreads = 256;
constexpr int kRxItems = 16;
constexpr int kRxTile = kRxThreads * kRxItems;        // keys per radix block
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

struct ClParams {
    uint32_t M;
    uint32_t part_gap, part_max;
    double max_dist, normalizer;
    uint32_t centre_bits, type_bits;                  // key = ((contig << type_bits | type) << centre_bits) | centre
    const uint16_t *contig;
    const uint8_t *type;
    const uint32_t *pos, *span;
    const uint32_t *sorted;                           // mark index at each sorted position
    const uint32_t *part_start;                       // [P+1]
    const uint32_t *n_parts;                          // device scalar
    uint8_t *label;                                   // [M] root (index inside its partition) of each sorted position
    uint32_t *pc;                                     // [P] clusters per partition
    const uint32_t *cbase;                            // [M] at a partition's start position: its first candidate
    // outputs
    uint32_t *order, *cand_off, *cand_pos, *cand_span;
    uint16_t *cand_contig;
    uint8_t *cand_type;
};

__device__ __forceinline__ uint64_t centre_of(uint32_t pos, uint32_t span) { return (uint64_t)pos + (span >> 1); }

// ---------------------------------------------------------------------------------------------
// keys + radix sort
// ---------------------------------------------------------------------------------------------

__global__ void cl_keys(const ClParams p, uint64_t *keys, uint32_t *vals)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.M) return;
    const uint64_t hi = ((uint64_t)p.contig[i] << p.type_bits) | (uint64_t)p.type[i];
    keys[i] = (hi << p.centre_bits) | centre_of(p.pos[i], p.span[i]);
    vals[i] = i;
}

__global__ __launch_bounds__(kRxThreads) void rx_hist(const uint64_t *keys, uint32_t n, uint32_t shift, uint32_t nb,
                                                      uint32_t *hist /* [256][nb] */)
{
    __shared__ uint32_t s_h[256];
    const uint32_t tid = threadIdx.x;
    s_h[tid] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kRxTile;
#pragma unroll
    for (int it = 0; it < kRxItems; ++it) {
        const uint32_t i = base + it * kRxThreads + tid;
        if (i < n) atomicAdd(&s_h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)tid * nb + blockIdx.x] = s_h[tid];
}

// generic block-tiled scan: op 0 = exclusive sum, op 1 = inclusive max
template <int OP>
__device__ __forceinline__ uint32_t scan_op(uint32_t a, uint32_t b) { return OP == 0 ? a + b : (a > b ? a : b); }

template <int OP>
__global__ __launch_bounds__(kScanThreads) void scan_reduce(const uint32_t *in, uint32_t n, uint32_t *part)
{
    __shared__ uint32_t s_w[kScanThreads / 64];
    const uint32_t tid = threadIdx.x;
    const uint32_t base = blockIdx.x * kScanTile + tid * kScanItems;
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j)
        if (base + j < n) acc = scan_op<OP>(acc, in[base + j]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc = scan_op<OP>(acc, __shfl_xor(acc, d, 64));
    if ((tid & 63) == 0) s_w[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kScanThreads / 64; ++w) t = scan_op<OP>(t, s_w[w]);
        part[blockIdx.x] = t;
    }
}

// single block: part[i] <- combination of part[0..i) (exclusive); *total <- combination of everything
template <int OP>
__global__ __launch_bounds__(1024) void scan_spine(uint32_t *part, uint32_t n, uint32_t *total)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + tid;
        const uint32_t v = i < n ? part[i] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if ((int)lane >= d) x = scan_op<OP>(x, y);
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        uint32_t before = s_carry;
        for (uint32_t w = 0; w < wave; ++w) before = scan_op<OP>(before, s_w[w]);
        // exclusive value for element i: everything before it
        uint32_t excl = before;
        const uint32_t prev_in_wave = __shfl_up(x, 1, 64);
        if (lane > 0) excl = scan_op<OP>(before, prev_in_wave);
        if (i < n) part[i] = excl;
        __syncthreads();
        if (tid == 1023) s_carry = scan_op<OP>(before, x);
        __syncthreads();
    }
    if (tid == 0 && total) *total = s_carry;
}

// out[i] = exclusive sum (OP 0) / inclusive max (OP 1) of in[0..i] given the per-tile carries in part[]
template <int OP>
__global__ __launch_bounds__(kScanThreads) void scan_apply(const uint32_t *in, uint32_t n, const uint32_t *part,
                                                           uint32_t *out)
{
    __shared__ uint32_t s_w[kScanThreads / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t base = blockIdx.x * kScanTile + tid * kScanItems;
    uint32_t v[kScanItems];
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        v[j] = base + j < n ? in[base + j] : 0u;
        acc = scan_op<OP>(acc, v[j]);
    }
    uint32_t x = acc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if ((int)lane >= d) x = scan_op<OP>(x, y);
    }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    uint32_t run = part[blockIdx.x];
    for (uint32_t w = 0; w < wave; ++w) run = scan_op<OP>(run, s_w[w]);
    const uint32_t prev = __shfl_up(x, 1, 64);
    if (lane > 0) run = scan_op<OP>(run, prev);
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        if (OP == 0) {
            if (base + j < n) out[base + j] = run;
            run += v[j];
        } else {
            run = scan_op<OP>(run, v[j]);
            if (base + j < n) out[base + j] = run;
        }
    }
}

// stable scatter of one 8-bit digit; hist holds the scanned (digit-major) offsets
__global__ __launch_bounds__(kRxThreads) void rx_scatter(const uint64_t *keys_in, const uint32_t *vals_in, uint32_t n,
                                                         uint32_t shift, uint32_t nb, const uint32_t *hist,
                                                         uint64_t *keys_out, uint32_t *vals_out)
{
    __shared__ uint32_t s_base[256];
    __shared__ uint32_t s_wcnt[kRxThreads / 64][256];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    s_base[tid] = hist[(size_t)tid * nb + blockIdx.x];
#pragma unroll
    for (int w = 0; w < kRxThreads / 64; ++w) s_wcnt[w][tid] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kRxTile;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int it = 0; it < kRxItems; ++it) {
        const uint32_t i = base + it * kRxThreads + tid;
        const bool valid = i < n;
        const uint64_t key = valid ? keys_in[i] : 0ull;
        const uint32_t val = valid ? vals_in[i] : 0u;
        const uint32_t d = (uint32_t)(key >> shift) & 255u;
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bm = __ballot(bit);
            same &= bit ? bm : ~bm;
        }
        const uint32_t rank = (uint32_t)__popcll(same & lt);
        if (valid && rank == 0) s_wcnt[wave][d] = (uint32_t)__popcll(same);
        __syncthreads();
        if (valid) {
            uint32_t at = s_base[d] + rank;
            for (uint32_t w = 0; w < wave; ++w) at += s_wcnt[w][d];
            keys_out[at] = key;
            vals_out[at] = val;
        }
        __syncthreads();
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < kRxThreads / 64; ++w) {
            tot += s_wcnt[w][tid];
            s_wcnt[w][tid] = 0;
        }
        s_base[tid] += tot;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// partitions
// ---------------------------------------------------------------------------------------------

// headpos[i] = i if sorted position i starts a natural partition (contig/type change or centre gap), else 0.
// Everything needed is in the sorted keys: (contig, type) in the high bits, the centre in the low bits.
__global__ void cl_heads1(const ClParams p, const uint64_t *keys, uint32_t *headpos)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.M) return;
    uint32_t h = 0;
    if (i > 0) {
        const uint64_t a = keys[i - 1], b = keys[i];
        const uint64_t cm = (1ull << p.centre_bits) - 1ull;
        const bool cut = (a >> p.centre_bits) != (b >> p.centre_bits) || (b & cm) - (a & cm) > (uint64_t)p.part_gap;
        h = cut ? i : 0u;
    }
    headpos[i] = h;
}

// flag[i] = 1 if i starts a partition: natural head, or every part_max marks after it
__global__ void cl_heads2(const ClParams p, const uint32_t *head_of, uint32_t *flag)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.M) return;
    flag[i] = ((i - head_of[i]) % p.part_max) == 0 ? 1u : 0u;
}

__global__ void cl_parts(const ClParams p, const uint32_t *flag, const uint32_t *pid, uint32_t *part_start,
                         const uint32_t *n_parts)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p.M && flag[i]) part_start[pid[i]] = i;
    if (i == 0) part_start[*n_parts] = p.M;
}

// work lists by partition size (which agglomeration kernel variant takes it); list order is irrelevant --
// every partition writes to its own fixed output range -- so a (wave-aggregated) atomic append is fine
constexpr int kClasses = 5;
__device__ __forceinline__ int size_class(uint32_t n) { return n <= 8 ? 0 : (n <= 16 ? 1 : (n <= 32 ? 2 : (n <= 48 ? 3 : 4))); }

__global__ void cl_classes(const ClParams p, uint32_t *lists /* [kClasses][M] */, uint32_t *counts /* [kClasses] */)
{
    const uint32_t part = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = part < *p.n_parts;
    const int cls = live ? size_class(p.part_start[part + 1] - p.part_start[part]) : -1;
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int c = 0; c < kClasses; ++c) {
        const unsigned long long m = __ballot(cls == c);
        if (!m) continue;
        uint32_t base = 0;
        if (lane == (uint32_t)__ffsll((long long)m) - 1u) base = atomicAdd(&counts[c], (uint32_t)__popcll(m));
        base = __shfl(base, __ffsll((long long)m) - 1, 64);
        if (cls == c) lists[(size_t)c * p.M + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = part;
    }
}

// pcat[i] = clusters of the partition that starts at sorted position i, 0 elsewhere (may alias flag)
__global__ void cl_pcat(const ClParams p, const uint32_t *flag, const uint32_t *pid, uint32_t *pcat)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.M) return;
    pcat[i] = flag[i] ? p.pc[pid[i]] : 0u;
}

// ---------------------------------------------------------------------------------------------
// agglomeration: one wavefront (64-thread workgroup) per partition
// ---------------------------------------------------------------------------------------------

// pair index q of the upper triangle (row i: j = i+1..n-1, rows in order) -> (i, j)
__device__ __forceinline__ uint32_t tri_row_start(uint32_t i, uint32_t n) { return i * n - i * (i + 1) / 2; }

__device__ __forceinline__ void tri_decode(uint32_t q, uint32_t n, uint32_t pairs, uint32_t &i, uint32_t &j)
{
    // rows counted from the end have lengths 1, 2, 3, ...: the pair's distance from the end picks the row
    const uint32_t r = pairs - 1 - q;
    uint32_t t = (uint32_t)((__fsqrt_rn(8.0f * (float)r + 1.0f) - 1.0f) * 0.5f);
    while ((t + 1) * (t + 2) / 2 <= r) ++t;
    while (t * (t + 1) / 2 > r) --t;
    i = n - 2 - t;
    j = i + 1 + (q - tri_row_start(i, n));
}

// GROUP lanes of a wavefront work on one partition (GROUP = 16 / 32 / 64 for up to 16 / 32 / 128 marks), so a
// wave carries 4 / 2 / 1 partitions at once and the fixed cost of a merge step (argmin butterfly, decode, two
// barriers) is shared.  Control flow is wave-uniform; groups that are done (or idle) are predicated off.
template <int GROUP, int NMAX>
__global__ __launch_bounds__(64) void cl_cluster(const ClParams p, const uint32_t *list, const uint32_t *count)
{
    constexpr int SUBS = 64 / GROUP;
    constexpr int TRI = NMAX * (NMAX - 1) / 2;
    __shared__ double s_d[SUBS][TRI];
    __shared__ uint32_t s_pos[SUBS][NMAX], s_span[SUBS][NMAX], s_lab[SUBS][NMAX], s_size[SUBS][NMAX];
    const uint32_t lane = threadIdx.x, sub = lane / GROUP, sl = lane % GROUP;
    const unsigned long long gmask = GROUP == 64 ? ~0ull : (((1ull << (GROUP & 63)) - 1ull) << (sub * GROUP));
    const uint32_t L = *count;
    const double inf = __builtin_inf();
    for (uint32_t base = blockIdx.x * SUBS; base < L; base += gridDim.x * SUBS) {
        const uint32_t li = base + sub;
        const bool has = li < L;
        const uint32_t part = has ? list[li] : 0u;
        const uint32_t s = has ? p.part_start[part] : 0u;
        const uint32_t n = has ? p.part_start[part + 1] - s : 0u;
        __syncthreads();
        for (uint32_t i = sl; i < n; i += GROUP) {
            const uint32_t a = p.sorted[s + i];
            s_pos[sub][i] = p.pos[a];
            s_span[sub][i] = p.span[a];
            s_lab[sub][i] = i;
            s_size[sub][i] = 1;
        }
        __syncthreads();
        const uint32_t pairs = n ? n * (n - 1) / 2 : 0u;
        // span-position distance of every pair, one pass over the triangle
        for (uint32_t q = sl; q < pairs; q += GROUP) {
            uint32_t i, j;
            tri_decode(q, n, pairs, i, j);
            const uint32_t pi = s_pos[sub][i], spi = s_span[sub][i], pj = s_pos[sub][j], spj = s_span[sub][j];
            const uint64_t si = pi, ei = (uint64_t)pi + spi, ci = centre_of(pi, spi);
            const uint64_t sj = pj, ej = (uint64_t)pj + spj, cj = centre_of(pj, spj);
            uint64_t m = si > sj ? si - sj : sj - si;
            const uint64_t m2 = ei > ej ? ei - ej : ej - ei, m3 = ci > cj ? ci - cj : cj - ci;
            m = m2 < m ? m2 : m;
            m = m3 < m ? m3 : m;
            const uint32_t smax = spi > spj ? spi : spj, sdif = spi > spj ? spi - spj : spj - spi;
            const double dp = (double)m / p.normalizer;
            const double ds = smax ? (double)sdif / (double)smax : 0.0;
            s_d[sub][q] = dp + ds;
        }
        __syncthreads();
        // each lane of the group owns a contiguous run of pairs, so lane order is pair order
        const uint32_t per = (pairs + GROUP - 1) / GROUP;
        const uint32_t q_lo = min(pairs, sl * per), q_hi = min(pairs, q_lo + per);
        bool active = n >= 2;
        uint32_t merges = 0;
        while (__ballot(active)) {
            // group-wide argmin over the triangle; inactive pairs hold +inf; ties -> smallest pair index
            double bd = inf;
            uint32_t bq = 0xFFFFFFFFu;
            if (active) {
                for (uint32_t q = q_lo; q < q_hi; ++q) {
                    const double v = s_d[sub][q];
                    if (v < bd) { bd = v; bq = q; }
                }
            }
            double md = bd;
#pragma unroll
            for (int off = GROUP / 2; off > 0; off >>= 1) {
                const double od = __shfl_xor(md, off, 64);
                md = od < md ? od : md;
            }
            const bool go = active && md <= p.max_dist;
            const unsigned long long tie = __ballot(go && bd == md) & gmask;
            bq = __shfl(bq, tie ? __ffsll((long long)tie) - 1 : (int)lane, 64);
            uint32_t a = 0, b = 0;
            double na = 0, nb = 0;
            if (go) {
                tri_decode(bq, n, pairs, a, b);
                na = (double)s_size[sub][a];
                nb = (double)s_size[sub][b];
            }
            __syncthreads();
            if (go) {
                for (uint32_t k = sl; k < n; k += GROUP) {
                    if (k == a || k == b) continue;
                    const uint32_t lo_a = k < a ? k : a, hi_a = k < a ? a : k;
                    const uint32_t lo_b = k < b ? k : b, hi_b = k < b ? b : k;
                    const uint32_t qa = tri_row_start(lo_a, n) + (hi_a - lo_a - 1);
                    const uint32_t qb = tri_row_start(lo_b, n) + (hi_b - lo_b - 1);
                    const double da = s_d[sub][qa], db = s_d[sub][qb];
                    if (da == inf) continue;                   // k already merged away
                    s_d[sub][qa] = (na * da + nb * db) / (na + nb);
                    s_d[sub][qb] = inf;
                }
                if (sl == 0) {
                    s_d[sub][bq] = inf;
                    s_size[sub][a] += s_size[sub][b];
                }
                for (uint32_t k = sl; k < n; k += GROUP)
                    if (s_lab[sub][k] == b) s_lab[sub][k] = a;
            }
            __syncthreads();
            ++merges;
            active = go && merges + 1 < n;
        }
        __syncthreads();
        uint32_t roots = 0;
        for (uint32_t k = sl; k < n; k += GROUP) {
            p.label[s + k] = (uint8_t)s_lab[sub][k];
            roots += s_lab[sub][k] == k;
        }
#pragma unroll
        for (int off = GROUP / 2; off > 0; off >>= 1) roots += __shfl_xor(roots, off, 64);
        if (has && sl == 0) p.pc[part] = roots;
    }
}

// Small partitions (n <= NS): one LANE per partition.  Each lane runs the whole agglomeration on its own
// triangle, stored transposed in LDS ([entry][lane]) so that the 64 lanes of a wave hit 64 different banks.
// Same arithmetic, same pair order, same tie-break as the wave-per-partition kernel.
template <int NS>
__global__ __launch_bounds__(64) void cl_cluster_small(const ClParams p, const uint32_t *list, const uint32_t *count)
{
    constexpr int TRI = NS * (NS - 1) / 2;
    __shared__ double s_d[TRI][64];
    __shared__ uint32_t s_pos[NS][64], s_span[NS][64];
    __shared__ uint8_t s_lab[NS][64], s_size[NS][64];
    const uint32_t lane = threadIdx.x;
    const uint32_t L = *count;
    const double inf = __builtin_inf();
    for (uint32_t base = blockIdx.x * 64u; base < L; base += gridDim.x * 64u) {
        const uint32_t li = base + lane;
        if (li >= L) continue;
        const uint32_t part = list[li];
        const uint32_t s = p.part_start[part], n = p.part_start[part + 1] - s;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t a = p.sorted[s + i];
            s_pos[i][lane] = p.pos[a];
            s_span[i][lane] = p.span[a];
            s_lab[i][lane] = (uint8_t)i;
            s_size[i][lane] = 1;
        }
        uint32_t q = 0;
        for (uint32_t i = 0; i + 1 < n; ++i) {
            const uint32_t pi = s_pos[i][lane], spi = s_span[i][lane];
            const uint64_t si = pi, ei = (uint64_t)pi + spi, ci = centre_of(pi, spi);
            for (uint32_t j = i + 1; j < n; ++j, ++q) {
                const uint32_t pj = s_pos[j][lane], spj = s_span[j][lane];
                const uint64_t sj = pj, ej = (uint64_t)pj + spj, cj = centre_of(pj, spj);
                uint64_t m = si > sj ? si - sj : sj - si;
                const uint64_t m2 = ei > ej ? ei - ej : ej - ei, m3 = ci > cj ? ci - cj : cj - ci;
                m = m2 < m ? m2 : m;
                m = m3 < m ? m3 : m;
                const uint32_t smax = spi > spj ? spi : spj, sdif = spi > spj ? spi - spj : spj - spi;
                const double dp = (double)m / p.normalizer;
                const double ds = smax ? (double)sdif / (double)smax : 0.0;
                s_d[q][lane] = dp + ds;
            }
        }
        const uint32_t pairs = q;
        for (uint32_t merges = 0; merges + 1 < n; ++merges) {
            double bd = inf;
            uint32_t ba = 0, bb = 0, bq = 0;
            q = 0;
            for (uint32_t i = 0; i + 1 < n; ++i)
                for (uint32_t j = i + 1; j < n; ++j, ++q) {
                    const double v = s_d[q][lane];
                    if (v < bd) { bd = v; ba = i; bb = j; bq = q; }
                }
            if (!(bd <= p.max_dist)) break;
            const double na = (double)s_size[ba][lane], nb = (double)s_size[bb][lane];
            for (uint32_t k = 0; k < n; ++k) {
                if (k == ba || k == bb) continue;
                const uint32_t lo_a = k < ba ? k : ba, hi_a = k < ba ? ba : k;
                const uint32_t lo_b = k < bb ? k : bb, hi_b = k < bb ? bb : k;
                const uint32_t qa = tri_row_start(lo_a, n) + (hi_a - lo_a - 1);
                const uint32_t qb = tri_row_start(lo_b, n) + (hi_b - lo_b - 1);
                const double da = s_d[qa][lane], db = s_d[qb][lane];
                if (da == inf) continue;
                s_d[qa][lane] = (na * da + nb * db) / (na + nb);
                s_d[qb][lane] = inf;
            }
            s_d[bq][lane] = inf;
            s_size[ba][lane] = (uint8_t)(s_size[ba][lane] + s_size[bb][lane]);
            for (uint32_t k = 0; k < n; ++k)
                if (s_lab[k][lane] == bb) s_lab[k][lane] = (uint8_t)ba;
        }
        (void)pairs;
        uint32_t roots = 0;
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t l = s_lab[k][lane];
            p.label[s + k] = (uint8_t)l;
            roots += l == k;
        }
        p.pc[part] = roots;
    }
}

// one wavefront per partition: clusters by smallest member, members in sorted order
__global__ __launch_bounds__(64) void cl_emit(const ClParams p)
{
    __shared__ uint32_t s_lab[128], s_pos[128], s_span[128];
    const uint32_t lane = threadIdx.x;
    const uint32_t P = *p.n_parts;
    for (uint32_t part = blockIdx.x; part < P; part += gridDim.x) {
        const uint32_t s = p.part_start[part], n = p.part_start[part + 1] - s;
        const uint32_t cb = p.cbase[s];
        __syncthreads();
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t a = p.sorted[s + i];
            s_lab[i] = p.label[s + i];
            s_pos[i] = p.pos[a];
            s_span[i] = p.span[a];
        }
        __syncthreads();
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t my = s_lab[i];
            uint32_t cr = 0, before = 0, same_before = 0, size = 0;
            uint64_t sp = 0, ss = 0;
            for (uint32_t j = 0; j < n; ++j) {
                const uint32_t lj = s_lab[j];
                cr += (lj == j && j < my);
                before += lj < my;
                same_before += (lj == my && j < i);
                if (lj == my) { ++size; sp += s_pos[j]; ss += s_span[j]; }
            }
            const uint32_t a = p.sorted[s + i];
            p.order[s + before + same_before] = a;
            if (my == i) {
                const uint32_t cand = cb + cr;
                p.cand_off[cand + 1] = s + before + size;
                p.cand_contig[cand] = p.contig[a];
                p.cand_type[cand] = p.type[a];
                p.cand_pos[cand] = (uint32_t)(sp / size);
                p.cand_span[cand] = (uint32_t)(ss / size);
            }
        }
        if (part == 0 && lane == 0) p.cand_off[0] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// fused SVIM-mode pipeline: clusters -> the arrays ef_classify reads
// ---------------------------------------------------------------------------------------------

// ctg_off[k] = first candidate whose contig is >= k (candidates are sorted by contig); ctg_off[K] = N
__global__ void sv_contig_offsets(const uint16_t *cand_contig, const uint32_t *n_cands, uint32_t K, uint32_t *ctg_off)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > K) return;
    const uint32_t N = *n_cands;
    uint32_t lo = 0, hi = N;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (cand_contig[mid] < k) lo = mid + 1; else hi = mid;
    }
    ctg_off[k] = k == K ? N : lo;
}

__global__ void sv_adapt_cands(uint32_t N, const uint32_t *cand_off, const uint16_t *cand_contig, const uint32_t *cand_pos,
                               const uint32_t *depth, const uint32_t *depth_off_dev, uint32_t depth_bin,
                               uint32_t *svread, uint32_t *refread, uint8_t *gt_ok)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const uint32_t support = cand_off[c + 1] - cand_off[c];
    const uint32_t k = cand_contig[c];
    const uint32_t nb = depth_off_dev[k + 1] - depth_off_dev[k];
    uint32_t d = 0;
    if (nb) {
        uint32_t bin = cand_pos[c] / depth_bin;
        bin = bin < nb ? bin : nb - 1;
        d = depth[depth_off_dev[k] + bin];
    }
    svread[c] = support;
    refread[c] = d > support ? d - support : 0u;
    gt_ok[c] = 1;
}

__global__ void sv_adapt_marks(uint32_t M, const uint32_t *order, const uint32_t *raw_mark_read, uint32_t *mark_read)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < M) mark_read[m] = raw_mark_read[order[m]];
}

uint32_t bits_for(uint64_t max_value)
{
    uint32_t b = 0;
    while (b < 64 && (max_value >> b)) ++b;
    return b ? b : 1;
}

template <int OP>
void launch_scan(const uint32_t *in, uint32_t n, uint32_t *part, uint32_t *out, uint32_t *total, hipStream_t st)
{
    const uint32_t nb = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(scan_reduce<OP>, dim3(nb), dim3(kScanThreads), 0, st, in, n, part);
    hipLaunchKernelGGL(scan_spine<OP>, dim3(1), dim3(1024), 0, st, part, nb, total);
    hipLaunchKernelGGL(scan_apply<OP>, dim3(nb), dim3(kScanThreads), 0, st, in, n, (const uint32_t *)part, out);
}

}  // namespace

extern "C" {

int duet_cluster_run_device(duet_ctx *ctx, const duet_cluster_problem *pr, const duet_cluster_result *res, void *stream_)
{
    if (!ctx) return duet_fail(nullptr, DUET_ERR_INVALID, "null context");
    if (!pr || !res) return duet_fail(ctx, DUET_ERR_INVALID, "null argument");
    if (pr->part_max < 1 || pr->part_max > 128) return duet_fail(ctx, DUET_ERR_INVALID, "part_max must be in 1..128");
    if (!(pr->normalizer > 0)) return duet_fail(ctx, DUET_ERR_INVALID, "normalizer must be positive");
    if (!res->n_cands) return duet_fail(ctx, DUET_ERR_INVALID, "null n_cands");
    hipStream_t st = (hipStream_t)stream_;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t M = pr->n_marks;
    if (M == 0) {
        HIP_TRY(ctx, hipMemsetAsync(res->n_cands, 0, 4, st));
        return DUET_OK;
    }
    if (!pr->mark_contig || !pr->mark_type || !pr->mark_pos || !pr->mark_span || !res->order || !res->cand_off ||
        !res->cand_contig || !res->cand_type || !res->cand_pos || !res->cand_span)
        return duet_fail(ctx, DUET_ERR_INVALID, "null array");

    const uint32_t nb_rx = (M + kRxTile - 1) / kRxTile;
    const uint32_t nb_sc = (M + kScanTile - 1) / kScanTile;
    const uint32_t nb_hs = (256u * nb_rx + kScanTile - 1) / kScanTile;      // scan tiles of the radix histogram
    const size_t sizes[14] = {(size_t)M * 8, (size_t)M * 8, (size_t)M * 4, (size_t)M * 4, (size_t)256 * nb_rx * 4,
                              ((size_t)M + 1) * 4, ((size_t)M + 1) * 4, ((size_t)(nb_sc > nb_hs ? nb_sc : nb_hs) + 1) * 4,
                              ((size_t)M + 1) * 4, (size_t)M, (size_t)M * 4, 64, (size_t)M * 4 * kClasses, 16};
    int rc;
    for (int i = 0; i < 14; ++i)
        if ((rc = duet_reserve(ctx, ctx->cl_ws[i], sizes[i]))) return rc;
    uint64_t *keysA = (uint64_t *)ctx->cl_ws[0].ptr, *keysB = (uint64_t *)ctx->cl_ws[1].ptr;
    uint32_t *valsA = (uint32_t *)ctx->cl_ws[2].ptr, *valsB = (uint32_t *)ctx->cl_ws[3].ptr;
    uint32_t *hist = (uint32_t *)ctx->cl_ws[4].ptr;
    uint32_t *tmpA = (uint32_t *)ctx->cl_ws[5].ptr, *tmpB = (uint32_t *)ctx->cl_ws[6].ptr;
    uint32_t *spart = (uint32_t *)ctx->cl_ws[7].ptr;
    uint32_t *part_start = (uint32_t *)ctx->cl_ws[8].ptr;
    uint8_t *label = (uint8_t *)ctx->cl_ws[9].ptr;
    uint32_t *pc = (uint32_t *)ctx->cl_ws[10].ptr;
    uint32_t *scal = (uint32_t *)ctx->cl_ws[11].ptr;      // [0] = n_parts

    ClParams p;
    memset(&p, 0, sizeof(p));
    p.M = M; p.part_gap = pr->part_gap; p.part_max = pr->part_max;
    p.max_dist = pr->max_dist; p.normalizer = pr->normalizer;
    const uint64_t max_centre = pr->max_pos_hint ? (uint64_t)pr->max_pos_hint + (pr->max_span_hint ? pr->max_span_hint : 0xFFFFFFFFull) / 2
                                                 : 0x17FFFFFFFull;
    p.centre_bits = bits_for(max_centre);
    p.type_bits = bits_for(pr->n_types_hint ? pr->n_types_hint - 1 : 255);
    const uint32_t contig_bits = bits_for(pr->n_contigs_hint ? pr->n_contigs_hint - 1 : 65535);
    const uint32_t key_bits = p.centre_bits + p.type_bits + contig_bits;
    if (key_bits > 64) return duet_fail(ctx, DUET_ERR_INVALID, "sort key does not fit 64 bits");
    p.contig = pr->mark_contig; p.type = pr->mark_type; p.pos = pr->mark_pos; p.span = pr->mark_span;

    const dim3 g256((M + 255) / 256), b256(256);
    hipLaunchKernelGGL(cl_keys, g256, b256, 0, st, p, keysA, valsA);
    uint64_t *kin = keysA, *kout = keysB;
    uint32_t *vin = valsA, *vout = valsB;
    const uint32_t nh = 256 * nb_rx;
    for (uint32_t shift = 0; shift < key_bits; shift += 8) {
        hipLaunchKernelGGL(rx_hist, dim3(nb_rx), dim3(kRxThreads), 0, st, (const uint64_t *)kin, M, shift, nb_rx, hist);
        launch_scan<0>(hist, nh, spart, hist, nullptr, st);      // in place: scan_apply reads a tile before writing it
        hipLaunchKernelGGL(rx_scatter, dim3(nb_rx), dim3(kRxThreads), 0, st, (const uint64_t *)kin,
                           (const uint32_t *)vin, M, shift, nb_rx, (const uint32_t *)hist, kout, vout);
        uint64_t *tk = kin; kin = kout; kout = tk;
        uint32_t *tv = vin; vin = vout; vout = tv;
    }
    p.sorted = vin;
    hipLaunchKernelGGL(cl_heads1, g256, b256, 0, st, p, (const uint64_t *)kin, tmpA);
    launch_scan<1>(tmpA, M, spart, tmpB, nullptr, st);            // tmpB[i] = start of i's natural partition
    hipLaunchKernelGGL(cl_heads2, g256, b256, 0, st, p, (const uint32_t *)tmpB, tmpA);     // tmpA = head flags
    launch_scan<0>(tmpA, M, spart, tmpB, scal, st);               // tmpB = partition id, scal[0] = #partitions
    hipLaunchKernelGGL(cl_parts, g256, b256, 0, st, p, (const uint32_t *)tmpA, (const uint32_t *)tmpB, part_start,
                       (const uint32_t *)scal);
    p.part_start = part_start; p.n_parts = scal; p.label = label; p.pc = pc;
    const uint32_t grid = M < 16384u ? M : 16384u;               // partitions <= marks; kernels stride over them
    uint32_t *lists = (uint32_t *)ctx->cl_ws[12].ptr;            // [kClasses][M]
    uint32_t *cnts = scal + 2;
    HIP_TRY(ctx, hipMemsetAsync(cnts, 0, 4 * kClasses, st));
    hipLaunchKernelGGL(cl_classes, g256, b256, 0, st, p, lists, cnts);
    // the few large partitions (49..128 marks) take long, serial agglomerations: run them on a side stream
    // beside the bulk
    HIP_TRY(ctx, hipEventRecord(ctx->cl_fork, st));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->cl_side, ctx->cl_fork, 0));
    hipLaunchKernelGGL((cl_cluster<64, 128>), dim3(grid < 1024u ? grid : 1024u), dim3(64), 0, ctx->cl_side, p,
                       (const uint32_t *)(lists + 4 * (size_t)M), (const uint32_t *)(cnts + 4));
    HIP_TRY(ctx, hipEventRecord(ctx->cl_join, ctx->cl_side));
    const uint32_t g_lane = (M + 63) / 64 < 4096u ? (M + 63) / 64 : 4096u;
    hipLaunchKernelGGL((cl_cluster_small<8>), dim3(g_lane), dim3(64), 0, st, p, (const uint32_t *)lists,
                       (const uint32_t *)(cnts + 0));
    hipLaunchKernelGGL((cl_cluster<16, 16>), dim3(grid), dim3(64), 0, st, p, (const uint32_t *)(lists + 1 * (size_t)M),
                       (const uint32_t *)(cnts + 1));
    hipLaunchKernelGGL((cl_cluster<32, 32>), dim3(grid), dim3(64), 0, st, p, (const uint32_t *)(lists + 2 * (size_t)M),
                       (const uint32_t *)(cnts + 2));
    hipLaunchKernelGGL((cl_cluster<64, 48>), dim3(grid), dim3(64), 0, st, p, (const uint32_t *)(lists + 3 * (size_t)M),
                       (const uint32_t *)(cnts + 3));
    HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->cl_join, 0));
    // clusters per partition -> candidate bases.  The partition count lives on the device, so the counts are
    // spread to the partitions' start positions (zero elsewhere) and scanned over the M sorted positions.
    hipLaunchKernelGGL(cl_pcat, g256, b256, 0, st, p, (const uint32_t *)tmpA, (const uint32_t *)tmpB, tmpA);
    launch_scan<0>(tmpA, M, spart, tmpB, res->n_cands, st);       // tmpB[s] = first candidate of the partition at s
    p.cbase = tmpB;
    p.order = res->order; p.cand_off = res->cand_off; p.cand_pos = res->cand_pos; p.cand_span = res->cand_span;
    p.cand_contig = res->cand_contig; p.cand_type = res->cand_type;
    hipLaunchKernelGGL(cl_emit, dim3(grid), dim3(64), 0, st, p);
    HIP_TRY(ctx, hipGetLastError());
    return DUET_OK;
}

int duet_cluster_run_host(duet_ctx *ctx, const duet_cluster_problem *pr, const duet_cluster_result *res)
{
    if (!ctx) return duet_fail(nullptr, DUET_ERR_INVALID, "null context");
    if (!pr || !res || !res->n_cands) return duet_fail(ctx, DUET_ERR_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t M = pr->n_marks;
    if (M == 0) {
        *res->n_cands = 0;
        return DUET_OK;
    }
    hipStream_t s = ctx->own_stream;
    int rc;
    const void *src[4] = {pr->mark_contig, pr->mark_type, pr->mark_pos, pr->mark_span};
    const size_t ib[4] = {(size_t)M * 2, (size_t)M, (size_t)M * 4, (size_t)M * 4};
    for (int i = 0; i < 4; ++i) {
        if (!src[i]) return duet_fail(ctx, DUET_ERR_INVALID, "null array");
        if ((rc = duet_reserve(ctx, ctx->cl_in[i], ib[i]))) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->cl_in[i].ptr, src[i], ib[i], hipMemcpyHostToDevice, s));
    }
    const size_t ob[6] = {(size_t)M * 4, ((size_t)M + 1) * 4 + 16, (size_t)M * 2, (size_t)M, (size_t)M * 4, (size_t)M * 4};
    for (int i = 0; i < 6; ++i)
        if ((rc = duet_reserve(ctx, ctx->cl_out[i], ob[i]))) return rc;
    duet_cluster_problem d = *pr;
    d.mark_contig = (const uint16_t *)ctx->cl_in[0].ptr;
    d.mark_type = (const uint8_t *)ctx->cl_in[1].ptr;
    d.mark_pos = (const uint32_t *)ctx->cl_in[2].ptr;
    d.mark_span = (const uint32_t *)ctx->cl_in[3].ptr;
    duet_cluster_result r;
    r.order = (uint32_t *)ctx->cl_out[0].ptr;
    r.cand_off = (uint32_t *)ctx->cl_out[1].ptr;
    r.cand_contig = (uint16_t *)ctx->cl_out[2].ptr;
    r.cand_type = (uint8_t *)ctx->cl_out[3].ptr;
    r.cand_pos = (uint32_t *)ctx->cl_out[4].ptr;
    r.cand_span = (uint32_t *)ctx->cl_out[5].ptr;
    r.n_cands = (uint32_t *)((char *)ctx->cl_out[1].ptr + ((size_t)M + 1) * 4);        // spare word after cand_off
    if ((rc = duet_cluster_run_device(ctx, &d, &r, s))) return rc;
    uint32_t n = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n, r.n_cands, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    *res->n_cands = n;
    HIP_TRY(ctx, hipMemcpy(res->order, r.order, (size_t)M * 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(res->cand_off, r.cand_off, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(res->cand_contig, r.cand_contig, (size_t)n * 2, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(res->cand_type, r.cand_type, (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(res->cand_pos, r.cand_pos, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(res->cand_span, r.cand_span, (size_t)n * 4, hipMemcpyDeviceToHost));
    return DUET_OK;
}

int duet_svim_phase_device(duet_ctx *ctx, const duet_svim_problem *pr, const duet_cluster_result *res, uint8_t *out_pred,
                           uint32_t *out_ps, uint32_t *n_cands_host, void *stream_)
{
    if (!ctx) return duet_fail(nullptr, DUET_ERR_INVALID, "null context");
    if (!pr || !res || !out_pred || !out_ps || !n_cands_host) return duet_fail(ctx, DUET_ERR_INVALID, "null argument");
    if (!pr->depth_off || pr->depth_bin == 0 || pr->n_contigs == 0 || pr->n_contigs > 65535)
        return duet_fail(ctx, DUET_ERR_INVALID, "bad depth / contig description");
    hipStream_t st = (hipStream_t)stream_;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t M = pr->marks.n_marks, K = pr->n_contigs;
    *n_cands_host = 0;
    if (M == 0) return DUET_OK;
    int rc = duet_cluster_run_device(ctx, &pr->marks, res, st);
    if (rc) return rc;
    // workspace: ctg_off (device + host), adapted candidate columns, gathered marks, device depth_off
    const size_t sz[6] = {((size_t)K + 1) * 4 * 2, (size_t)M * 4, (size_t)M * 4, (size_t)M, (size_t)M * 4, 0};
    for (int i = 0; i < 5; ++i)
        if ((rc = duet_reserve(ctx, ctx->sv_ws[i], sz[i]))) return rc;
    uint32_t *d_ctg_off = (uint32_t *)ctx->sv_ws[0].ptr, *d_depth_off = d_ctg_off + (K + 1);
    uint32_t *d_svread = (uint32_t *)ctx->sv_ws[1].ptr, *d_refread = (uint32_t *)ctx->sv_ws[2].ptr;
    uint8_t *d_gt = (uint8_t *)ctx->sv_ws[3].ptr;
    uint32_t *d_mark = (uint32_t *)ctx->sv_ws[4].ptr;
    HIP_TRY(ctx, hipMemcpyAsync(d_depth_off, pr->depth_off, ((size_t)K + 1) * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(sv_contig_offsets, dim3((K + 1 + 255) / 256), dim3(256), 0, st, (const uint16_t *)res->cand_contig,
                       (const uint32_t *)res->n_cands, K, d_ctg_off);
    std::vector<uint32_t> ctg_off(K + 1);
    HIP_TRY(ctx, hipMemcpyAsync(ctg_off.data(), d_ctg_off, ((size_t)K + 1) * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));             // the one host round trip: candidates per contig
    const uint32_t N = ctg_off[K];
    *n_cands_host = N;
    if (N == 0) return DUET_OK;
    hipLaunchKernelGGL(sv_adapt_cands, dim3((N + 255) / 256), dim3(256), 0, st, N, (const uint32_t *)res->cand_off,
                       (const uint16_t *)res->cand_contig, (const uint32_t *)res->cand_pos, pr->depth,
                       (const uint32_t *)d_depth_off, pr->depth_bin, d_svread, d_refread, d_gt);
    hipLaunchKernelGGL(sv_adapt_marks, dim3((M + 255) / 256), dim3(256), 0, st, M, (const uint32_t *)res->order,
                       pr->mark_read, d_mark);
    HIP_TRY(ctx, hipGetLastError());
    duet_ef_problem ef;
    memset(&ef, 0, sizeof(ef));
    ef.n_contigs = K; ef.n_cands = N; ef.n_marks = M; ef.n_reads = pr->n_reads;
    ef.cand_ctg_off = ctg_off.data();
    ef.read_tag = pr->read_tag;
    ef.cand_pos = res->cand_pos; ef.cand_svlen = res->cand_span; ef.cand_svread = d_svread; ef.cand_refread = d_refread;
    ef.cand_gt_ok = d_gt; ef.cand_off = res->cand_off; ef.mark_read = d_mark;
    ef.svlen_thres = pr->svlen_thres; ef.suppread_thres = pr->suppread_thres;
    return duet_ef_run_device(ctx, &ef, out_pred, out_ps, st);
}

}  // extern "C"
