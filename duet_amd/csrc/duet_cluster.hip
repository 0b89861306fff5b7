// duet_cluster.hip -- gfx950 kernels and C ABI for stage A0: span-position clustering of SV marks into
// candidates (what `--cluster_max_distance` controls).
//
// The reference delegates this stage to the external `svim alignment` binary (src/duet/sv_calling.py:13-15),
// so there is no reference code to follow; the rule implemented here is this repository's own deterministic
// statement of the published SVIM 1.4.2 scheme, normative text in oracle/cluster_oracle.c / DESIGN.md section 9.
//
// Pipeline:
//   cl_keys        key = (contig, type, centre = pos + span/2) packed into the fewest bits, the mark index in the spare
//                  bits above them when it fits (else a separate value array)
//   radix sort     stable LSD, 8-bit digits: rx_hist -> tile offsets -> rx_scatter per pass (ballot-ranked, no atomics
//                  on the data path, so the order is deterministic); keys only when the index rides in the key
//   partitions     one scan over a composite element straight off the sorted keys: natural partition starts (contig/type
//                  change or centre gap > part_gap), a partition there and every part_max marks after -> partition ids
//                  and the partition start list
//   cl_classes     work lists by partition size (<= 8 / 16 / 32 / 64 / 128 marks)
//   fast pass      per size class, GROUP lanes per partition: clusters read off the threshold graph where that is
//                  provably what average linkage produces (see the comment above fast_unit); what it cannot settle
//                  goes, component by component, on work lists
//   exact pass     binary64 average linkage for the listed components (nearest-neighbour cache per row)
//   rank pass      finishes the partitions that had listed components
//                  fast and rank passes write each mark to its place in the partition's output (order[], and in the
//                  fused pipeline its read index) and leave, per cluster, a record (rank, end, floor means) at the
//                  partition's start + the cluster's index
//   scan + cl_emit clusters per partition -> candidate bases; one thread per partition turns its clusters' records into cand_*[]
//
// Bit-exactness vs the oracle: what is emitted depends only on the final clusters; the exact pass evaluates the same
// binary64 expressions in the same order as the oracle (-ffp-contract=off; ties to the smallest (first, second)
// index pair), and the fast pass only accepts clusters it can prove (guard bands wider than any rounding error),
// so both produce the oracle's clusters.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <type_traits>
#include <vector>

#include "duet_ef.h"
#include "duet_internal.h"

#pragma clang fp contract(off)

namespace {

#include "duet_prims.hip.h"


struct ClParams {
    uint32_t M;
    uint32_t part_gap, part_max;
    double max_dist, normalizer;
    uint32_t centre_bits, type_bits;                  // key = ((contig << type_bits | type) << centre_bits) | centre
    uint32_t key_bits, idx_packed;                    // idx_packed: the mark index rides in the key's spare bits above key_bits (the
                                                      // sort moves keys only); otherwise it is the sort's value array `sorted`
    const uint16_t *contig;
    const uint8_t *type;
    const uint32_t *pos, *span;
    const uint32_t *sorted;                           // mark index at each sorted position
    const uint2 *ps;                                  // (pos, span) per mark, side by side: one gather instead of two
    const uint4 *rec4;                                // fused pipeline instead: (pos, span, read index, -) -- the agglomeration's gather
                                                      // also fetches what the output needs, cl_emit does not gather again
    const uint64_t *skeys;                            // the sorted keys (contig | type | centre)
    const uint32_t *part_start;                       // [P+1]
    const uint32_t *n_parts;                          // device scalar
    float inv_norm, t_lo[3], t_hi[3];                 // fast pass: 1/normalizer; max_dist / {1, 2, 4} * (1 -/+ 1e-5) in binary32
    uint32_t fast;                                    // 0: parameters outside the fast pass's vetted range, everything goes to the exact pass
    // fused SVIM-mode pipeline (all null otherwise): cl_emit also writes the columns ef_classify reads
    const uint32_t *sv_mark_in, *sv_depth, *sv_depth_off;
    uint32_t sv_depth_bin;
    uint32_t *sv_mark_out, *sv_svread, *sv_refread;
    uint8_t *sv_gt;
    uint8_t *label8;                                  // [M] per sorted position: its cluster's smallest member (row inside the partition)
    uint8_t *comp8;                                   // [M] rows left to the exact pass: smallest row of their component; else 0xFF
    uint4 *e_rec;                                     // [M] cluster c of the partition that starts at s, at s + c: (rank | end << 8, floor mean pos, floor mean span, -)
    uint32_t *pc;                                     // [P] clusters per partition
    const uint32_t *cbase;                            // [P] first candidate of each partition
    // outputs
    uint32_t *order, *cand_off, *cand_pos, *cand_span;
    uint16_t *cand_contig;
    uint8_t *cand_type;
};

__device__ __forceinline__ uint64_t centre_of(uint32_t pos, uint32_t span) { return (uint64_t)pos + (span >> 1); }
__host__ __device__ __forceinline__ uint64_t key_mask(uint32_t key_bits) { return key_bits >= 64u ? ~0ull : (1ull << key_bits) - 1ull; }

// ---------------------------------------------------------------------------------------------
// keys + radix sort
// ---------------------------------------------------------------------------------------------

// without the caller's hints: the largest contig, type and centre, so that the sort key uses only the bits it needs
__global__ __launch_bounds__(256) void cl_maxima(const ClParams p, uint32_t *out /* [0] contig, [1] type, [2..3] centre (u64) */)
{
    uint32_t mc = 0, mt = 0;
    uint64_t mx = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < p.M; i += gridDim.x * blockDim.x) {
        mc = max(mc, (uint32_t)p.contig[i]);
        mt = max(mt, (uint32_t)p.type[i]);
        const uint64_t c = centre_of(p.pos[i], p.span[i]);
        mx = c > mx ? c : mx;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        mc = max(mc, (uint32_t)__shfl_xor((int)mc, d, 64));
        mt = max(mt, (uint32_t)__shfl_xor((int)mt, d, 64));
        const uint64_t o = ((uint64_t)(uint32_t)__shfl_xor((int)(mx >> 32), d, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)mx, d, 64);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63u) == 0) {
        atomicMax(&out[0], mc);
        atomicMax(&out[1], mt);
        atomicMax((unsigned long long *)(out + 2), (unsigned long long)mx);
    }
}

// sort keys, the marks' records -- and, since the keys are at hand, the sort's first digit histogram (what rx_hist would
// read them back for): one workgroup per radix tile
__global__ __launch_bounds__(kRxHistThreads) void cl_keys(const ClParams p, uint64_t *keys, uint32_t *vals, uint2 *ps, uint4 *rec4, uint32_t dmask,
                                                          uint32_t nb, uint32_t *hist /* [256][nb] */, uint32_t *dtot /* or null */)
{
    __shared__ uint32_t s_h[256];
    const uint32_t tid = threadIdx.x;
    if (tid < 256) s_h[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kRxTile / kRxHistThreads; ++it) {
        const uint32_t i = blockIdx.x * kRxTile + it * kRxHistThreads + tid;
        if (i < p.M) {
            const uint64_t hi = ((uint64_t)p.contig[i] << p.type_bits) | (uint64_t)p.type[i];
            const uint32_t ps_ = p.pos[i], sp_ = p.span[i];
            const uint64_t key = (hi << p.centre_bits) | centre_of(ps_, sp_);
            if (p.sv_mark_in) rec4[i] = make_uint4(ps_, sp_, p.sv_mark_in[i], 0u);
            else ps[i] = make_uint2(ps_, sp_);
            if (p.idx_packed) keys[i] = key | ((uint64_t)i << p.key_bits);
            else { keys[i] = key; vals[i] = i; }
            atomicAdd(&s_h[(uint32_t)key & dmask], 1u);
        }
    }
    __syncthreads();
    if (tid < 256) {
        const uint32_t c = s_h[tid];
        hist[(size_t)tid * nb + blockIdx.x] = c;
        if (dtot && c) atomicAdd(&dtot[(blockIdx.x % kDtotCopies) * 256u + tid], c);
    }
}

// element sources / sinks of the scans: what used to be separate elementwise kernels rides on the scan's own
// loads and stores
// i if sorted position i starts a natural partition (contig/type change or centre gap), else 0: everything needed
// is in the sorted keys -- (contig, type) in the high bits, the centre in the low bits
struct LoadHead {
    const uint64_t *keys;
    uint32_t centre_bits, part_gap;
    uint64_t km;                                            // the key proper (a packed mark index sits above it)
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const
    {
        if (i == 0) return 0u;
        const uint64_t a = keys[i - 1] & km, b = keys[i] & km;
        const uint64_t cm = (1ull << centre_bits) - 1ull;
        const bool cut = (a >> centre_bits) != (b >> centre_bits) || (b & cm) - (a & cm) > (uint64_t)part_gap;
        return cut ? i : 0u;
    }
};
// the same for the kScanItems consecutive positions base .. base + kScanItems - 1 of one scan thread, from 16-byte loads (base is a
// multiple of kScanItems and the key buffer is hipMalloc-aligned): head[j] = position base + j starts a natural partition
__device__ __forceinline__ void load_heads(const LoadHead &h, uint32_t base, uint32_t n, bool (&head)[kScanItems])
{
    static_assert(kScanItems % 2 == 0, "two keys per load");
    uint64_t k[kScanItems + 1];
    k[0] = base > 0 && base <= n ? h.keys[base - 1] & h.km : 0ull;
#pragma unroll
    for (int j = 0; j < kScanItems; j += 2) {
        if (base + j + 1 < n) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(h.keys + base + j);
            k[j + 1] = v.x & h.km;
            k[j + 2] = v.y & h.km;
        } else {
            k[j + 1] = base + j < n ? h.keys[base + j] & h.km : 0ull;
            k[j + 2] = 0ull;
        }
    }
    const uint64_t cm = (1ull << h.centre_bits) - 1ull;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        const uint32_t i = base + j;
        const uint64_t a = k[j], b = k[j + 1];
        head[j] = i < n && (i == 0 || (a >> h.centre_bits) != (b >> h.centre_bits) || (b & cm) - (a & cm) > (uint64_t)h.part_gap);
    }
}

// clusters of partition i.  The partition count lives on the device: the scan is launched over the M positions (an upper
// bound) and the elements past the count neither load nor store anything
struct LoadPc {
    const uint32_t *pc, *n_parts;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return i < *n_parts ? pc[i] : 0u; }
};
struct StorePc {
    uint32_t *out;
    const uint32_t *n_parts;
    __device__ __forceinline__ void operator()(uint32_t i, uint32_t v, uint32_t) const { if (i < *n_parts) out[i] = v; }
};
// ---------------------------------------------------------------------------------------------
// partitions: ONE scan over a composite element instead of a max-scan + a sum-scan
// ---------------------------------------------------------------------------------------------
//
// Position i starts a NATURAL partition when it is position 0 or LoadHead says so; a partition starts there and every
// part_max marks after.  With H_i = start of i's natural partition and P_i = partitions of the natural partitions that end
// before H_i:   partition of i = P_i + (i - H_i) / part_max,   i starts one <=> (i - H_i) % part_max == 0.
// (H, P) over a range of positions is summarised by (f, l, s): first and last natural start inside it and the partitions
// of the natural partitions that lie between two starts of the range; ranges combine associatively:
//     (f1, l1, s1) + (f2, l2, s2) = (f1, l2, s1 + s2 + ceil((f2 - l1) / part_max)).
constexpr uint32_t kNoHead = 0xFFFFFFFFu;
struct PartSum {
    uint32_t f, l, s;
};
__device__ __forceinline__ uint32_t parts_in(uint32_t len, uint32_t pm) { return len <= pm ? 1u : (len + pm - 1u) / pm; }
__device__ __forceinline__ PartSum part_combine(const PartSum &a, const PartSum &b, uint32_t pm)
{
    if (b.f == kNoHead) return a;
    if (a.f == kNoHead) return b;
    return PartSum{a.f, b.l, a.s + b.s + parts_in(b.f - a.l, pm)};
}
__device__ __forceinline__ PartSum part_shfl_up(const PartSum &v, int d)
{
    return PartSum{(uint32_t)__shfl_up((int)v.f, d, 64), (uint32_t)__shfl_up((int)v.l, d, 64), (uint32_t)__shfl_up((int)v.s, d, 64)};
}
// scan over the NT threads of a block; returns the thread's EXCLUSIVE prefix (everything before it in the block) and leaves
// the block total in s_w[NT / 64]
template <int NT>
__device__ __forceinline__ PartSum part_block_exscan(const PartSum &mine, uint32_t pm, PartSum *s_w /* [NT / 64 + 1] */)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    PartSum x = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const PartSum y = part_shfl_up(x, d);
        if ((int)lane >= d) x = part_combine(y, x, pm);
    }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    PartSum before{kNoHead, 0, 0};
    for (uint32_t w = 0; w < wave; ++w) before = part_combine(before, s_w[w], pm);
    const PartSum prev = part_shfl_up(x, 1);
    if (lane > 0) before = part_combine(before, prev, pm);
    if (tid == NT - 1) s_w[NT / 64] = part_combine(before, mine, pm);
    return before;
}

__global__ __launch_bounds__(kScanThreads) void part_reduce(const LoadHead in, uint32_t n, uint32_t pm, PartSum *tiles, uint32_t *zero14)
{
    __shared__ PartSum s_w[kScanThreads / 64 + 1];
    const uint32_t tid = threadIdx.x;
    if (zero14 && blockIdx.x == 0 && tid < 14) zero14[tid] = 0;         // the work-list counters of the kernels that follow
    const uint32_t base = blockIdx.x * kScanTile + tid * kScanItems;
    PartSum acc{kNoHead, 0, 0};
    bool head[kScanItems];
    load_heads(in, base, n, head);
#pragma unroll
    for (int j = 0; j < kScanItems; ++j)
        if (head[j]) acc = part_combine(acc, PartSum{base + j, base + j, 0}, pm);
    (void)part_block_exscan<kScanThreads>(acc, pm, s_w);
    __syncthreads();
    if (tid == 0) tiles[blockIdx.x] = s_w[kScanThreads / 64];
}

// more tiles than a block wants to combine by itself: exclusive scan of the tile summaries by ONE block -- every thread
// folds a contiguous run of them, one block scan over the threads, every thread writes its run's prefixes
__global__ __launch_bounds__(1024) void part_spine(PartSum *tiles, uint32_t nb, uint32_t pm)
{
    __shared__ PartSum s_w[1024 / 64 + 1];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (nb + 1023u) / 1024u, lo = min(nb, tid * per), hi = min(nb, lo + per);
    // a thread's summaries go through registers sixteen at a time: their loads leave together (one round trip per sixteen
    // instead of one per summary -- 2 x 10 dependent trips at 2e7 marks)
    constexpr uint32_t kHold = 16;
    PartSum acc{kNoHead, 0, 0};
    for (uint32_t t0 = lo; t0 < hi; t0 += kHold) {
        PartSum mine[kHold];
#pragma unroll
        for (uint32_t j = 0; j < kHold; ++j) mine[j] = tiles[min(t0 + j, hi - 1u)];
#pragma unroll
        for (uint32_t j = 0; j < kHold; ++j)
            if (t0 + j < hi) acc = part_combine(acc, mine[j], pm);
    }
    PartSum run = part_block_exscan<1024>(acc, pm, s_w);
    for (uint32_t t0 = lo; t0 < hi; t0 += kHold) {
        PartSum mine[kHold];
#pragma unroll
        for (uint32_t j = 0; j < kHold; ++j) mine[j] = tiles[min(t0 + j, hi - 1u)];
#pragma unroll
        for (uint32_t j = 0; j < kHold; ++j) {
            if (t0 + j < hi) {
                tiles[t0 + j] = run;
                run = part_combine(run, mine[j], pm);
            }
        }
    }
}

// SELF: tiles[] holds the tiles' own summaries and every block combines the ones before it by itself.  pid (optional): every
// position's partition id -- nobody downstream needs it since cl_emit walks partitions
template <bool SELF>
__global__ __launch_bounds__(kScanThreads) void part_apply(const LoadHead in, uint32_t n, uint32_t pm, const PartSum *tiles, uint32_t *pid,
                                                           uint32_t *part_start, uint32_t *n_parts)
{
    __shared__ PartSum s_w[kScanThreads / 64 + 1];
    __shared__ PartSum s_c[kScanThreads / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t base = blockIdx.x * kScanTile + tid * kScanItems;
    bool head[kScanItems];
    PartSum acc{kNoHead, 0, 0};
    load_heads(in, base, n, head);
#pragma unroll
    for (int j = 0; j < kScanItems; ++j)
        if (head[j]) acc = part_combine(acc, PartSum{base + j, base + j, 0}, pm);
    PartSum carry{kNoHead, 0, 0};
    if (SELF) {
        // the tiles before this one, combined in order: thread t takes a contiguous run of them
        const uint32_t nbef = blockIdx.x, per = (nbef + kScanThreads - 1) / kScanThreads;
        const uint32_t lo = min(nbef, tid * per), hi = min(nbef, lo + per);
        PartSum c{kNoHead, 0, 0};
        for (uint32_t t = lo; t < hi; ++t) c = part_combine(c, tiles[t], pm);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const PartSum y = part_shfl_up(c, d);
            if ((int)lane >= d) c = part_combine(y, c, pm);
        }
        if (lane == 63) s_c[wave] = c;
    } else {
        carry = tiles[blockIdx.x];
    }
    const PartSum before = part_block_exscan<kScanThreads>(acc, pm, s_w);       // (synchronises: s_c is visible after it)
    if (SELF) {
#pragma unroll
        for (int w = 0; w < kScanThreads / 64; ++w) carry = part_combine(carry, s_c[w], pm);
    }
    const PartSum st = part_combine(carry, before, pm);
    // the state in front of this thread's first element: its natural partition's start H and the partitions before H.
    // (st.s counts the natural partitions between the starts of everything before; position 0 is itself a start, so nothing
    // lies in front of the first one)
    uint32_t H = st.f == kNoHead ? 0u : st.l, P = st.f == kNoHead ? 0u : st.s;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        const uint32_t i = base + j;
        if (i >= n) break;
        if (head[j]) {
            if (i) P += parts_in(i - H, pm);
            H = i;
        }
        const uint32_t d = i - H, q = d < pm ? 0u : d / pm;
        const bool starts = d == q * pm;
        if (pid) pid[i] = P + q;
        if (starts) part_start[P + q] = i;
        if (i == n - 1) {
            part_start[P + q + 1] = n;
            *n_parts = P + q + 1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// partitions
// ---------------------------------------------------------------------------------------------

// work lists by partition size (which agglomeration kernel variant takes it); list order is irrelevant --
// every partition writes to its own fixed output range -- so a (wave-aggregated) atomic append is fine
constexpr int kClasses = 5;
__device__ __forceinline__ int size_class(uint32_t n) { return n <= 8 ? 0 : (n <= 16 ? 1 : (n <= 32 ? 2 : (n <= 64 ? 3 : 4))); }

__global__ __launch_bounds__(1024) void cl_classes(const ClParams p, uint32_t *lists /* [kClasses][M] */, uint32_t *counts /* [kClasses] */)
{
    __shared__ uint32_t s_cnt[kClasses], s_base[kClasses];
    // (the partitions' number lives on the device: the grid is a fraction of its bound -- the marks -- and strides)
    const uint32_t n_parts = *p.n_parts;
    for (uint32_t tile = blockIdx.x; tile * blockDim.x < n_parts; tile += gridDim.x) {
        if (threadIdx.x < kClasses) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t part = tile * blockDim.x + threadIdx.x;
        const bool live = part < n_parts;
        const int cls = live ? size_class(p.part_start[part + 1] - p.part_start[part]) : -1;
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t at = 0;
#pragma unroll
        for (int c = 0; c < kClasses; ++c) {
            const unsigned long long m = __ballot(cls == c);
            if (!m) continue;
            uint32_t base = 0;
            if (lane == (uint32_t)__ffsll((long long)m) - 1u) base = atomicAdd(&s_cnt[c], (uint32_t)__popcll(m));
            base = __shfl(base, __ffsll((long long)m) - 1, 64);
            if (cls == c) at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        }
        __syncthreads();
        if (threadIdx.x < kClasses && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
        __syncthreads();
        if (cls >= 0) lists[(size_t)cls * p.M + s_base[cls] + at] = part;
        __syncthreads();                                     // s_cnt / s_base are reused
    }
}

// ---------------------------------------------------------------------------------------------
// agglomeration
// ---------------------------------------------------------------------------------------------
//
// GROUP lanes of a wavefront work on one component of up to GROUP * R marks; lane sl owns the rows
// sl, sl + GROUP, ... of the component's distance matrix, whose upper triangle lives in LDS.  A wave carries
// 64 / GROUP components in lockstep; control flow is wave-uniform and groups that have nothing to do in a phase
// are predicated off.
//
// The oracle's merge step is "global argmin over the active upper triangle, ties to the smallest (row, col)".
// Here every row caches (rmin, rarg) = its minimum over the active columns to its right and the smallest
// column that attains it, so the global argmin is a butterfly over one value per row.  Invariant:
//     rmin <= the row's true minimum;  if the row is not marked stale, (rmin, rarg) is exact.
// A merge (a, b) changes column a and removes column b.  A row k < a folds the new d(k, a) into its cache when
// that keeps it exact, and is marked stale when its cached column went away; row a's new minimum is a butterfly
// over the values the other rows just computed.  A stale row is rescanned (cooperatively: one LDS read per
// lane + a butterfly) only when the argmin picks it -- its rmin is still a lower bound, so rows whose bound
// never becomes the smallest are never rescanned.  The picked pair is the oracle's: the smallest row index
// among the rows with the smallest bound, exact once clean, and any row before it has a larger bound.

constexpr uint32_t kNoCol = 0xFFFFu;

// Minimum of a 32-bit value over each group of GROUP consecutive lanes, delivered to every lane of the group.
// Inside a 16-lane row the steps are DPP lane permutes fused into v_min_u32 (pairs, quads, halves, row: each step
// combines two sets that are already uniform); across rows the four row minima go through SGPRs.
#define DUET_DPP_MIN(v, ctrl) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xF, 0xF, true))
template <int GROUP>
__device__ __forceinline__ uint32_t group_min_u32(uint32_t v)
{
    if (GROUP >= 2) DUET_DPP_MIN(v, 0xB1);        // quad_perm [1,0,3,2]
    if (GROUP >= 4) DUET_DPP_MIN(v, 0x4E);        // quad_perm [2,3,0,1]
    if (GROUP >= 8) DUET_DPP_MIN(v, 0x141);       // row_half_mirror
    if (GROUP >= 16) DUET_DPP_MIN(v, 0x140);      // row_mirror
    if (GROUP >= 32) {
        const uint32_t r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16);
        const uint32_t r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
        const uint32_t lo = min(r0, r1), hi = min(r2, r3);
        v = GROUP == 64 ? min(lo, hi) : (threadIdx.x < 32 ? lo : hi);
    }
    return v;
}

// minimum of the candidates val[r] (column r * GROUP + sl; non-negative binary64, whose bit patterns order like
// unsigned integers) over the group, and the smallest column attaining it
template <int GROUP, int R>
__device__ __forceinline__ void group_argmin(const double (&val)[R], uint32_t sub, double &gval, uint32_t &gcol)
{
    double m = val[0];
#pragma unroll
    for (int r = 1; r < R; ++r) m = val[r] < m ? val[r] : m;
    const uint32_t hi = (uint32_t)__double2hiint(m), lo = (uint32_t)__double2loint(m);
    const uint32_t mh = group_min_u32<GROUP>(hi);
    uint32_t ml;
    if (GROUP == 64) {
        // one group per wave: when a single lane holds the smallest high word (the usual case: two distances share their top
        // 32 bits only when they agree to 1e-6) its low word is the answer -- one reduction chain instead of two
        const unsigned long long top = __ballot(hi == mh);
        if (__popcll(top) == 1) ml = (uint32_t)__builtin_amdgcn_readlane((int)lo, (int)__ffsll((long long)top) - 1);
        else ml = group_min_u32<GROUP>(hi == mh ? lo : 0xFFFFFFFFu);
    } else {
        ml = group_min_u32<GROUP>(hi == mh ? lo : 0xFFFFFFFFu);
    }
    m = __hiloint2double((int)mh, (int)ml);
    gval = m;
    gcol = kNoCol;
    constexpr unsigned long long gm = GROUP == 64 ? ~0ull : ((1ull << (GROUP & 63)) - 1ull);
#pragma unroll
    for (int r = R - 1; r >= 0; --r) {
        const unsigned long long bal = (__ballot(val[r] == m) >> (sub * GROUP)) & gm;
        if (bal) gcol = (uint32_t)r * GROUP + (uint32_t)__ffsll((long long)bal) - 1u;
    }
}

__device__ __forceinline__ double sp_distance(uint32_t pi, uint32_t spi, uint32_t pj, uint32_t spj, double normalizer)
{
    const uint64_t si = pi, ei = (uint64_t)pi + spi, ci = centre_of(pi, spi);
    const uint64_t sj = pj, ej = (uint64_t)pj + spj, cj = centre_of(pj, spj);
    uint64_t m = si > sj ? si - sj : sj - si;
    const uint64_t m2 = ei > ej ? ei - ej : ej - ei, m3 = ci > cj ? ci - cj : cj - ci;
    m = m2 < m ? m2 : m;
    m = m3 < m ? m3 : m;
    const uint32_t smax = spi > spj ? spi : spj, sdif = spi > spj ? spi - spj : spj - spi;
    const double dp = (double)m / normalizer;
    const double ds = smax ? (double)sdif / (double)smax : 0.0;
    return dp + ds;
}

// index (into the caller's arrays) of the mark at sorted position i
__device__ __forceinline__ uint32_t mark_at(const ClParams &p, uint32_t i)
{
    return p.idx_packed ? (uint32_t)(p.skeys[i] >> p.key_bits) : p.sorted[i];
}

// (pos, span, read index) of mark a (the read index only in the fused pipeline)
__device__ __forceinline__ uint3 load_rec(const ClParams &p, uint32_t a)
{
    if (p.rec4) {
        const uint4 q = p.rec4[a];
        return make_uint3(q.x, q.y, q.z);
    }
    const uint2 q = p.ps[a];
    return make_uint3(q.x, q.y, 0u);
}

// Work item: one connected component of a partition's threshold graph (list entry = partition, then
// root row | rows << 8), gathered in row order from the partition through comp8; writes label8 for its rows.
// NCAP = the most rows a unit can have (<= GROUP * R): with the default part_max of 100 the triangle of a >64-row unit
// takes 39.6 KB instead of 65 KB, i.e. four wavefronts per CU -- one per SIMD -- instead of two.
template <int GROUP, int R, int NCAP = GROUP * R>
struct ExactSmem {
    static constexpr int SUBS = 64 / GROUP, NMAX = NCAP;
    double d[SUBS][NMAX * (NMAX - 1) / 2];                   // upper triangle, row by row
    uint32_t pos[SUBS][NMAX], span[SUBS][NMAX];
    uint8_t row[SUBS][NMAX];
};

template <int GROUP, int R, bool WHOLE = false, int NCAP = GROUP * R>
__device__ __forceinline__ void exact_unit(const ClParams &p, const uint32_t *list, uint32_t L, uint32_t base, unsigned char *smem)
{
    static_assert(NCAP <= GROUP * R && NCAP <= 128, "");
    constexpr int NMAX = NCAP;
    ExactSmem<GROUP, R, NCAP> &X = *reinterpret_cast<ExactSmem<GROUP, R, NCAP> *>(smem);
    const uint32_t lane = threadIdx.x, sub = lane / GROUP, sl = lane % GROUP;
    constexpr unsigned long long gm = GROUP == 64 ? ~0ull : ((1ull << (GROUP & 63)) - 1ull);
    double *D = X.d[sub];
    const double inf = __builtin_inf();
    // d(i, j), i < j, in the upper triangle: half the LDS of a square matrix, so twice the waves per CU
    auto tri = [](uint32_t i, uint32_t j) -> uint32_t { return i * (2u * NMAX - i - 1u) / 2u + (j - i - 1u); };
    {
        const uint32_t li = base + sub;
        const bool has = li < L;
        // WHOLE: the list holds partitions (one word each) and the item is the whole partition
        const uint32_t part = has ? (WHOLE ? list[li] : list[2 * (size_t)li]) : 0u;
        const uint32_t item = has && !WHOLE ? list[2 * (size_t)li + 1] : 0u;
        const uint32_t s = has ? p.part_start[part] : 0u;
        const uint32_t np = has ? p.part_start[part + 1] - s : 0u;       // rows of the partition
        const uint32_t root = item & 0xFFu, n = WHOLE ? np : (has ? item >> 8 : 0u);     // n = rows of the component
        __syncthreads();
        // gather the component's rows, in row order
        uint32_t filled = 0;
        uint32_t np_max = np;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) np_max = max(np_max, (uint32_t)__shfl_xor((int)np_max, d, 64));
        for (uint32_t k0 = 0; k0 < np_max; k0 += GROUP) {
            const uint32_t row = k0 + sl;
            const bool mem = row < np && (WHOLE || p.comp8[s + row] == root);
            const unsigned long long bal = (__ballot(mem) >> (sub * GROUP)) & gm;
            if (mem) {
                const uint32_t ci = filled + (uint32_t)__popcll(bal & ((1ull << sl) - 1ull));
                const uint32_t a = mark_at(p, s + row);
                const uint3 q = load_rec(p, a);
                X.pos[sub][ci] = q.x;
                X.span[sub][ci] = q.y;
                X.row[sub][ci] = (uint8_t)row;
            }
            filled += (uint32_t)__popcll(bal);
        }
        __syncthreads();
        // every unordered pair once: row k takes the columns k+1 .. k+n/2 (mod n); for even n the distance-n/2
        // pairs only from the lower half of the rows
        const uint32_t half = n >> 1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t k = sl + r * GROUP;
            if (k < n) {
                const uint32_t pk = X.pos[sub][k], spk = X.span[sub][k];
                const uint32_t tmax = (!(n & 1u) && k >= half) ? half - 1 : half;
                for (uint32_t t = 1; t <= tmax; ++t) {
                    uint32_t j = k + t;
                    j = j >= n ? j - n : j;
                    const double v = sp_distance(pk, spk, X.pos[sub][j], X.span[sub][j], p.normalizer);
                    D[k < j ? tri(k, j) : tri(j, k)] = v;
                }
            }
        }
        __syncthreads();
        double rmin[R];
        uint32_t rarg[R], lab[R], csize[R];
        bool alive[R], stale[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t k = sl + r * GROUP;
            csize[r] = 1;
            rmin[r] = inf;
            rarg[r] = kNoCol;
            lab[r] = k;
            alive[r] = k < n;
            stale[r] = false;
            for (uint32_t j = k + 1; j < n; ++j) {
                const double v = D[tri(k, j)];
                if (v < rmin[r]) { rmin[r] = v; rarg[r] = j; }
            }
        }
        bool done = n < 2;
        // every iteration merges or cleans a stale row, so the loop ends by itself; the bound is a safety net
        for (uint32_t iter = 0; iter < (uint32_t)NMAX * NMAX; ++iter) {
            double cand[R];
#pragma unroll
            for (int r = 0; r < R; ++r) cand[r] = alive[r] ? rmin[r] : inf;
            double gval;
            uint32_t g;
            group_argmin<GROUP, R>(cand, sub, gval, g);
            const bool act = !done && gval <= p.max_dist && g != kNoCol;      // distances are finite: inf = no pair left
            done = done || !act;
            if (!__ballot(act)) break;
            // the picked row's cache, from its owner lane
            uint32_t mine = 0;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if ((uint32_t)r == g / GROUP) mine = rarg[r] | (stale[r] ? 0x80000000u : 0u);
            // (one group per wave: the owner lane is uniform -- a lane read instead of a trip through the LDS crossbar)
            const uint32_t info = GROUP == 64 ? (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)(g == kNoCol ? 0u : g % GROUP))
                                              : (uint32_t)__shfl(mine, act ? (int)(sub * GROUP + g % GROUP) : (int)lane, 64);
            const bool do_rescan = act && (info >> 31);
            const bool do_merge = act && !(info >> 31);
            if (__ballot(do_rescan)) {
                double cv[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t k = sl + r * GROUP;
                    cv[r] = (do_rescan && alive[r] && k > g) ? D[tri(g, k)] : inf;
                }
                double nv;
                uint32_t nc;
                group_argmin<GROUP, R>(cv, sub, nv, nc);
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (do_rescan && sl + r * GROUP == g) { rmin[r] = nv; rarg[r] = nv < inf ? nc : kNoCol; stale[r] = false; }
            }
            if (__ballot(do_merge)) {
                const uint32_t a = g, b = info & 0xFFFFu;
                // cluster sizes live with their rows' lanes (csize[r] of the lane that owns the row): a lane read per merge
                // instead of two LDS round trips at the head of the merge's dependent chain
                uint32_t za = 0, zb = 0;
                if (GROUP == 64) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t va = (uint32_t)__builtin_amdgcn_readlane((int)csize[r], (int)(a % GROUP));
                        const uint32_t vb = (uint32_t)__builtin_amdgcn_readlane((int)csize[r], (int)(b % GROUP));
                        za = (uint32_t)r == a / GROUP ? va : za;
                        zb = (uint32_t)r == b / GROUP ? vb : zb;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t va = (uint32_t)__shfl((int)csize[r], do_merge ? (int)(sub * GROUP + a % GROUP) : (int)lane, 64);
                        const uint32_t vb = (uint32_t)__shfl((int)csize[r], do_merge ? (int)(sub * GROUP + b % GROUP) : (int)lane, 64);
                        za = (uint32_t)r == a / GROUP ? va : za;
                        zb = (uint32_t)r == b / GROUP ? vb : zb;
                    }
                }
                const double na = (double)za, nb = (double)zb;
                double cv[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t k = sl + r * GROUP;
                    const bool act = do_merge && alive[r] && k != a && k != b;
                    const bool ka = k < a;
                    double v = inf;
                    if (act) {
                        const uint32_t ia = ka ? tri(k, a) : tri(a, k);
                        const double da = D[ia], db = D[k < b ? tri(k, b) : tri(b, k)];
                        v = (na * da + nb * db) / (na + nb);
                        D[ia] = v;
                    }
                    // the row's cached minimum, as selects (no divergent control flow behind the division)
                    const bool upd = act && ka, hit = rarg[r] == a || rarg[r] == b;
                    const bool lt = upd && v < rmin[r], eq = upd && v == rmin[r];
                    const bool take = lt || (eq && !stale[r] && (hit || a < rarg[r]));
                    const bool spoil = (upd && !lt && !eq && hit) || (act && !ka && k < b && rarg[r] == b);
                    rmin[r] = lt ? v : rmin[r];
                    rarg[r] = take ? a : rarg[r];
                    stale[r] = lt ? false : (spoil ? true : stale[r]);
                    cv[r] = (act && !ka) ? v : inf;
                    if (do_merge && lab[r] == b) lab[r] = a;
                }
                double nv;
                uint32_t nc;
                group_argmin<GROUP, R>(cv, sub, nv, nc);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t k = sl + r * GROUP;
                    if (do_merge && k == a) { rmin[r] = nv; rarg[r] = nv < inf ? nc : kNoCol; stale[r] = false; csize[r] = za + zb; }
                    if (do_merge && k == b) alive[r] = false;
                }
            }
            __syncthreads();
        }
        // the component's clusters, named after their smallest row of the partition
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t k = sl + r * GROUP;
            if (k < n) p.label8[s + X.row[sub][k]] = X.row[sub][lab[r]];
        }
    }
}

// components of up to 64 rows, every size class in one launch (largest first).  For small inputs the partitions of more
// than 64 marks do not go through the fast pass at all: one wavefront would spend ~100 us on one of them there, as long
// as the exact agglomeration takes, so cl_exact_big<true> takes them whole, on a side stream, while the fast pass handles
// the rest.  For large inputs there can be very many of them and throughput counts: the fast pass (20 waves per CU
// instead of 2) settles what it can first, cl_exact_big<false> gets the components it leaves.
__global__ __launch_bounds__(64) void cl_exact_small(const ClParams p, const uint32_t *lists, const uint32_t *cnts)
{
    __shared__ __align__(16) unsigned char smem[sizeof(ExactSmem<64, 1>)];
    static_assert(sizeof(ExactSmem<64, 1>) >= sizeof(ExactSmem<32, 1>) && sizeof(ExactSmem<64, 1>) >= sizeof(ExactSmem<16, 1>), "");
    const uint32_t c0 = cnts[0], c1 = cnts[1], c2 = cnts[2];
    const uint32_t b2 = c2, b1 = b2 + (c1 + 1) / 2, b0 = b1 + (c0 + 3) / 4;
    const size_t M = p.M;
    for (uint32_t vb = blockIdx.x; vb < b0; vb += gridDim.x) {
        if (vb < b2) exact_unit<64, 1>(p, lists + 2 * M, c2, vb, smem);
        else if (vb < b1) exact_unit<32, 1>(p, lists + 1 * M, c1, (vb - b2) * 2, smem);
        else exact_unit<16, 1>(p, lists, c0, (vb - b1) * 4, smem);
    }
}

template <bool WHOLE, int NCAP>
__global__ __launch_bounds__(64) void cl_exact_big(const ClParams p, const uint32_t *list, const uint32_t *count)
{
    __shared__ __align__(16) unsigned char smem[sizeof(ExactSmem<64, 2, NCAP>)];
    static_assert(sizeof(ExactSmem<64, 2, 100>) <= 40960, "four units per CU");
    const uint32_t L = *count;
    for (uint32_t vb = blockIdx.x; vb < L; vb += gridDim.x) exact_unit<64, 2, WHOLE, NCAP>(p, list, L, vb, smem);
}

// ---------------------------------------------------------------------------------------------
// agglomeration, fast path: partitions whose clusters can be read off the threshold graph
// ---------------------------------------------------------------------------------------------
//
// What stage A0 emits depends only on the FINAL clusters of a partition (clusters by smallest member, members
// in sorted order, floor means), not on the order of the merges.  Two facts about average linkage pin them down
// without running it (u = 2^-53; a Lance-Williams step rounds 3 times, a merge tree is at most 127 deep, so a
// computed cluster distance lies within (1 +/- 4.3e-14) of the range of the pair distances it averages):
//   * marks in different connected components of the graph {d(i,j) <= max_dist * (1 + 9e-6)} are never merged:
//     every cross distance the oracle ever computes stays above max_dist;
//   * a component in which EVERY pair has d <= max_dist * (1 - 9e-6) ends as exactly one cluster: while two of
//     its clusters remain, their computed distance is below max_dist, so the oracle keeps merging.
// The fast pass evaluates d in binary32 (relative error < 4e-7, hence the 1e-5 guard band around max_dist), builds each
// mark's closed neighbourhood as a bit mask, and accepts the partition when no pair falls inside the guard
// band and every neighbourhood equals the neighbourhood of its smallest member (<=> every component is a
// clique).  Everything else -- about one partition in a few hundred on SV-like data, nearly all on random data
// -- gets the exact binary64 agglomeration (exact_unit), component by component.  Both paths produce
// the oracle's clusters; tests/test_gpu_cluster.py and tools/stress.py cover both.

template <int NW>
struct BitSet {
    uint64_t w[NW];
    __device__ __forceinline__ void clear() { for (int i = 0; i < NW; ++i) w[i] = 0; }
    __device__ __forceinline__ uint32_t count() const { uint32_t c = 0; for (int i = 0; i < NW; ++i) c += __popcll(w[i]); return c; }
    __device__ __forceinline__ uint32_t count_below(uint32_t k) const      // set bits at positions < k
    {
        uint32_t c = 0;
        for (int i = 0; i < NW; ++i) {
            const uint32_t lo = 64u * i;
            const uint64_t m = k >= lo + 64u ? ~0ull : (k > lo ? (1ull << (k - lo)) - 1ull : 0ull);
            c += __popcll(w[i] & m);
        }
        return c;
    }
    __device__ __forceinline__ uint32_t first() const                      // lowest set bit (undefined if empty)
    {
        for (int i = 0; i < NW - 1; ++i)
            if (w[i]) return 64u * i + (uint32_t)__ffsll((long long)w[i]) - 1u;
        return 64u * (NW - 1) + (uint32_t)__ffsll((long long)w[NW - 1]) - 1u;
    }
    // (no dynamically indexed w[]: that would put the set in scratch memory)
    __device__ __forceinline__ bool test(uint32_t j) const
    {
        uint64_t x = w[0];
        for (int i = 1; i < NW; ++i) x = (j >> 6) == (uint32_t)i ? w[i] : x;
        return (x >> (j & 63u)) & 1ull;
    }
    __device__ __forceinline__ void set_if(bool c, uint32_t j)
    {
        for (int i = 0; i < NW; ++i) w[i] |= (c && (NW == 1 || (j >> 6) == (uint32_t)i)) ? 1ull << (j & 63u) : 0ull;
    }
    __device__ __forceinline__ bool equals(const BitSet &o) const { bool e = true; for (int i = 0; i < NW; ++i) e = e && w[i] == o.w[i]; return e; }
};

// |a - b| in ONE instruction (__usad compiles to min, max, sub)
__device__ __forceinline__ uint32_t absdiff_u32(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// What cl_emit needs, per mark: its place in the partition's output, and at each cluster's smallest member the
// cluster's rank, end and floor means.  F[r] = the cluster (bit set over the partition's rows) of this lane's row
// sl + r * GROUP; groups with go == false only keep the collective operations company.  Uses s_mask as scratch.
template <int GROUP, int R, int NW, int NMAX, class PsT>
__device__ __forceinline__ void emit_prep(const ClParams &p, bool go, uint32_t part, uint32_t s, uint32_t n, uint32_t sub,
                                          uint32_t sl, const BitSet<NW> (&F)[R], uint64_t (*s_mask)[NW], const PsT *s_ps,
                                          unsigned long long (*s_sum)[2], const uint32_t (&mk)[R], const uint32_t (&rd)[R])
{
    constexpr unsigned long long gm = GROUP == 64 ? ~0ull : ((1ull << (GROUP & 63)) - 1ull);
    __syncthreads();
    uint32_t rt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t k = sl + r * GROUP;
        rt[r] = k < n ? F[r].first() : k;
        if (go && k < n) {
            for (int i = 0; i < NW; ++i) s_mask[k][i] = F[r].w[i];
            s_sum[k][0] = 0;
            s_sum[k][1] = 0;
        }
    }
    __syncthreads();
    // every mark adds its (pos, span) to its cluster's sums, kept at the cluster head
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t k = sl + r * GROUP;
        if (go && k < n) {
            const PsT q = s_ps[k];
            atomicAdd(&s_sum[rt[r]][0], (unsigned long long)q.x);
            atomicAdd(&s_sum[rt[r]][1], (unsigned long long)q.y);
        }
    }
    __syncthreads();
    BitSet<NW> heads;                                       // cluster heads of the group, held by every lane of the group
    heads.clear();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned long long b = (__ballot(go && sl + r * GROUP < n && rt[r] == sl + r * GROUP) >> (sub * GROUP)) & gm;
        heads.w[(r * GROUP) >> 6] |= b << ((r * GROUP) & 63);
    }
    if (!go) return;
    uint32_t before[R];
#pragma unroll
    for (int r = 0; r < R; ++r) before[r] = 0;
    for (int i = 0; i < NW; ++i) {
        uint64_t hm = heads.w[i];
        while (hm) {
            const uint32_t h = 64u * i + (uint32_t)__ffsll((long long)hm) - 1u;
            hm &= hm - 1ull;
            uint32_t sz = 0;
            for (int c = 0; c < NW; ++c) sz += __popcll(s_mask[h][c]);
#pragma unroll
            for (int r = 0; r < R; ++r) before[r] += h < rt[r] ? sz : 0u;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t k = sl + r * GROUP;
        if (k < n) {
            const bool head = rt[r] == k;
            const uint32_t size = F[r].count(), rank = before[r] + F[r].count_below(k);
            // the mark's place in the output: clusters in order of their smallest member, members in sorted order
            p.order[s + rank] = mk[r];
            if (p.sv_mark_out) p.sv_mark_out[s + rank] = rd[r];
            if (head) {
                // the cluster's record, at the partition's start + the cluster's index: cl_emit walks a partition's clusters
                // from there (one dense record per cluster instead of a word per mark with the heads scattered among them).
                // Floor means: a sum stays below 2^40 and size <= 128, so the correctly rounded binary64 quotient
                // lies strictly between the same two integers as the true one (or is that integer): no 64-bit division
                const uint32_t ci = heads.count_below(rt[r]);
                p.e_rec[s + ci] = make_uint4(rank | ((before[r] + size) << 8), (uint32_t)((double)s_sum[k][0] / (double)size),
                                             (uint32_t)((double)s_sum[k][1] / (double)size), 0u);
            }
        }
    }
    if (sl == 0) p.pc[part] = heads.count();
}

// work lists the fast pass leaves behind: per size class of the components (<= 16 / 32 / 64 / 128 rows) the
// components for cl_exact (two words each), per size class of the partitions the partitions for cl_rank
struct ClWork {
    uint32_t *comp_list;      // [4][M]
    uint32_t *comp_count;     // [4]
    uint32_t *rank_list;      // [kClasses][M]
    uint32_t *rank_count;     // [kClasses]
    uint32_t *why;            // diagnostics (DUET_CL_DEBUG): why the whole-partition test gave up, or null
};

constexpr int kLevels = 3;              // thresholds max_dist, max_dist / 2, max_dist / 4

// Average linkage over m <= KA "atoms" (clusters already known to form first), every lane of the group running
// the same steps on the group's LDS scratch: D[a * KA + b] (a < b) the atoms' average distances, sz their sizes,
// lab[a] the atom that a's cluster is named after (its smallest atom).  Returns false when a decision -- which
// pair is closest, whether it is within max_dist -- is not safe against a 1e-4 relative error of D.
template <int KA>
__device__ __forceinline__ bool atoms_linkage(uint32_t m, double *D, double *sz, uint32_t *lab, double max_dist)
{
    uint32_t alive = m >= 32 ? ~0u : (1u << m) - 1u;
    for (uint32_t step = 0; step + 1 < m; ++step) {
        double d1 = __builtin_inf(), d2 = __builtin_inf();
        uint32_t a1 = 0, b1 = 1;
        for (uint32_t a = 0; a + 1 < m; ++a) {
            if (!((alive >> a) & 1u)) continue;
            for (uint32_t b = a + 1; b < m; ++b) {
                if (!((alive >> b) & 1u)) continue;
                const double v = D[a * KA + b];
                if (v < d1) { d2 = d1; d1 = v; a1 = a; b1 = b; }
                else if (v < d2) d2 = v;
            }
        }
        if (!(d2 > d1 * (1.0 + 1e-4))) return false;                       // a near-tie for the closest pair
        if (!(fabs(d1 - max_dist) > 1e-4 * max_dist)) return false;         // too close to the threshold to call
        if (d1 > max_dist) break;
        const double na = sz[a1], nb = sz[b1];
        for (uint32_t c = 0; c < m; ++c) {
            if (!((alive >> c) & 1u) || c == a1 || c == b1) continue;
            const uint32_t ia = c < a1 ? c * KA + a1 : a1 * KA + c, ib = c < b1 ? c * KA + b1 : b1 * KA + c;
            D[ia] = (na * D[ia] + nb * D[ib]) / (na + nb);
        }
        sz[a1] = na + nb;
        alive &= ~(1u << b1);
        for (uint32_t c = 0; c < m; ++c)
            if (lab[c] == b1) lab[c] = a1;
    }
    return true;
}

template <int GROUP, int R, int KA>
struct FastSmem {
    static constexpr int SUBS = 64 / GROUP, NMAX = GROUP * R, NW = NMAX > 64 ? 2 : 1;
    uint4 ps[SUBS][NMAX];                                    // (pos, span, end, centre): one 16-byte broadcast read per pair
    uint64_t mask[SUBS][NMAX][NW];
    double D[SUBS][KA][KA], sz[SUBS][KA];
    uint32_t csz[SUBS][NMAX];                                // rows per component, at the component's smallest row
    unsigned long long sum[SUBS][NMAX][2];
    uint32_t lab[SUBS][KA], aroot[SUBS][KA];
    uint8_t atom[SUBS][NMAX];
};

// one wave's worth of partitions (64 / GROUP of them, list[base ...]) of one size class
template <int GROUP, int R, int KA>
__device__ __forceinline__ void fast_unit(const ClParams &p, const uint32_t *list, uint32_t L, uint32_t base, const ClWork &work,
                                          unsigned char *smem)
{
    constexpr int NMAX = GROUP * R, NW = NMAX > 64 ? 2 : 1;
    static_assert(NMAX <= 128 && (R == 1 || (GROUP * R) % 64 == 0 || GROUP * R <= 64), "unsupported shape");
    FastSmem<GROUP, R, KA> &S = *reinterpret_cast<FastSmem<GROUP, R, KA> *>(smem);
    const uint32_t lane = threadIdx.x, sub = lane / GROUP, sl = lane % GROUP;
    constexpr unsigned long long gm = GROUP == 64 ? ~0ull : ((1ull << (GROUP & 63)) - 1ull);
    auto group_any = [&](bool x) -> bool { return ((__ballot(x) >> (sub * GROUP)) & gm) != 0ull; };
    {
        const uint32_t li = base + sub;
        const bool has = li < L;
        const uint32_t part = has ? list[li] : 0u;
        const uint32_t s = has ? p.part_start[part] : 0u;
        const uint32_t n = has ? p.part_start[part + 1] - s : 0u;
        __syncthreads();
        uint32_t pk[R], spk[R], ek[R], ck[R], mk[R], rd[R];
        bool bad = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t k = sl + r * GROUP;
            pk[r] = spk[r] = mk[r] = rd[r] = 0;
            if (k < n) {
                mk[r] = mark_at(p, s + k);
                const uint3 q = load_rec(p, mk[r]);
                pk[r] = q.x;
                spk[r] = q.y;
                rd[r] = q.z;
            }
            ek[r] = pk[r] + spk[r];
            ck[r] = pk[r] + (spk[r] >> 1);
            if (k < n) S.ps[sub][k] = make_uint4(pk[r], spk[r], ek[r], ck[r]);
            bad = bad || ek[r] < pk[r];                      // end does not fit 32 bits: leave it to the exact path
        }
        __syncthreads();
        // closed neighbourhoods at the thresholds; amb: some pair sits inside a threshold's guard band.  Level 0
        // (max_dist itself) for everybody; the finer levels only where level 0 does not settle the partition.
        BitSet<NW> N[kLevels][R];
        bool amb[kLevels], amb0r[R];
#pragma unroll
        for (int r = 0; r < R; ++r) amb0r[r] = false;
#pragma unroll
        for (int l = 0; l < kLevels; ++l) {
            amb[l] = false;
#pragma unroll
            for (int r = 0; r < R; ++r) N[l][r].clear();
        }
        // (32 columns at a time: the bits of one mask word are collected in one register, two instructions per pair)
        auto level0 = [&](auto wc) {
            constexpr uint32_t W = decltype(wc)::value;
            const uint32_t j1 = min(n, 32u * (W + 1u));
            uint32_t acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = 0;
            for (uint32_t j = 32u * W; j < j1; ++j) {
                const uint4 q = S.ps[sub][j];
                const uint32_t ej = q.z, cj = q.w;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t m = min(min(absdiff_u32(pk[r], q.x), absdiff_u32(ek[r], ej)), absdiff_u32(ck[r], cj));
                    const float fm = (float)max(max(spk[r], q.y), 1u), fs = (float)absdiff_u32(spk[r], q.y);
                    const float dp = (float)m * p.inv_norm;
                    const bool e_hi = fs <= (p.t_hi[0] - dp) * fm, e_lo = fs <= (p.t_lo[0] - dp) * fm;
                    amb0r[r] = amb0r[r] || e_hi != e_lo;
                    acc[r] |= (e_hi ? 1u : 0u) << (j - 32u * W);
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) N[0][r].w[W >> 1] |= (uint64_t)acc[r] << (32u * (W & 1u));
        };
        level0(std::integral_constant<uint32_t, 0>{});
        if constexpr (NMAX > 32) level0(std::integral_constant<uint32_t, 1>{});
        if constexpr (NMAX > 64) {
            level0(std::integral_constant<uint32_t, 2>{});
            level0(std::integral_constant<uint32_t, 3>{});
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            N[0][r].set_if(sl + r * GROUP < n, sl + r * GROUP);
            amb[0] = amb[0] || amb0r[r];
        }
        // every component of a level's graph is a clique <=> each mark's neighbourhood equals that of its
        // smallest member; leaves the level's masks in s_mask
        auto cliques = [&](const BitSet<NW> (&Nl)[R]) -> bool {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t k = sl + r * GROUP;
                if (k < n)
                    for (int i = 0; i < NW; ++i) S.mask[sub][k][i] = Nl[r].w[i];
            }
            __syncthreads();
            bool differs = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t k = sl + r * GROUP;
                if (k < n) {
                    const uint32_t f = Nl[r].first();
                    BitSet<NW> o;
                    for (int i = 0; i < NW; ++i) o.w[i] = S.mask[sub][f][i];
                    differs = differs || !o.equals(Nl[r]);
                }
            }
            return !group_any(differs);
        };
        const bool unfit = group_any(bad) || !p.fast;
        BitSet<NW> F[R];                                     // the final cluster of each of this lane's marks
#pragma unroll
        for (int r = 0; r < R; ++r) F[r] = N[0][r];
        const bool amb0 = group_any(amb[0]), cl0 = cliques(N[0]);       // collective: every lane takes part
        bool solved = n < 2 || (!unfit && !amb0 && cl0);
        const bool want2 = has && !solved && !unfit;
        if (work.why && want2 && sl == 0) atomicAdd(&work.why[(GROUP == 64 ? (R == 2 ? 0 : 4) : 8) + 3], 1u);     // level 0 did not settle it
        if (__ballot(want2)) {
            // atoms: the cliques of the finest usable level -- they are complete clusters before anything else
            // happens (every pair inside is closer than every pair across), so what remains is average linkage over
            // the atoms, decided from binary32 estimates of their average distances when that is safe
            const uint32_t nw = want2 ? n : 0u;
            auto finer = [&](auto wc) {
                constexpr uint32_t W = decltype(wc)::value;
                const uint32_t j1 = min(nw, 32u * (W + 1u));
                uint32_t acc[kLevels][R];
#pragma unroll
                for (int l = 1; l < kLevels; ++l)
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[l][r] = 0;
                for (uint32_t j = 32u * W; j < j1; ++j) {
                    const uint4 q = S.ps[sub][j];
                    const uint32_t ej = q.z, cj = q.w;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t m = min(min(absdiff_u32(pk[r], q.x), absdiff_u32(ek[r], ej)), absdiff_u32(ck[r], cj));
                        const float fm = (float)max(max(spk[r], q.y), 1u), fs = (float)absdiff_u32(spk[r], q.y);
                        const float dp = (float)m * p.inv_norm;
#pragma unroll
                        for (int l = 1; l < kLevels; ++l) {
                            const bool e_hi = fs <= (p.t_hi[l] - dp) * fm, e_lo = fs <= (p.t_lo[l] - dp) * fm;
                            amb[l] = amb[l] || e_hi != e_lo;
                            acc[l][r] |= (e_hi ? 1u : 0u) << (j - 32u * W);
                        }
                    }
                }
#pragma unroll
                for (int l = 1; l < kLevels; ++l)
#pragma unroll
                    for (int r = 0; r < R; ++r) N[l][r].w[W >> 1] |= (uint64_t)acc[l][r] << (32u * (W & 1u));
            };
            finer(std::integral_constant<uint32_t, 0>{});
            if constexpr (NMAX > 32) finer(std::integral_constant<uint32_t, 1>{});
            if constexpr (NMAX > 64) {
                finer(std::integral_constant<uint32_t, 2>{});
                finer(std::integral_constant<uint32_t, 3>{});
            }
#pragma unroll
            for (int l = 1; l < kLevels; ++l)
#pragma unroll
                for (int r = 0; r < R; ++r) N[l][r].set_if(want2 && sl + r * GROUP < n, sl + r * GROUP);
            const bool amb2 = group_any(amb[2]), cl2 = cliques(N[2]);
            const bool amb1 = group_any(amb[1]), cl1 = cliques(N[1]);
            const bool ok2 = !amb2 && cl2, ok1 = !amb1 && cl1;
            bool two = want2 && (ok1 || ok2);
            if (work.why && want2 && !two && sl == 0) atomicAdd(&work.why[GROUP == 64 ? (R == 2 ? 0 : 4) : 8], 1u);       // no clean level
            BitSet<NW> A[R];
            uint32_t ra[R], ai[R];
            BitSet<NW> heads;
            heads.clear();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                A[r] = ok1 ? N[1][r] : N[2][r];
                ra[r] = A[r].first();
                const unsigned long long b = (__ballot(two && sl + r * GROUP < n && ra[r] == sl + r * GROUP) >> (sub * GROUP)) & gm;
                heads.w[(r * GROUP) >> 6] |= b << ((r * GROUP) & 63);
            }
            const uint32_t m = heads.count();
            if (work.why && two && m > (uint32_t)KA && sl == 0) atomicAdd(&work.why[(GROUP == 64 ? (R == 2 ? 0 : 4) : 8) + 1], 1u);   // too many atoms
            two = two && m <= (uint32_t)KA;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t k = sl + r * GROUP;
                ai[r] = two ? heads.count_below(ra[r]) : 0u;
                if (two && k < n) {
                    for (int i = 0; i < NW; ++i) S.mask[sub][k][i] = A[r].w[i];      // the chosen level's atoms
                    S.atom[sub][k] = (uint8_t)ai[r];
                    if (ra[r] == k) {
                        S.aroot[sub][ai[r]] = k;
                        S.sz[sub][ai[r]] = (double)A[r].count();
                        S.lab[sub][ai[r]] = ai[r];
                    }
                }
            }
            if (two)
                for (uint32_t e = sl; e < (uint32_t)(KA * KA); e += GROUP) S.D[sub][e / KA][e % KA] = 0.0;
            __syncthreads();
            if (__ballot(two)) {
                // Row k's distances to the columns of one atom, summed while consecutive columns stay in that atom and
                // then added to the atom pair's total in LDS (rows are in centre order, atoms mostly contiguous runs:
                // about one flush per atom instead of a select per atom for every column; any order is correct)
                float cur[R];
#pragma unroll
                for (int r = 0; r < R; ++r) cur[r] = 0.f;
                const uint32_t n2 = two ? n : 0u;
                uint32_t prev = n2 ? (uint32_t)S.atom[sub][0] : 0u;
                auto flush = [&](uint32_t b) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if (sl + r * GROUP < n && b != ai[r]) atomicAdd(&S.D[sub][ai[r]][b], (double)cur[r]);
                        cur[r] = 0.f;
                    }
                };
                for (uint32_t j = 0; j < n2; ++j) {
                    const uint4 q = S.ps[sub][j];
                    const uint32_t aj = S.atom[sub][j];
                    const uint32_t ej = q.z, cj = q.w;
                    if (aj != prev) {
                        flush(prev);
                        prev = aj;
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t mm = min(min(absdiff_u32(pk[r], q.x), absdiff_u32(ek[r], ej)), absdiff_u32(ck[r], cj));
                        const float fm = (float)max(max(spk[r], q.y), 1u), fs = (float)absdiff_u32(spk[r], q.y);
                        cur[r] += (float)mm * p.inv_norm + fs * __builtin_amdgcn_rcpf(fm);
                    }
                }
                if (n2) flush(prev);
                __syncthreads();
                if (two) {
                    // averages into the upper triangle (lanes share the pairs), then every lane of the group
                    // runs the linkage on them (all lanes write the same values)
                    for (uint32_t e = sl; e < m * m; e += GROUP) {
                        const uint32_t a = e / m, b = e % m;
                        if (a < b) S.D[sub][a][b] = (S.D[sub][a][b] + S.D[sub][b][a]) / (2.0 * S.sz[sub][a] * S.sz[sub][b]);
                    }
                }
                __syncthreads();
                bool okl = false;
                if (two) okl = atoms_linkage<KA>(m, &S.D[sub][0][0], S.sz[sub], S.lab[sub], p.max_dist);
                __syncthreads();
                if (work.why && two && !okl && sl == 0) atomicAdd(&work.why[(GROUP == 64 ? (R == 2 ? 0 : 4) : 8) + 2], 1u);       // a decision too close
                if (two && okl) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t mine = S.lab[sub][ai[r]];
                        F[r].clear();
                        for (uint32_t b = 0; b < m; ++b)
                            if (S.lab[sub][b] == mine)
                                for (int i = 0; i < NW; ++i) F[r].w[i] |= S.mask[sub][S.aroot[sub][b]][i];
                    }
                    solved = true;
                }
            }
        }
        if (__ballot(has && !solved)) {
            // Not settled as a whole: settle it component by component.  A row is clean when it and all its
            // neighbours (level 0) have guard-band-free neighbourhoods equal to that of their smallest member: its
            // component is then a clique and one cluster.  The other rows are labelled with the smallest row of their
            // component (min-label propagation over the neighbourhoods) and handed to cl_exact component by
            // component; cl_rank finishes the partition once every row has its label.
            const bool todo = has && !solved;
            __syncthreads();
            BitSet<NW> pass;
            pass.clear();
            bool clean[R];
            // s_mask may hold another level by now: put level 0 back, then test the rows
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t k = sl + r * GROUP;
                if (todo && k < n)
                    for (int i = 0; i < NW; ++i) S.mask[sub][k][i] = N[0][r].w[i];
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t k = sl + r * GROUP;
                bool ok = todo && !unfit && k < n && !amb0r[r];
                if (ok) {
                    const uint32_t f = N[0][r].first();
                    BitSet<NW> o;
                    for (int i = 0; i < NW; ++i) o.w[i] = S.mask[sub][f][i];
                    ok = o.equals(N[0][r]);
                }
                const unsigned long long b = (__ballot(ok) >> (sub * GROUP)) & gm;
                pass.w[(r * GROUP) >> 6] |= b << ((r * GROUP) & 63);
            }
            uint32_t lab[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t k = sl + r * GROUP;
                bool c = todo && !unfit && k < n;
                for (int i = 0; i < NW; ++i) c = c && (N[0][r].w[i] & ~pass.w[i]) == 0ull;
                clean[r] = c;
                lab[r] = unfit ? 0u : k;                     // unfit: the masks mean nothing, the partition is one component
                if (todo && k < n) { S.atom[sub][k] = (uint8_t)lab[r]; S.csz[sub][k] = 0; }
            }
            __syncthreads();
            for (;;) {
                bool changed = false;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t k = sl + r * GROUP;
                    if (todo && !unfit && k < n && !clean[r]) {
                        uint32_t m = lab[r];
                        for (int i = 0; i < NW; ++i) {
                            uint64_t w = N[0][r].w[i];
                            while (w) {
                                const uint32_t j = 64u * i + (uint32_t)__ffsll((long long)w) - 1u;
                                w &= w - 1ull;
                                m = min(m, (uint32_t)S.atom[sub][j]);
                            }
                        }
                        changed = changed || m != lab[r];
                        lab[r] = m;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (todo && sl + r * GROUP < n && !clean[r]) S.atom[sub][sl + r * GROUP] = (uint8_t)lab[r];
                __syncthreads();
                if (!__ballot(changed)) break;
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (todo && sl + r * GROUP < n && !clean[r]) atomicAdd(&S.csz[sub][lab[r]], 1u);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t k = sl + r * GROUP;
                if (todo && k < n) {
                    p.label8[s + k] = clean[r] ? (uint8_t)N[0][r].first() : (uint8_t)0xFF;
                    p.comp8[s + k] = clean[r] ? (uint8_t)0xFF : (uint8_t)lab[r];
                    if (!clean[r] && lab[r] == k) {
                        const uint32_t m = S.csz[sub][k];
                        const uint32_t hc = m <= 16 ? 0u : (m <= 32 ? 1u : (m <= 64 ? 2u : 3u));
                        const uint32_t at = atomicAdd(&work.comp_count[hc], 1u);
                        work.comp_list[(size_t)hc * p.M + 2 * (size_t)at] = part;
                        work.comp_list[(size_t)hc * p.M + 2 * (size_t)at + 1] = k | (m << 8);
                    }
                }
            }
            if (todo && sl == 0) {
                const int rc = size_class(n);
                work.rank_list[(size_t)rc * p.M + atomicAdd(&work.rank_count[rc], 1u)] = part;
            }
        }
        emit_prep<GROUP, R, NW, NMAX>(p, has && solved, part, s, n, sub, sl, F, S.mask[sub], S.ps[sub], S.sum[sub], mk, rd);
    }
}

// every size class in one launch, largest partitions first (they are the longest chains): a virtual block is one
// wave's worth of partitions of one class
constexpr size_t kFastSmemBytes = sizeof(FastSmem<64, 1, 8>) > sizeof(FastSmem<8, 1, 4>) ? sizeof(FastSmem<64, 1, 8>) : sizeof(FastSmem<8, 1, 4>);
static_assert(kFastSmemBytes >= sizeof(FastSmem<32, 1, 4>) && kFastSmemBytes >= sizeof(FastSmem<16, 1, 4>), "shared scratch too small");

// one size class per launch: what large inputs use (the fused kernel needs the registers of all five variants at
// once, which halves the occupancy; with millions of partitions per class there is nothing to gain from fusing)
template <int GROUP, int R, int KA>
__global__ __launch_bounds__(64) void cl_fast_one(const ClParams p, const uint32_t *list, const uint32_t *count, const ClWork work)
{
    __shared__ __align__(16) unsigned char smem[sizeof(FastSmem<GROUP, R, KA>)];
    const uint32_t L = *count;
    for (uint32_t base = blockIdx.x * (64 / GROUP); base < L; base += gridDim.x * (64 / GROUP))
        fast_unit<GROUP, R, KA>(p, list, L, base, work, smem);
}

__global__ __launch_bounds__(64, 4) void cl_fast_all(const ClParams p, const uint32_t *lists, const uint32_t *cnts, const ClWork work)
{
    __shared__ __align__(16) unsigned char smem[kFastSmemBytes];
    const uint32_t c0 = cnts[0], c1 = cnts[1], c2 = cnts[2], c3 = cnts[3];
    const uint32_t b3 = c3, b2 = b3 + (c2 + 1) / 2, b1 = b2 + (c1 + 3) / 4, b0 = b1 + (c0 + 7) / 8;
    const size_t M = p.M;
    for (uint32_t vb = blockIdx.x; vb < b0; vb += gridDim.x) {
        if (vb < b3) fast_unit<64, 1, 8>(p, lists + 3 * M, c3, vb, work, smem);
        else if (vb < b2) fast_unit<32, 1, 4>(p, lists + 2 * M, c2, (vb - b3) * 2, work, smem);
        else if (vb < b1) fast_unit<16, 1, 4>(p, lists + 1 * M, c1, (vb - b2) * 4, work, smem);
        else fast_unit<8, 1, 4>(p, lists, c0, (vb - b1) * 8, work, smem);
    }
}

// Partitions the fast pass did not settle as a whole, after cl_exact: every row has its label; group the rows by
// label and finish like the fast pass does.
template <int GROUP, int R>
struct RankSmem {
    static constexpr int SUBS = 64 / GROUP, NMAX = GROUP * R, NW = NMAX > 64 ? 2 : 1;
    uint2 ps[SUBS][NMAX];
    uint64_t mask[SUBS][NMAX][NW];
    unsigned long long sum[SUBS][NMAX][2];
};

template <int GROUP, int R>
__device__ __forceinline__ void rank_unit(const ClParams &p, const uint32_t *list, uint32_t L, uint32_t base, unsigned char *smem)
{
    constexpr int NMAX = GROUP * R, NW = NMAX > 64 ? 2 : 1;
    RankSmem<GROUP, R> &S = *reinterpret_cast<RankSmem<GROUP, R> *>(smem);
    const uint32_t lane = threadIdx.x, sub = lane / GROUP, sl = lane % GROUP;
    constexpr unsigned long long gm = GROUP == 64 ? ~0ull : ((1ull << (GROUP & 63)) - 1ull);
    {
        const uint32_t li = base + sub;
        const bool has = li < L;
        const uint32_t part = has ? list[li] : 0u;
        const uint32_t s = has ? p.part_start[part] : 0u;
        const uint32_t n = has ? p.part_start[part + 1] - s : 0u;
        __syncthreads();
        uint32_t lab[R], mk[R], rd[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t k = sl + r * GROUP;
            lab[r] = 0xFFFFu;
            mk[r] = rd[r] = 0;
            if (k < n) {
                mk[r] = mark_at(p, s + k);
                const uint3 q = load_rec(p, mk[r]);
                S.ps[sub][k] = make_uint2(q.x, q.y);
                rd[r] = q.z;
                lab[r] = p.label8[s + k];
            }
        }
        BitSet<NW> heads, F[R];
        heads.clear();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            F[r].clear();
            const unsigned long long b = (__ballot(lab[r] == sl + r * GROUP) >> (sub * GROUP)) & gm;
            heads.w[(r * GROUP) >> 6] |= b << ((r * GROUP) & 63);
        }
        // one round per cluster head of any group of the wave (wave-uniform trip count: ballots inside)
        BitSet<NW> left = heads;
        for (;;) {
            bool any = false;
            for (int i = 0; i < NW; ++i) any = any || left.w[i] != 0ull;
            if (!__ballot(any)) break;
            const uint32_t h = any ? left.first() : 0xFFFFFFu;
            BitSet<NW> mh;
            mh.clear();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const unsigned long long b = (__ballot(any && lab[r] == h) >> (sub * GROUP)) & gm;
                mh.w[(r * GROUP) >> 6] |= b << ((r * GROUP) & 63);
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (any && lab[r] == h) F[r] = mh;
            if (any)
                for (int i = 0; i < NW; ++i) left.w[i] &= (h >> 6) == (uint32_t)i ? ~(1ull << (h & 63u)) : ~0ull;
        }
        emit_prep<GROUP, R, NW, NMAX>(p, has, part, s, n, sub, sl, F, S.mask[sub], S.ps[sub], S.sum[sub], mk, rd);
    }
}

// (the partitions of more than 64 marks -- all of them went to the exact pass -- come straight from their class list)
__global__ __launch_bounds__(64) void cl_rank_all(const ClParams p, const uint32_t *lists, const uint32_t *cnts, const uint32_t *big_list,
                                                  const uint32_t *big_count)
{
    __shared__ __align__(16) unsigned char smem[sizeof(RankSmem<64, 2>)];
    const uint32_t c0 = cnts[0], c1 = cnts[1], c2 = cnts[2], c3 = cnts[3], c4 = *big_count;
    const uint32_t b4 = c4, b3 = b4 + c3, b2 = b3 + (c2 + 1) / 2, b1 = b2 + (c1 + 3) / 4, b0 = b1 + (c0 + 7) / 8;
    const size_t M = p.M;
    for (uint32_t vb = blockIdx.x; vb < b0; vb += gridDim.x) {
        if (vb < b4) rank_unit<64, 2>(p, big_list, c4, vb, smem);
        else if (vb < b3) rank_unit<64, 1>(p, lists + 3 * M, c3, vb - b4, smem);
        else if (vb < b2) rank_unit<32, 1>(p, lists + 2 * M, c2, (vb - b3) * 2, smem);
        else if (vb < b1) rank_unit<16, 1>(p, lists + 1 * M, c1, (vb - b2) * 4, smem);
        else rank_unit<8, 1>(p, lists, c0, (vb - b1) * 8, smem);
    }
}

// one thread per partition (their count lives on the device: launched over an upper bound): its clusters' records, dense from the
// partition's start, become the candidates cbase[part] ...; neighbouring partitions write neighbouring candidates
__global__ void cl_emit(const ClParams p)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) p.cand_off[0] = 0;
    const uint32_t n_parts = *p.n_parts;
    // (a grid of a fraction of the bound -- the marks -- strides over the partitions: see cl_classes)
    for (uint32_t part = blockIdx.x * blockDim.x + threadIdx.x; part < n_parts; part += gridDim.x * blockDim.x) {
        const uint32_t s = p.part_start[part], nc = p.pc[part], c0 = p.cbase[part];
        const uint64_t hi = (p.skeys[s] & key_mask(p.key_bits)) >> p.centre_bits;                        // contig | type, straight from the sorted key
        const uint32_t k = (uint32_t)(hi >> p.type_bits), type = (uint32_t)(hi & ((1ull << p.type_bits) - 1ull));
        uint32_t d_lo = 0, nb = 0;
        if (p.sv_svread) {
            d_lo = p.sv_depth_off[k];
            nb = p.sv_depth_off[k + 1] - d_lo;
        }
        // four clusters at a time, their loads side by side: records first, then the depth bins the records' positions select
        // (one cluster per trip made a partition's clusters queue behind each other's two round trips)
        for (uint32_t cb = 0; cb < nc; cb += 4) {
            uint4 rec[4];
            uint32_t d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) rec[j] = p.e_rec[s + min(cb + j, nc - 1u)];
            if (p.sv_svread) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t bin = rec[j].y / p.sv_depth_bin;
                    bin = bin < nb ? bin : nb - 1;
                    d[j] = nb ? p.sv_depth[d_lo + bin] : 0u;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (cb + j >= nc) break;
                const uint32_t info = rec[j].x, cand = c0 + cb + j;
                p.cand_off[cand + 1] = s + (info >> 8);
                p.cand_contig[cand] = (uint16_t)k;
                p.cand_type[cand] = (uint8_t)type;
                p.cand_pos[cand] = rec[j].y;
                p.cand_span[cand] = rec[j].z;
                if (p.sv_svread) {
                    // what a caller VCF would have carried: support = members, reference reads = depth(contig, pos) - support
                    const uint32_t support = (info >> 8) - (info & 0xFFu);                  // a cluster's end - its first member's rank
                    p.sv_svread[cand] = support;
                    p.sv_refread[cand] = d[j] > support ? d[j] - support : 0u;
                    p.sv_gt[cand] = 1;
                }
            }
        }
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


}  // namespace

extern "C" {

}  // extern "C"

namespace {
struct SvExtra {
    const uint32_t *mark_in, *depth, *depth_off;
    uint32_t depth_bin;
    uint32_t *mark_out, *svread, *refread;
    uint8_t *gt;
};
int cluster_run(duet_ctx *ctx, const duet_cluster_problem *pr, const duet_cluster_result *res, void *stream_, const SvExtra *sv);
}  // namespace

extern "C" {

int duet_cluster_run_device(duet_ctx *ctx, const duet_cluster_problem *pr, const duet_cluster_result *res, void *stream_)
{
    return cluster_run(ctx, pr, res, stream_, nullptr);
}

}  // extern "C"

namespace {

int cluster_run(duet_ctx *ctx, const duet_cluster_problem *pr, const duet_cluster_result *res, void *stream_, const SvExtra *sv)
{
    if (!ctx) return duet_fail(nullptr, DUET_ERR_INVALID, "null context");
    if (!pr || !res) return duet_fail(ctx, DUET_ERR_INVALID, "null argument");
    if (pr->part_max < 1 || pr->part_max > 128) return duet_fail(ctx, DUET_ERR_INVALID, "part_max must be in 1..128");
    if (!(pr->normalizer >= 1e-280 && pr->normalizer <= 1e300))
        return duet_fail(ctx, DUET_ERR_INVALID, "normalizer must be in [1e-280, 1e300]");     // keeps every distance finite
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

    const uint32_t nb_sc = (M + kScanTile - 1) / kScanTile;
    const uint32_t nb_rx = (M + kRxTile - 1) / kRxTile;
    const uint32_t nb_hs = (256u * nb_rx + kScanTile - 1) / kScanTile;      // scan tiles of the radix histogram
    const size_t sizes[14] = {(size_t)M * 8, (size_t)M * 8, (size_t)M * 4, (size_t)M * 4, (size_t)256 * nb_rx * 4,
                              ((size_t)M + 1) * 4, ((size_t)M + 1) * 4, ((size_t)(nb_sc > nb_hs ? nb_sc : nb_hs) + 1) * 4,
                              ((size_t)M + 1) * 4, (size_t)M * 16, (size_t)M * 4, 128, (size_t)M * 4 * (2 * kClasses + 4), ((size_t)M + 1) * 4 * 2 + 16 + (size_t)M * (sv ? 16 : 8)};
    int rc;
    for (int i = 0; i < 14; ++i)
        if ((rc = duet_reserve(ctx, ctx->cl_ws[i], sizes[i]))) return rc;
    uint64_t *keysA = (uint64_t *)ctx->cl_ws[0].ptr, *keysB = (uint64_t *)ctx->cl_ws[1].ptr;
    uint32_t *valsA = (uint32_t *)ctx->cl_ws[2].ptr, *valsB = (uint32_t *)ctx->cl_ws[3].ptr;
    uint32_t *hist = (uint32_t *)ctx->cl_ws[4].ptr;
    uint32_t *tmpA = (uint32_t *)ctx->cl_ws[5].ptr;
    uint32_t *spart = (uint32_t *)ctx->cl_ws[7].ptr;
    uint32_t *part_start = (uint32_t *)ctx->cl_ws[8].ptr;
    uint32_t *cbase = (uint32_t *)ctx->cl_ws[13].ptr + (M + 1);      // (the first M + 1 words hold label8 / comp8)
    uint32_t *pc = (uint32_t *)ctx->cl_ws[10].ptr;
    uint32_t *scal = (uint32_t *)ctx->cl_ws[11].ptr;      // [0] = n_parts

    ClParams p;
    memset(&p, 0, sizeof(p));
    p.M = M; p.part_gap = pr->part_gap; p.part_max = pr->part_max;
    p.max_dist = pr->max_dist; p.normalizer = pr->normalizer;
    p.contig = pr->mark_contig; p.type = pr->mark_type; p.pos = pr->mark_pos; p.span = pr->mark_span;
    uint32_t contig_bits;
    if (pr->max_pos_hint && pr->max_span_hint && pr->n_types_hint && pr->n_contigs_hint) {
        p.centre_bits = bits_for((uint64_t)pr->max_pos_hint + pr->max_span_hint / 2);
        p.type_bits = bits_for(pr->n_types_hint - 1);
        contig_bits = bits_for(pr->n_contigs_hint - 1);
    } else {
        // no (or partial) hints: measure -- one small kernel and one host round trip, instead of sorting 58-bit keys
        uint32_t *d_max = scal + 28;
        uint32_t h_max[4] = {0, 0, 0, 0};
        HIP_TRY(ctx, hipMemsetAsync(d_max, 0, 16, st));
        hipLaunchKernelGGL(cl_maxima, dim3((M + 2047) / 2048 < 1024u ? (M + 2047) / 2048 : 1024u), dim3(256), 0, st, p, d_max);
        HIP_TRY(ctx, hipMemcpyAsync(h_max, d_max, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        p.centre_bits = bits_for(((uint64_t)h_max[3] << 32) | h_max[2]);
        p.type_bits = bits_for(h_max[1]);
        contig_bits = bits_for(h_max[0]);
    }
    const uint32_t key_bits = p.centre_bits + p.type_bits + contig_bits;
    if (key_bits > 64) return duet_fail(ctx, DUET_ERR_INVALID, "sort key does not fit 64 bits");
    p.key_bits = key_bits;
    p.idx_packed = key_bits + bits_for(M - 1) <= 64 && !(ctx->dbg & DUET_DBG_CLUSTER_PAIRS);

    // outputs (the agglomeration kernels write the marks' output order themselves)
    if (sv) {
        p.sv_mark_in = sv->mark_in; p.sv_depth = sv->depth; p.sv_depth_off = sv->depth_off; p.sv_depth_bin = sv->depth_bin;
        p.sv_mark_out = sv->mark_out; p.sv_svread = sv->svread; p.sv_refread = sv->refread; p.sv_gt = sv->gt;
    }
    p.order = res->order; p.cand_off = res->cand_off; p.cand_pos = res->cand_pos; p.cand_span = res->cand_span;
    p.cand_contig = res->cand_contig; p.cand_type = res->cand_type;

    const dim3 g256((M + 255) / 256), b256(256);
    unsigned char *recs = (unsigned char *)ctx->cl_ws[13].ptr + ((8 * ((size_t)M + 1) + 15) & ~(size_t)15);
    if (sv) p.rec4 = (const uint4 *)recs;
    else p.ps = (const uint2 *)recs;
    const bool big_sort = (ctx->dbg & DUET_DBG_CLUSTER_LARGE) != 0;     // (tests: the tile-offset path of > 4 M keys)
    const bool rx_totals = nb_rx <= kRxTotalsTiles && !big_sort;
    hipLaunchKernelGGL(cl_keys, dim3(nb_rx), dim3(kRxHistThreads), 0, st, p, keysA, valsA, (uint2 *)recs, (uint4 *)recs,
                       key_bits >= 8u ? 255u : (1u << key_bits) - 1u, nb_rx, hist, rx_totals ? ctx->rx_dtot : (uint32_t *)nullptr);
    uint64_t *kin = nullptr, *kout = nullptr;
    uint32_t *vin = nullptr;
    if (p.idx_packed) radix_sort_pairs(keysA, keysB, nullptr, nullptr, M, key_bits, hist, spart, ctx->rx_dtot, st, &kin, nullptr, &kout, big_sort, true);
    else radix_sort_pairs(keysA, keysB, valsA, valsB, M, key_bits, hist, spart, ctx->rx_dtot, st, &kin, &vin, &kout, big_sort, true);
    p.sorted = vin;
    p.skeys = kin;
    // partitions: one composite scan straight off the sorted keys -> each position's partition id, the partition start
    // list and their number (scal[0]); it also zeroes the work-list counters
    {
        const LoadHead heads{(const uint64_t *)kin, p.centre_bits, p.part_gap, key_mask(key_bits)};
        PartSum *tiles = (PartSum *)tmpA;                         // 3 words per 2048 marks
        hipLaunchKernelGGL(part_reduce, dim3(nb_sc), dim3(kScanThreads), 0, st, heads, M, p.part_max, tiles, scal + 2);
        if (nb_sc <= kSelfSpine && !big_sort) {
            hipLaunchKernelGGL(part_apply<true>, dim3(nb_sc), dim3(kScanThreads), 0, st, heads, M, p.part_max, (const PartSum *)tiles, (uint32_t *)nullptr,
                               part_start, scal);
        } else {
            hipLaunchKernelGGL(part_spine, dim3(1), dim3(1024), 0, st, tiles, nb_sc, p.part_max);
            hipLaunchKernelGGL(part_apply<false>, dim3(nb_sc), dim3(kScanThreads), 0, st, heads, M, p.part_max, (const PartSum *)tiles, (uint32_t *)nullptr,
                               part_start, scal);
        }
    }
    p.part_start = part_start; p.n_parts = scal; p.pc = pc;
    p.e_rec = (uint4 *)ctx->cl_ws[9].ptr;
    const uint32_t grid = M < 16384u ? M : 16384u;               // partitions <= marks; kernels stride over them
    uint32_t *lists = (uint32_t *)ctx->cl_ws[12].ptr;            // [kClasses][M] partitions by size class, then the work lists
    uint32_t *cnts = scal + 2;
    ClWork work;
    work.comp_list = lists + (size_t)kClasses * M;               // [4][M]
    work.rank_list = work.comp_list + 4 * (size_t)M;             // [kClasses][M]
    work.comp_count = scal + 7;
    work.rank_count = scal + 11;
    work.why = nullptr;
    if (getenv("DUET_CL_DEBUG")) {
        work.why = scal + 16;
        HIP_TRY(ctx, hipMemsetAsync(work.why, 0, 64, st));
    }
    p.label8 = (uint8_t *)ctx->cl_ws[13].ptr;
    p.comp8 = p.label8 + M;
    p.inv_norm = (float)(1.0 / pr->normalizer);
    for (int l = 0; l < 3; ++l) {
        p.t_lo[l] = (float)(pr->max_dist / (double)(1 << l) * (1.0 - 1e-5));
        p.t_hi[l] = (float)(pr->max_dist / (double)(1 << l) * (1.0 + 1e-5));
    }
    p.fast = (pr->max_dist >= 0 && pr->max_dist <= 1e6 && pr->normalizer >= 1e-3 && pr->normalizer <= 1e9) ? 1u : 0u;
    if (ctx->dbg & DUET_DBG_CLUSTER_EXACT) p.fast = 0;
    // launches over the partitions, whose number only the device knows: an eighth of the bound (SV-like data: ~10 marks per
    // partition), at least 256 workgroups, striding
    const uint32_t g_parts = std::min((M + 1023u) / 1024u, std::max(256u, (M + 1023u) / 1024u / 8u));
    hipLaunchKernelGGL(cl_classes, dim3(g_parts), dim3(1024), 0, st, p, lists, cnts);
    const uint32_t gridw = M < 32768u ? M : 32768u;
    const bool small = M <= (4u << 20) && !(ctx->dbg & DUET_DBG_CLUSTER_LARGE);
    HIP_TRY(ctx, hipEventRecord(ctx->cl_fork, st));
    const bool cap100 = p.part_max <= 100u;          // no unit has more rows than part_max
    if (small) {
        // partitions of more than 64 marks go whole to the exact pass right away, beside the fast pass over the other size
        // classes (one launch) and the exact pass over what that leaves.  The two chains take about as long as each other; the
        // whole-partition chain stays on the CALLER's stream and the other one goes to the side stream: a cross-stream event
        // that is already signalled when its waiter arrives costs nothing, one that is not costs ~13 us after the signal --
        // so the stream that goes on afterwards should be the one that finishes last, and on SV-like data that is this chain.
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->cl_side[0], ctx->cl_fork, 0));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(cap100 ? cl_exact_big<true, 100> : cl_exact_big<true, 128>), dim3(grid < 4096u ? grid : 4096u),
                           dim3(64), 0, st, p, (const uint32_t *)(lists + 4 * (size_t)M), (const uint32_t *)(cnts + 4));
        hipLaunchKernelGGL(cl_fast_all, dim3(gridw), dim3(64), 0, ctx->cl_side[0], p, (const uint32_t *)lists, (const uint32_t *)cnts, work);
        hipLaunchKernelGGL(cl_exact_small, dim3(grid), dim3(64), 0, ctx->cl_side[0], p, (const uint32_t *)work.comp_list,
                           (const uint32_t *)work.comp_count);
        HIP_TRY(ctx, hipEventRecord(ctx->cl_join[0], ctx->cl_side[0]));
    } else {
        // the side stream takes the partitions of more than 64 marks (few, long chains: a launch of their own would leave most
        // of the chip idle) and then the components of more than 64 rows they leave; beside them one launch per size class
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->cl_side[0], ctx->cl_fork, 0));
        hipLaunchKernelGGL((cl_fast_one<64, 2, 16>), dim3(grid), dim3(64), 0, ctx->cl_side[0], p, (const uint32_t *)(lists + 4 * (size_t)M),
                           (const uint32_t *)(cnts + 4), work);
        HIP_TRY(ctx, hipEventRecord(ctx->cl_join[1], ctx->cl_side[0]));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(cap100 ? cl_exact_big<false, 100> : cl_exact_big<false, 128>), dim3(grid < 4096u ? grid : 4096u),
                           dim3(64), 0, ctx->cl_side[0], p, (const uint32_t *)(work.comp_list + 3 * (size_t)M),
                           (const uint32_t *)(work.comp_count + 3));
        HIP_TRY(ctx, hipEventRecord(ctx->cl_join[0], ctx->cl_side[0]));
        hipLaunchKernelGGL((cl_fast_one<64, 1, 8>), dim3(grid), dim3(64), 0, st, p, (const uint32_t *)(lists + 3 * (size_t)M),
                           (const uint32_t *)(cnts + 3), work);
        hipLaunchKernelGGL((cl_fast_one<32, 1, 4>), dim3(grid), dim3(64), 0, st, p, (const uint32_t *)(lists + 2 * (size_t)M),
                           (const uint32_t *)(cnts + 2), work);
        hipLaunchKernelGGL((cl_fast_one<16, 1, 4>), dim3(grid), dim3(64), 0, st, p, (const uint32_t *)(lists + 1 * (size_t)M),
                           (const uint32_t *)(cnts + 1), work);
        hipLaunchKernelGGL((cl_fast_one<8, 1, 4>), dim3(grid), dim3(64), 0, st, p, (const uint32_t *)lists, (const uint32_t *)(cnts + 0), work);
        HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->cl_join[1], 0));        // cl_exact_small also takes what the >64 class left
        hipLaunchKernelGGL(cl_exact_small, dim3(grid), dim3(64), 0, st, p, (const uint32_t *)work.comp_list,
                           (const uint32_t *)work.comp_count);
    }
    HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->cl_join[0], 0));
    // ranks: the partitions of more than 64 marks come from their class list (small: all of them) or from the fast pass's
    // list of unsettled ones
    hipLaunchKernelGGL(cl_rank_all, dim3(grid), dim3(64), 0, st, p, (const uint32_t *)work.rank_list,
                       (const uint32_t *)work.rank_count,
                       (const uint32_t *)(small ? lists + 4 * (size_t)M : work.rank_list + 4 * (size_t)M),
                       (const uint32_t *)(small ? cnts + 4 : work.rank_count + 4));
    // clusters per partition -> candidate bases (a scan over the partitions; see LoadPc about their count)
    launch_scan<0>(LoadPc{pc, scal}, M, spart, StorePc{cbase, scal}, res->n_cands, st, nullptr, big_sort, scal);       // cbase[part] = its first candidate
    p.cbase = cbase;
    hipLaunchKernelGGL(cl_emit, dim3(std::min(g256.x, std::max(1024u, g256.x / 8u))), b256, 0, st, p);
    HIP_TRY(ctx, hipGetLastError());
    if (getenv("DUET_CL_DEBUG")) {
        uint32_t h[32];
        HIP_TRY(ctx, hipMemcpyAsync(h, scal, sizeof(h), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        fprintf(stderr, "[duet_cluster] parts %u classes %u %u %u %u %u components %u %u %u %u partitions to rank %u %u %u %u %u\n", h[0], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15]);
        fprintf(stderr, "[duet_cluster] gave up (no clean level / too many atoms / decision too close): >64: %u %u %u  33..64: %u %u %u  <=32: %u %u %u;  "
                        "not settled by level 0: %u %u %u\n",
                h[16], h[17], h[18], h[20], h[21], h[22], h[24], h[25], h[26], h[19], h[23], h[27]);
    }
    return DUET_OK;
}

}  // namespace

extern "C" {

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
    if (!pr || !res || !out_pred || !out_ps) return duet_fail(ctx, DUET_ERR_INVALID, "null argument");
    if (!pr->depth_off || pr->depth_bin == 0 || pr->n_contigs == 0 || pr->n_contigs > 65535)
        return duet_fail(ctx, DUET_ERR_INVALID, "bad depth / contig description");
    hipStream_t st = (hipStream_t)stream_;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t M = pr->marks.n_marks, K = pr->n_contigs;
    if (n_cands_host) *n_cands_host = 0;
    if (M == 0) {
        if (res->n_cands) HIP_TRY(ctx, hipMemsetAsync(res->n_cands, 0, 4, st));
        return DUET_OK;
    }
    if (!pr->mark_read || !pr->depth) return duet_fail(ctx, DUET_ERR_INVALID, "null array");
    // workspace: ctg_off + depth_off on the device, the adapted candidate columns, the gathered marks
    int rc;
    const size_t sz[5] = {((size_t)K + 1) * 4 * 2, (size_t)M * 4, (size_t)M * 4, (size_t)M, (size_t)M * 4};
    for (int i = 0; i < 5; ++i)
        if ((rc = duet_reserve(ctx, ctx->sv_ws[i], sz[i]))) return rc;
    uint32_t *d_ctg_off = (uint32_t *)ctx->sv_ws[0].ptr, *d_depth_off = d_ctg_off + (K + 1);
    // (uploaded only when they change: a pageable host-to-device copy in front of every run keeps the host from queueing the
    // run's thirty launches ahead of the device -- 45 us of gaps per 0.37 ms run at 1 M marks)
    if (ctx->sv_depth_off_at != (void *)d_depth_off || ctx->sv_depth_off.size() != (size_t)K + 1 ||
        memcmp(ctx->sv_depth_off.data(), pr->depth_off, ((size_t)K + 1) * 4) != 0) {
        HIP_TRY(ctx, hipStreamSynchronize(st));                  // the previous copy's source is about to change
        ctx->sv_depth_off.assign(pr->depth_off, pr->depth_off + K + 1);
        HIP_TRY(ctx, hipMemcpyAsync(d_depth_off, ctx->sv_depth_off.data(), ((size_t)K + 1) * 4, hipMemcpyHostToDevice, st));
        ctx->sv_depth_off_at = (void *)d_depth_off;
    }
    // clustering; its emit kernel also writes what a caller VCF would have carried (support, reference reads, GT)
    // and the marks' read indices in output order
    SvExtra sv;
    sv.mark_in = pr->mark_read; sv.depth = pr->depth; sv.depth_off = d_depth_off; sv.depth_bin = pr->depth_bin;
    sv.mark_out = (uint32_t *)ctx->sv_ws[4].ptr; sv.svread = (uint32_t *)ctx->sv_ws[1].ptr;
    sv.refread = (uint32_t *)ctx->sv_ws[2].ptr; sv.gt = (uint8_t *)ctx->sv_ws[3].ptr;
    if ((rc = cluster_run(ctx, &pr->marks, res, st, &sv))) return rc;
    if (!n_cands_host) {
        // fully asynchronous: E/F is planned on the device from the candidates' contig column; buffers and grids are
        // sized for the upper bound (a candidate has at least one mark) and the kernels read the real count
        duet_ef_problem ef;
        memset(&ef, 0, sizeof(ef));
        ef.n_contigs = K; ef.n_cands = M; ef.n_marks = M; ef.n_reads = pr->n_reads;
        ef.read_tag = pr->read_tag;
        ef.cand_pos = res->cand_pos; ef.cand_svlen = res->cand_span; ef.cand_svread = sv.svread; ef.cand_refread = sv.refread;
        ef.cand_gt_ok = sv.gt; ef.cand_off = res->cand_off; ef.mark_read = sv.mark_out;
        ef.svlen_thres = pr->svlen_thres; ef.suppread_thres = pr->suppread_thres;
        return duet_ef_run_planned_on_device(ctx, &ef, M, (const uint32_t *)res->n_cands, nullptr, (const uint16_t *)res->cand_contig,
                                             out_pred, out_ps, st);
    }
    hipLaunchKernelGGL(sv_contig_offsets, dim3((K + 1 + 255) / 256), dim3(256), 0, st, (const uint16_t *)res->cand_contig,
                       (const uint32_t *)res->n_cands, K, d_ctg_off);
    std::vector<uint32_t> ctg_off(K + 1);
    HIP_TRY(ctx, hipMemcpyAsync(ctg_off.data(), d_ctg_off, ((size_t)K + 1) * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));             // the one host round trip: candidates per contig
    const uint32_t N = ctg_off[K];
    *n_cands_host = N;
    if (N == 0) return DUET_OK;
    duet_ef_problem ef;
    memset(&ef, 0, sizeof(ef));
    ef.n_contigs = K; ef.n_cands = N; ef.n_marks = M; ef.n_reads = pr->n_reads;
    ef.cand_ctg_off = ctg_off.data();
    ef.read_tag = pr->read_tag;
    ef.cand_pos = res->cand_pos; ef.cand_svlen = res->cand_span; ef.cand_svread = sv.svread; ef.cand_refread = sv.refread;
    ef.cand_gt_ok = sv.gt; ef.cand_off = res->cand_off; ef.mark_read = sv.mark_out;
    ef.svlen_thres = pr->svlen_thres; ef.suppread_thres = pr->suppread_thres;
    return duet_ef_run_device(ctx, &ef, out_pred, out_ps, st);
}

}  // extern "C"
