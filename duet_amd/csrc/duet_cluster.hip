// duet_cluster.hip -- gfx950 kernels and C ABI for stage A0: span-position clustering of SV marks into
// candidates (what `--cluster_max_distance` controls).
//
// The reference delegates this stage to the external `svim alignment` binary (src/duet/sv_calling.py:13-15),
// so there is no reference code to follow; the rule implemented here is this repository's own deterministic
// statement of the published SVIM 1.4.2 scheme, normative text in oracle/cluster_oracle.c / DESIGN.md section 9.
//
// Pipeline (DESIGN.md section 9 has the rule, the argument and the measurements):
//   record sort    (round 4, from 1.25 M marks on; duet_recsort.hip.h) the 16-byte mark record travels with the key -- LSD passes
//                  over the key's top bits, the low bits ordered group by group --, so that everything behind the sort reads its
//                  rows where they lie: no permutation, no gather.  Else:
//   cl_keys        key = (contig, type, centre = pos + span/2) packed into the fewest bits, the mark index in the spare
//                  bits above them when it fits (else a separate value array); (pos, span[, read index]) side by side
//   radix sort     stable LSD passes, 8-bit digits: rx_hist -> tile offsets -> rx_scatter per pass (ballot-ranked, no atomics
//                  on the data path, so the order is deterministic); keys only when the index rides in the key.  Where that
//                  saves two passes the passes cover the top 16 / 24 key bits only and the keys that agree in them -- a few to
//                  a few dozen -- are ordered by their low bits where they lie (rx_local: a rank count in LDS; rx_big)
//   partitions     one scan over a composite element straight off the sorted keys: natural partition starts (contig/type
//                  change or centre gap > part_gap), a partition there and every part_max marks after -> the partition
//                  start list and the first partition of every 2048-position tile
//   cl_box         one workgroup per tile: takes the tile's rows (record sort) or gathers them through the sort permutation; on large
//                  inputs it also numbers the partitions that start in its tile (no part_apply launch); finishes the partitions whose
//                  bounding box proves ONE cluster (most, on SV-like data), lays the others' rows out in sorted order and
//                  lists them by size class (<= 8 / 16 / 32 / 64 marks; > 64: cl_tight_big finds them itself, on a side stream beside cl_box)
//   agglomeration  GROUP lanes per partition, one lane per mark; units that hold one or two partitions per wave evaluate every
//                  UNORDERED pair once (the compare masks rotated on the scalar unit hand the result to the pair's other row).
//                  small inputs: cl_fast_all -- the threshold graph and, in the same wavefront, the exact linkage on the full
//                  triangle of sums for what that does not settle (fast_unit, link_unit);
//                  large inputs: the contracted linkage in two tiers (tight_unit: threshold graph, tight groups, k x k sums,
//                  rounds on the groups; 5 KB of LDS per wavefront instead of 21), one launch per size class, what a tier does
//                  not take on the class's second list; the partitions of more than 64 marks take this path at every size
//                  every path writes each mark to its place in the partition's output (order[], and in the fused pipeline its
//                  read index) and leaves, per cluster, a record (rank, end, floor means) at the partition's start + the
//                  cluster's index (emit_prep)
//   scan + cl_emit clusters per partition -> candidate bases; one LANE per cluster turns the clusters' records into cand_*[]
//
// Bit-exactness vs the oracle: the rule works on integers (distances in fixed point relative to the threshold, cluster
// distances exact means), so what is emitted does not depend on the order in which provably-first merges are made; the
// binary32 pair tests only decide what lies outside their guard bands, the rest is evaluated in binary64 exactly as the
// oracle does (-ffp-contract=off).  All paths produce the oracle's clusters; tests/test_gpu_cluster.py and tools/stress.py
// run every case through every path (DUET_DBG_CLUSTER_* in include/duet_ef.h).

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
#include "duet_recsort.hip.h"


typedef float float2v __attribute__((ext_vector_type(2)));

// Diagnostic build (-DDUET_STAMPS, tools/stamps_cl.py): a log of (tag, wall clock) per workgroup of the agglomeration kernels -- the
// partitions are chains of dependent steps on one wavefront, and what a chain spends where does not show in a kernel trace.
#ifdef DUET_STAMPS
__shared__ uint32_t g_clst_area, g_clst_n;
#define CL_STAMP_INIT(area) do { if (threadIdx.x == 0) { g_clst_area = (area); g_clst_n = 0; } } while (0)
#define CL_STAMP(tag) do { if (threadIdx.x == 0 && p.stamps && g_clst_area) { const uint32_t n_ = g_clst_n++; \
    if (n_ < 62u && blockIdx.x < 8192u) p.stamps[((size_t)g_clst_area * 65536 + (size_t)blockIdx.x * 8) * 8 + 2 + n_] = \
        ((unsigned long long)(tag) << 56) | ((unsigned long long)wall_clock64() & 0x00FFFFFFFFFFFFFFull); } } while (0)
#else
#define CL_STAMP_INIT(area) do { } while (0)
#define CL_STAMP(tag) do { } while (0)
#endif

struct ClParams {
    unsigned long long *stamps;                       // (diagnostic build only, else null)
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
    uint32_t rec_mode;                                // the sort carried the records (duet_recsort.hip.h): srec holds EVERY sorted position's record,
                                                      // w = (contig | type) << idx_bits | mark index; no sorted keys, no permutation
    uint32_t idx_bits, idx_mask;                      // (idx_mask = ~0 outside rec_mode: srec's w is the mark index itself)
    const uint32_t *part_start;                       // [P+1]
    const uint32_t *n_parts;                          // device scalar
    float inv_norm, t_lo[3], t_hi[3];                 // binary32 pair tests: 1/normalizer; level * (1 -/+ 1e-5) for the levels max_dist, / 2, / 4
    float2v t_hl[3];                                  // (t_hi, t_lo) side by side: one packed subtract and multiply per level
    uint32_t kc;                                      // contracted linkage: the first tier's group limit is min(kc, the variant's own): tests lower it
    uint32_t fast;                                    // 0: parameters outside the fast pass's vetted range, everything goes to the exact linkage
    uint32_t box;                                     // 0: no bounding-box test (tests: every partition through the pair loops)
    double invn, scale;                               // exact linkage: 1 / normalizer, 2^26 / max_dist (oracle/cluster_oracle.c, rule 3)
    uint32_t mergeable;                               // max_dist >= 0
    uint32_t tps;                                     // scan tiles per work-list shard (kShards shards of consecutive tiles)
    uint32_t gather_rows;                             // this launch reads its rows through the sort permutation (no cl_box before it)
    uint32_t sym;                                     // one-partition-per-wave units evaluate every unordered pair once (0: DUET_DBG_CLUSTER_NOSYM, the column loop)
    // fused SVIM-mode pipeline (all null otherwise): cl_emit also writes the columns ef_classify reads
    const uint32_t *sv_mark_in, *sv_depth, *sv_depth_off;
    uint32_t sv_depth_bin;
    uint32_t *sv_mark_out, *sv_svread, *sv_refread;
    uint8_t *sv_gt;
    uint32_t *ef_ctg_off, *ef_zero;                   // ... and step E/F's plan: first candidate of every contig [n_contigs + 1], n_contigs + 8 words to zero
    uint32_t n_contigs;
    const uint32_t *n_cands;                          // (device scalar: the candidates' number, there before cl_emit starts)
    uint4 *srec;                                      // [M] per sorted position of a partition the box test left open: (pos, span, read index, mark index)
    uint4 *e_rec;                                     // [M] cluster c >= 1 of the partition that starts at s, at s + c: (rank | end << 8, floor mean pos, floor mean span,
                                                      // contig | type in rec_mode)
    uint4 *e_first;                                   // [P] cluster 0 of every partition, by partition number: most partitions have one cluster, and
                                                      // cl_emit reads these records side by side instead of one 16-byte record per line
    uint32_t *pc;                                     // [P] clusters per partition
    const uint32_t *csum, *tsum;                      // clusters per 64 partitions / per 2048 (cl_pc_sums): cl_emit numbers the candidates from them
    uint32_t *n_cands_w;                              // ... and leaves their number here
    // outputs
    uint32_t *order, *cand_off, *cand_pos, *cand_span;
    uint16_t *cand_contig;
    uint8_t *cand_type;
    // a fork without an event (cl_gate): the kernel in front of which the side stream forks off writes the run's epoch here as it starts
    uint32_t *fork_flag;
    uint32_t fork_epoch;
    // small inputs, the partitions of 33..64 marks on four wavefronts each (cl_wide_list on the side stream): cl_fast_all says when it
    // starts (cl_box is through then) and takes the classes of fast_classes only
    uint32_t *box_done_flag;
    uint32_t fast_classes;
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
__global__ __launch_bounds__(kRxHistThreads) void cl_keys(const ClParams p, uint64_t *keys, uint32_t *vals, uint2 *ps, uint4 *rec4, uint32_t dshift,
                                                          uint32_t dmask, uint32_t nb, uint32_t *hist /* [256][nb] */, uint32_t *dtot /* or null */,
                                                          uint32_t *zero /* a counter of a later launch */)
{
    __shared__ uint32_t s_h[256];
    const uint32_t tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) *zero = 0;
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
            atomicAdd(&s_h[(uint32_t)(key >> dshift) & dmask], 1u);
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
// (E: the sorted elements -- 8-byte keys, or the 16-byte records where the sort carried them; keyof: an element's sort key)
template <class E, class KeyOf>
struct LoadHead {
    const E *keys;
    uint32_t centre_bits, part_gap;
    KeyOf keyof;
};
// head[j] = position base + j starts a natural partition, for the kScanItems consecutive positions tid * kScanItems + j of one
// scan thread of the tile at t0.  The elements are read side by side (thread tid takes the positions j * kScanThreads + tid: a
// wave's load is one run of memory, where eight consecutive 16-byte records per lane were 64 different lines per instruction),
// the flags change hands in LDS; the tile's flags also go out as one byte per thread (hbits[t0 / 8 + tid], bit j): part_apply
// then reads 1 bit per mark instead of the elements again.
template <class E, class KeyOf>
__device__ __forceinline__ void load_heads(const LoadHead<E, KeyOf> &h, uint32_t t0, uint32_t n, bool (&head)[kScanItems], uint8_t *s_f /* LDS [kScanTile] */,
                                           uint8_t *hbits)
{
    static_assert(kScanItems == 8, "one flag byte per thread");
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    __shared__ uint64_t s_edge[kScanItems][kScanThreads / 64];      // every wave's last key, per round of loads
    E e[kScanItems];
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) e[j] = h.keys[min(t0 + j * kScanThreads + tid, n - 1u)];
    const E before = h.keys[(tid == 0 && t0 > 0u) ? t0 - 1u : 0u];   // (leaves with the others: no round trip of its own)
    const uint64_t cm = (1ull << h.centre_bits) - 1ull;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j)
        if (lane == 63u) s_edge[j][wave] = h.keyof(e[j]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        const uint32_t i = t0 + j * kScanThreads + tid;
        const uint64_t b = h.keyof(e[j]);
        // the key in front: the lane in front holds it; lane 0 has it from the wave in front
        uint64_t a = ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(b >> 32), 1, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)b, 1, 64);
        if (lane == 0) a = wave > 0 ? s_edge[j][wave - 1u] : (j > 0 ? s_edge[j > 0 ? j - 1 : 0][kScanThreads / 64 - 1] : h.keyof(before));
        s_f[j * kScanThreads + tid] = (i < n && (i == 0 || (a >> h.centre_bits) != (b >> h.centre_bits) || (b & cm) - (a & cm) > (uint64_t)h.part_gap)) ? 1 : 0;
    }
    __syncthreads();
    const uint2 f = *reinterpret_cast<const uint2 *>(s_f + tid * kScanItems);
    uint32_t byte = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        head[j] = (((j < 4 ? f.x : f.y) >> (8 * (j & 3))) & 1u) != 0;
        byte |= head[j] ? 1u << j : 0u;
    }
    hbits[t0 / kScanItems + tid] = (uint8_t)byte;
}
__device__ __forceinline__ void load_head_bits(const uint8_t *hbits, uint32_t t0, bool (&head)[kScanItems])
{
    const uint32_t byte = hbits[t0 / kScanItems + threadIdx.x];
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) head[j] = (byte >> j) & 1u;
}

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

template <class E, class KeyOf>
__global__ __launch_bounds__(kScanThreads) void part_reduce(const LoadHead<E, KeyOf> in, uint32_t n, uint32_t pm, PartSum *tiles, uint32_t *zero, uint32_t nzero,
                                                            uint8_t *hbits /* [tiles * kScanThreads] one flag byte per scan thread */)
{
    __shared__ PartSum s_w[kScanThreads / 64 + 1];
    __shared__ __align__(8) uint8_t s_f[kScanTile];
    const uint32_t tid = threadIdx.x;
    if (zero && blockIdx.x == 0)
        for (uint32_t i = tid; i < nzero; i += kScanThreads) zero[i] = 0;     // the work-list counters of the kernels that follow
    // The tiles are taken in DESCENDING order: the elements have just been written front to back by the sort's last stage, and
    // what the 256 MB Infinity Cache still holds of a 320 MB array is its END -- a scan from the front evicts, tile by tile, what
    // it is about to read (LRU) and every tile comes from HBM; from the back four tiles in five are hits (2e7 marks: 76 -> 57 us;
    // a plain streaming read of the same bytes takes 48, tools/probe/bw_probe.hip).  Nothing depends on the order of the tiles.
    const uint32_t tile = gridDim.x - 1u - blockIdx.x;
    const uint32_t base = tile * kScanTile + tid * kScanItems;
    PartSum acc{kNoHead, 0, 0};
    bool head[kScanItems];
    load_heads(in, tile * kScanTile, n, head, s_f, hbits);
#pragma unroll
    for (int j = 0; j < kScanItems; ++j)
        if (head[j]) acc = part_combine(acc, PartSum{base + j, base + j, 0}, pm);
    (void)part_block_exscan<kScanThreads>(acc, pm, s_w);
    __syncthreads();
    if (tid == 0) tiles[tile] = s_w[kScanThreads / 64];
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
__global__ __launch_bounds__(kScanThreads) void part_apply(const uint8_t *hbits, uint32_t n, uint32_t pm, const PartSum *tiles, uint32_t *pid,
                                                           uint32_t *part_start, uint32_t *n_parts, uint32_t *tile_first /* [tiles + 1]: the first
                                                           partition that starts in each tile (cl_box owns a tile's partitions) */)
{
    __shared__ PartSum s_w[kScanThreads / 64 + 1];
    __shared__ PartSum s_c[kScanThreads / 64];
    __shared__ uint32_t s_first;
    if (threadIdx.x == 0) s_first = 0xFFFFFFFFu;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t base = blockIdx.x * kScanTile + tid * kScanItems;
    bool head[kScanItems];
    PartSum acc{kNoHead, 0, 0};
    load_head_bits(hbits, blockIdx.x * kScanTile, head);
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
    uint32_t last = 0xFFFFFFFFu;
    bool seen = false;
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
        if (starts) {
            part_start[P + q] = i;
            if (!seen) atomicMin(&s_first, P + q);
            seen = true;
        }
        if (i == n - 1) {
            part_start[P + q + 1] = n;
            *n_parts = P + q + 1;
            last = P + q + 1;
        }
    }
    __syncthreads();
    // (a tile holds kScanTile positions and a partition at most 128: every full tile has starts; the last tile may have none)
    if (last != 0xFFFFFFFFu) {
        tile_first[blockIdx.x + 1] = last;
        if (s_first == 0xFFFFFFFFu) tile_first[blockIdx.x] = last;
    }
    if (tid == 0 && s_first != 0xFFFFFFFFu) tile_first[blockIdx.x] = s_first;
}

// ---------------------------------------------------------------------------------------------
// partitions: the box test and the work lists
// ---------------------------------------------------------------------------------------------

// work lists by partition size (which agglomeration kernel variant takes it); list order is irrelevant --
// every partition writes to its own fixed output range -- so a (wave-aggregated) atomic append is fine.
// A list is kept in kShards pieces, one per run of consecutive scan tiles, each with its own counter (ten thousand workgroups
// adding to ONE counter queue up behind each other: 40 ns apiece, 0.4 ms at 2e7 marks); the piece of a shard starts at the
// shard's first position -- it has room, a shard cannot hold more partitions than positions.
constexpr int kClasses = 5;
constexpr int kShards = 64;
struct WorkList {
    const uint32_t *items;          // the class's list, [M]
    const uint32_t *pref;           // (LDS) pref[s] = items in the shards before s, pref[kShards] = all
    uint32_t shard_span;            // positions per shard
    uint32_t rev_end;               // 0: a shard's piece grows up from its first position; M: down from its last one (see over_append)
    __device__ __forceinline__ uint32_t size() const { return items ? pref[kShards] : shard_span; }
    __device__ __forceinline__ uint32_t operator[](uint32_t i) const
    {
        if (!items) return i;                              // (no list: the items are the numbers 0 .. shard_span - 1 themselves)
        uint32_t lo = 0, hi = kShards;                     // pref[lo] <= i < pref[hi]
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (pref[mid] <= i) lo = mid; else hi = mid;
        }
        if (rev_end) {
            const uint64_t e = (uint64_t)(lo + 1u) * shard_span;
            return items[(e < rev_end ? (size_t)e : (size_t)rev_end) - 1u - (i - pref[lo])];
        }
        return items[(size_t)lo * shard_span + (i - pref[lo])];
    }
};
// A partition the contracted linkage hands on (too many groups left, parameters or coordinates outside what it vouches for) goes
// on the class's SECOND list, kept in the same array: a listed partition has at least two marks, so a shard's piece fills at
// most half of the shard's positions from below, and the second list grows down from the shard's last position -- at most as
// many entries as the first one has.
__device__ __forceinline__ void over_append(const ClParams &p, uint32_t *items, uint32_t *over_counts, uint32_t part, uint32_t s)
{
    const uint32_t span = p.tps * kScanTile, shard = (s / kScanTile) / p.tps;
    const uint64_t e = (uint64_t)(shard + 1u) * span;
    const uint32_t end = e < p.M ? (uint32_t)e : p.M;
    items[end - 1u - atomicAdd(&over_counts[shard], 1u)] = part;
}
// one wavefront: the running sums of a class's kShards counters into LDS
__device__ __forceinline__ void worklist_prefix(const uint32_t *counts, uint32_t *pref /* LDS [kShards + 1] */)
{
    static_assert(kShards == 64, "one lane per shard");
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t x = counts[lane];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if ((int)lane >= d) x += y;
    }
    pref[lane + 1] = x;
    if (lane == 0) pref[0] = 0;
}
__device__ __forceinline__ int size_class(uint32_t n) { return n <= 8 ? 0 : (n <= 16 ? 1 : (n <= 32 ? 2 : (n <= 64 ? 3 : 4))); }

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

// Fixed point of the rule (oracle/cluster_oracle.c, rule 3): the threshold is 2^26
constexpr uint64_t kQOne = 1ull << 26;
constexpr double kQCap = 2199023255552.0;          // 2^41

// cl_box: one workgroup per scan tile (kScanTile sorted positions); it owns the partitions that START in its tile.
// A partition whose bounding box already proves that every pair of its marks is closer than max_dist,
//     min(range of pos, range of end, range of centre) / normalizer + (1 - min span / max span) <= max_dist * (1 - 1e-5)
// (every pair distance is bounded by this expression term by term), is ONE cluster: all its pairs are within the threshold, so
// average linkage keeps merging until one cluster is left.  On SV-like data that is the common case (a partition = the marks of
// one SV): such partitions are finished here -- their marks keep their sorted order, one cluster record -- and never see
// a pair loop.  The others go on the work lists by size class.
constexpr int kBoxThreads = 256;
constexpr int kBoxHalo = 128;                      // a partition has at most 128 marks: the last one of a tile ends within the halo
// REC: the sort carried the records -- the tile's rows are p.srec[t0 ...] as they lie (no permutation, no gather), and the rows of
// the partitions left open stay where they are for the agglomeration kernels
// APPLY (large inputs): the launch is also the partition scan's last stage -- it takes the tile's carry from part_spine and the head
// flags part_reduce left, numbers the partitions that start in its tile itself and writes their starts (what part_apply does
// for small inputs): no launch in between, and the tile's records leave together with the carry and the flags instead of
// behind two dependent round trips (first partition of the tile -> its partitions' starts).
template <bool REC, bool APPLY>
__global__ __launch_bounds__(kBoxThreads) void cl_box(const ClParams p, const uint32_t *tile_first, uint32_t *lists /* [kClasses][M] */,
                                                      uint32_t *counts /* [kClasses][kShards] */, const uint8_t *hbits, const PartSum *tiles,
                                                      uint32_t *part_start_out, uint32_t *n_parts_out)
{
    __shared__ uint32_t s_pos[kScanTile + kBoxHalo], s_span[kScanTile + kBoxHalo];
    // starts of the tile's partitions (+ the end of the last one) relative to the tile's first position (< 2048 + 128), bit 15: the
    // partition is finished here.  (16 bits and no separate flag array: 26 KB of LDS per workgroup instead of 32, six per CU)
    __shared__ uint16_t s_ps[kScanTile + 2];
    // per position: its partition is finished here (a bit; the partition's thread sets the bits of its run of positions word by
    // word).  With the 16-bit starts: 22 KB of LDS per workgroup, seven per CU (round 3: 32 KB, five)
    __shared__ uint32_t s_dbit[(kScanTile + kBoxHalo + 31) / 32];
    __shared__ uint32_t s_cnt[kClasses], s_base[kClasses];
    __shared__ PartSum s_w[kBoxThreads / 64 + 1];
    __shared__ uint32_t s_first, s_count, s_next;
    static_assert(!APPLY || (kBoxThreads == kScanThreads && kBoxHalo == 128), "the scan's thread layout; a partition ends within the halo");
    const uint32_t tid = threadIdx.x, lane = tid & 63u, tile = blockIdx.x;
    const uint32_t t0 = tile * kScanTile, shard = tile / p.tps;
    // (small inputs: the side stream's chain of the > 64-mark partitions starts BESIDE this kernel -- everything in front of it on the
    // stream is done when any of its workgroups runs: the first one says so, cl_gate on the side stream is waiting for it)
    if (!APPLY && p.fork_flag && tile == 0 && tid == 0) __hip_atomic_store(p.fork_flag, p.fork_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t p_lo = 0, np = 0;
    if (tid < kClasses) s_cnt[tid] = 0;
    if (tid < (kScanTile + kBoxHalo + 31) / 32) s_dbit[tid] = 0;
    uint32_t hb = 0;
    PartSum carry{kNoHead, 0, 0};
    if (APPLY) {
        hb = hbits[t0 / kScanItems + tid];
        carry = tiles[tile];
        if (tid == 0) { s_first = 0xFFFFFFFFu; s_count = 0; }
    } else {
        p_lo = tile_first[tile];
        np = tile_first[tile + 1] - p_lo;
        for (uint32_t j = tid; j <= np; j += kBoxThreads) s_ps[j] = (uint16_t)(p.part_start[p_lo + j] - t0);
    }
    // the marks' (pos, span), gathered through the sort permutation; each thread keeps its eight positions' mark indices
    // (and read indices) for the stores at the end.  Two rounds of loads, each round's loads side by side (the branches on the
    // layout are hoisted: with them inside, every position waited for its own two round trips in turn -- 30 us per tile)
    uint32_t mk[kScanItems + 1], rd[kScanItems + 1], ps_[kScanItems + 1], sp_[kScanItems + 1];
    {
        uint32_t at[kScanItems + 1];
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) at[j] = min(t0 + j * kBoxThreads + tid, p.M - 1u);
        at[kScanItems] = min(t0 + kScanTile + min(tid, (uint32_t)kBoxHalo - 1u), p.M - 1u);      // the halo (threads < kBoxHalo)
        if (REC) {
#pragma unroll
            for (int j = 0; j <= kScanItems; ++j) {
                const uint4 q = p.srec[at[j]];
                ps_[j] = q.x; sp_[j] = q.y; rd[j] = q.z; mk[j] = q.w & p.idx_mask;
            }
        } else if (p.idx_packed) {
#pragma unroll
            for (int j = 0; j <= kScanItems; ++j) mk[j] = (uint32_t)(p.skeys[at[j]] >> p.key_bits);
        } else {
#pragma unroll
            for (int j = 0; j <= kScanItems; ++j) mk[j] = p.sorted[at[j]];
        }
        if (REC) {
        } else if (p.rec4) {
#pragma unroll
            for (int j = 0; j <= kScanItems; ++j) {
                const uint4 q = p.rec4[mk[j]];
                ps_[j] = q.x; sp_[j] = q.y; rd[j] = q.z;
            }
        } else {
#pragma unroll
            for (int j = 0; j <= kScanItems; ++j) {
                const uint2 q = p.ps[mk[j]];
                ps_[j] = q.x; sp_[j] = q.y; rd[j] = 0;
            }
        }
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) {
            const uint32_t i = t0 + j * kBoxThreads + tid;
            if (i < p.M) { s_pos[i - t0] = ps_[j]; s_span[i - t0] = sp_[j]; }
        }
        if (tid < kBoxHalo && t0 + kScanTile + tid < p.M) { s_pos[kScanTile + tid] = ps_[kScanItems]; s_span[kScanTile + tid] = sp_[kScanItems]; }
    }
    if (APPLY) {
        // the partitions that start in this tile: part_apply's walk, the starts into s_ps (and out to part_start[])
        const uint32_t n = p.M, pm = p.part_max, base = t0 + tid * kScanItems;
        bool head[kScanItems];
        PartSum acc{kNoHead, 0, 0};
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) {
            head[j] = (hb >> j) & 1u;
            if (head[j]) acc = part_combine(acc, PartSum{base + j, base + j, 0}, pm);
        }
        const PartSum before = part_block_exscan<kBoxThreads>(acc, pm, s_w);      // (synchronises)
        const PartSum st = part_combine(carry, before, pm);
        const uint32_t H0 = st.f == kNoHead ? 0u : st.l, P0 = st.f == kNoHead ? 0u : st.s;
        uint32_t H = H0, P = P0, mine = 0, first = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) {
            const uint32_t i = base + j;
            if (i >= n) break;
            if (head[j]) {
                if (i) P += parts_in(i - H, pm);
                H = i;
            }
            const uint32_t d = i - H, q = d < pm ? 0u : d / pm;
            if (d == q * pm) {
                first = min(first, P + q);
                ++mine;
            }
            if (i == n - 1) {
                part_start_out[P + q + 1] = n;
                *n_parts_out = P + q + 1;
            }
        }
        if (mine) {
            atomicMin(&s_first, first);
            atomicAdd(&s_count, mine);
        }
        // the end of the tile's last partition: the next natural start among the 128 positions behind the tile, the next multiple
        // of part_max in the running natural partition, or the end of the marks
        if (tid == kBoxThreads - 1) {
            const uint32_t e = min(t0 + (uint32_t)kScanTile, n);                 // first position behind the tile
            uint32_t nx = n;
            if (e < n) {
                const uint32_t d = e - H;                                         // (H: the natural start in force at the tile's last position)
                nx = min(nx, H + ((d + pm - 1u) / pm) * pm);
                const uint4 hw = *reinterpret_cast<const uint4 *>(hbits + e / kScanItems);   // (e is a multiple of 2048: 16-byte aligned)
                const uint32_t w4[4] = {hw.x, hw.y, hw.z, hw.w};
#pragma unroll
                for (int k = 3; k >= 0; --k)
                    if (w4[k]) nx = min(nx, e + 32u * k + (uint32_t)__ffs((int)w4[k]) - 1u);
            }
            s_next = nx;
        }
        __syncthreads();
        p_lo = s_first;
        np = s_count;
        H = H0; P = P0;
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) {
            const uint32_t i = base + j;
            if (i >= n) break;
            if (head[j]) {
                if (i) P += parts_in(i - H, pm);
                H = i;
            }
            const uint32_t d = i - H, q = d < pm ? 0u : d / pm;
            if (d == q * pm) {
                s_ps[P + q - p_lo] = (uint16_t)(i - t0);
                part_start_out[P + q] = i;
            }
        }
        if (tid == 0) s_ps[np] = (uint16_t)(s_next - t0);
    }
    __syncthreads();
    // one thread per partition
    for (uint32_t j0 = 0; j0 < np; j0 += kBoxThreads) {
        const uint32_t j = j0 + tid;
        const bool live = j < np;
        int cls = -1;
        if (live) {
            const uint32_t s = t0 + (s_ps[j] & 0x7FFFu), e = t0 + (s_ps[j + 1] & 0x7FFFu), n = e - s;
            uint32_t plo = ~0u, phi = 0, elo = ~0u, ehi = 0, clo = ~0u, chi = 0, slo = ~0u, shi = 0;
            uint64_t sum_p = 0, sum_s = 0;
            bool bad = false;
            for (uint32_t i = s - t0; i < e - t0; ++i) {
                const uint32_t ps = s_pos[i], sp = s_span[i], en = ps + sp, ce = ps + (sp >> 1);
                bad = bad || en < ps;                        // end does not fit 32 bits: leave it to the exact path
                plo = min(plo, ps); phi = max(phi, ps);
                elo = min(elo, en); ehi = max(ehi, en);
                clo = min(clo, ce); chi = max(chi, ce);
                slo = min(slo, sp); shi = max(shi, sp);
                sum_p += ps;
                sum_s += sp;
            }
            const uint32_t r = min(min(phi - plo, ehi - elo), chi - clo);
            const float u = (float)r * p.inv_norm + (float)(shi - slo) * __builtin_amdgcn_rcpf((float)max(shi, 1u));
            // (partitions of more than 64 marks belong to the launch that started beside this one: cl_tight_big)
            const bool one = n < 2 || (n <= 64u && p.fast && p.box && !bad && u <= p.t_lo[0]);
            if (one) s_ps[j] = (uint16_t)(s_ps[j] | 0x8000u);
            if (one) {
                // the positions s - t0 .. e - t0 - 1 are done: at most five words of the bit array (a partition has <= 128 marks)
                for (uint32_t a = s - t0, b = e - t0; a < b;) {
                    const uint32_t w = a >> 5, hi = min(b, (w + 1u) << 5);
                    const uint32_t m = (hi - a == 32u ? 0xFFFFFFFFu : ((1u << (hi - a)) - 1u) << (a & 31u));
                    atomicOr(&s_dbit[w], m);
                    a = hi;
                }
                // (floor means: see emit_prep)
                // (w: contig | type where the sort carried the records -- the tile's rows have just been read, this word is in
                // cache; cl_emit then needs nothing of a partition but its cluster records)
                p.e_first[p_lo + j] = make_uint4(0u | (n << 8), (uint32_t)((double)sum_p / (double)n), (uint32_t)((double)sum_s / (double)n),
                                                 REC ? p.srec[s].w >> p.idx_bits : 0u);
                p.pc[p_lo + j] = 1;
            } else {
                cls = size_class(n);
                if (cls == 4) cls = -1;                      // (cl_tight_big / cl_find_big finds them itself: its launch starts before this kernel is done)
            }
        }
        // append the others to their class lists (wave-aggregated)
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
        if (tid < kClasses && s_cnt[tid]) s_base[tid] = shard * p.tps * kScanTile + atomicAdd(&counts[tid * kShards + shard], s_cnt[tid]);
        __syncthreads();
        if (cls >= 0) lists[(size_t)cls * p.M + s_base[cls] + at] = p_lo + j;
        __syncthreads();
        if (tid < kClasses) s_cnt[tid] = 0;
        __syncthreads();
    }
    // the finished partitions' marks keep their sorted order; the marks of the others are laid out in sorted order for the
    // agglomeration kernels, which then read their rows side by side instead of gathering them again.  (Positions before the
    // tile's first start belong to the previous tile's last partition: its block handles them as its halo.)
    const uint32_t first = np ? t0 + (s_ps[0] & 0x7FFFu) : 0xFFFFFFFFu, end = np ? t0 + (s_ps[np] & 0x7FFFu) : 0u;
#pragma unroll
    for (int j = 0; j <= kScanItems; ++j) {
        const uint32_t i = j < kScanItems ? t0 + j * kBoxThreads + tid : t0 + kScanTile + tid;
        if ((j == kScanItems && tid >= kBoxHalo) || i < first || i >= end) continue;
        if ((s_dbit[(i - t0) >> 5] >> ((i - t0) & 31u)) & 1u) {
            p.order[i] = mk[j];
            if (p.sv_mark_out) p.sv_mark_out[i] = rd[j];
        } else if (!REC) {
            p.srec[i] = make_uint4(ps_[j], sp_[j], rd[j], mk[j]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// agglomeration
// ---------------------------------------------------------------------------------------------

// |a - b| in ONE instruction (__usad compiles to min, max, sub)
__device__ __forceinline__ uint32_t absdiff_u32(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

template <int NW>
struct BitSet {
    uint64_t w[NW];
    __device__ __forceinline__ void clear() { for (int i = 0; i < NW; ++i) w[i] = 0; }
    __device__ __forceinline__ bool any() const { uint64_t x = 0; for (int i = 0; i < NW; ++i) x |= w[i]; return x != 0; }
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
    __device__ __forceinline__ uint32_t pop_first()                        // ... and clears it
    {
        const uint32_t f = first();
        for (int i = 0; i < NW; ++i) w[i] &= (NW == 1 || (f >> 6) == (uint32_t)i) ? ~(1ull << (f & 63u)) : ~0ull;
        return f;
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
    __device__ __forceinline__ void keep_above(uint32_t j)                 // drop the positions <= j
    {
        for (int i = 0; i < NW; ++i) {
            const uint32_t lo = 64u * i;
            const uint64_t m = j >= lo + 63u ? 0ull : (j >= lo ? ~0ull << (j - lo + 1u) : ~0ull);
            w[i] &= m;
        }
    }
    __device__ __forceinline__ bool equals(const BitSet &o) const { bool e = true; for (int i = 0; i < NW; ++i) e = e && w[i] == o.w[i]; return e; }
};

// bit set over a group's rows from one predicate per (lane, r): row sl + r * GROUP
template <int GROUP, int R, int NW>
__device__ __forceinline__ BitSet<NW> group_ballot(const bool (&pred)[R], uint32_t sub)
{
    constexpr unsigned long long gm = GROUP == 64 ? ~0ull : ((1ull << (GROUP & 63)) - 1ull);
    BitSet<NW> b;
    b.clear();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned long long v = (__ballot(pred[r]) >> (sub * GROUP)) & gm;
        b.w[(r * GROUP) >> 6] |= v << ((r * GROUP) & 63);
    }
    return b;
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
                uint4 *at_ = ci ? p.e_rec + (s + ci) : p.e_first + part;
                *at_ = make_uint4(rank | ((before[r] + size) << 8), (uint32_t)((double)s_sum[k][0] / (double)size),
                                  (uint32_t)((double)s_sum[k][1] / (double)size), p.rec_mode ? p.srec[s].w >> p.idx_bits : 0u);
            }
        }
    }
    if (sl == 0) p.pc[part] = heads.count();
}

// ---------------------------------------------------------------------------------------------
// exact average linkage, round by round
// ---------------------------------------------------------------------------------------------
//
// The rule (oracle/cluster_oracle.c, rule 4) is the serial one: merge the closest pair of clusters, ties to the smallest
// (first, second) index pair, while its mean is within the threshold.  Cluster distances are exact means of integers
// (sums of member-pair distances over n_a n_b), so they do not depend on the order of the merges, and average linkage is
// reducible: merging two clusters never brings the result closer to a third one than the nearer of the two was.
// Two clusters that are each other's nearest neighbour (nearest = smallest mean, ties to the smallest index) are therefore
// merged with each other by the serial rule, whatever it does elsewhere first -- nothing can come between them (a third
// cluster merging elsewhere only moves away), and the serial rule cannot stop before their mean, which is within the
// threshold, has been used.  So ALL mutual nearest-neighbour pairs of a round merge at once; the sums of the new clusters
// are sums of the old ones; a dozen rounds settle a partition of a hundred marks where the serial rule takes ninety-nine
// dependent steps.  Marks with the same (pos, span) have distance 0 and go first, as one group.
// tools/linkage_proto.py compares the two evaluations on the CPU, partition by partition.
//
// GROUP lanes of a wavefront work on one partition of up to GROUP * R rows; lane sl owns rows sl, sl + GROUP, ...
// The sums live in the upper triangle of a matrix in LDS.
template <int GROUP, int R, int NCAP>
struct LinkSmem {
    static constexpr int SUBS = 64 / GROUP, NMAX = NCAP, NW = NMAX > 64 ? 2 : 1, NT = NMAX * (NMAX - 1) / 2;
    double S[SUBS][NT + 2];                                  // upper triangle, row by row: exact integers (every sum stays <= 2^53);
                                                             // [NT] = "nothing there" (beyond any sum), [NT + 1] scratch / zero
    double inv[SUBS][NMAX];
    uint32_t pos[SUBS][NMAX], span[SUBS][NMAX];
    uint8_t size[SUBS][NMAX], nn[SUBS][NMAX], rep[SUBS][NMAX];
    uint8_t alist[SUBS][NMAX + 4], asize[SUBS][NMAX + 4];   // the clusters that are left, ascending, and their sizes (read four at a time)
    uint8_t mdst[SUBS][NMAX], msrc[SUBS][NMAX];             // this round's merges
};

// rule 3 of oracle/cluster_oracle.c, the result as a binary64 integer
__device__ __forceinline__ double quantise(double d, double scale)
{
    double t = __builtin_rint(d * scale);
    t = !(t < kQCap) ? kQCap : t;
    t = t < 1.0 ? 1.0 : t;
    return d == 0.0 ? 0.0 : t;
}

// Sums beyond 2^38 cannot be within the threshold (mean <= 2^26 over at most 64 x 64 member pairs) and their mean exceeds that
// of every pair that is: they count as "nothing there", and for the others s * n (n <= 128) stays below 2^53, so the rational
// comparison s1 / n1 < s2 / n2 is two exact binary64 products
constexpr double kFar = 274877906944.0;                    // 2^38
constexpr double kNothing = 1152921504606846976.0;         // 2^60

// rows of the partition -> F[r]: the final cluster of each of this lane's rows (groups with go == false keep the
// collective operations company).  pk / spk: this lane's rows' (pos, span); wide: some end position needs 33 bits.
template <int GROUP, int R, int NCAP, int NW>
__device__ __forceinline__ void link_unit(const ClParams &p, bool go, uint32_t n, uint32_t sub, uint32_t sl, const uint32_t (&pk)[R],
                                          const uint32_t (&spk)[R], bool wide, LinkSmem<GROUP, R, NCAP> &X, uint64_t (*s_mask)[NW],
                                          BitSet<NW> (&F)[R])
{
    constexpr int NMAX = NCAP, NT = NMAX * (NMAX - 1) / 2;
    static_assert(NW == (NMAX > 64 ? 2 : 1), "");
    double *S = X.S[sub];
    // entry {i, j}, i < j, sits at rowbase(i) + j
    auto rowbase = [](uint32_t i) -> int { return (int)(__umul24(i, 2u * NMAX - i - 1u) >> 1) - (int)i - 1; };
    auto at = [&](uint32_t a, uint32_t b) -> int { return a < b ? rowbase(a) + (int)b : rowbase(b) + (int)a; };
    const uint32_t nn_ = go ? n : 0u;
    CL_STAMP(0x20);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t k = sl + r * GROUP;
        if (k < nn_) {
            X.pos[sub][k] = pk[r];
            X.span[sub][k] = spk[r];
            X.inv[sub][k] = spk[r] ? 1.0 / (double)spk[r] : 0.0;
            X.size[sub][k] = 1;
            X.alist[sub][k] = (uint8_t)k;
            X.asize[sub][k] = 1;
        }
    }
    if (sl == 0) S[NT] = kNothing;
    __syncthreads();
    // every unordered pair once: row k takes the columns k+1 .. k+n/2 (mod n); for even n the distance-n/2
    // pairs only from the lower half of the rows
    {
        const uint32_t half = nn_ >> 1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t k = sl + r * GROUP;
            if (k < nn_) {
                const double ik = X.inv[sub][k];
                const uint32_t tmax = (!(nn_ & 1u) && k >= half) ? half - 1 : half;
                const uint32_t ek32 = pk[r] + spk[r], ck32 = pk[r] + (spk[r] >> 1);
                const uint64_t ek = (uint64_t)pk[r] + spk[r], ck = centre_of(pk[r], spk[r]);
                // (four columns at a time, their rows' loads side by side: a column that waited for its (pos, span) and then for the other
                // row's reciprocal was two dependent LDS trips -- 3.2 us of a unit's 16 at 1.0 M marks)
                for (uint32_t t0 = 1; t0 <= tmax; t0 += 4) {
                    uint32_t jj[4], pj4[4], spj4[4];
                    double ij4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        uint32_t j = k + min(t0 + u, tmax);
                        j = j >= nn_ ? j - nn_ : j;
                        jj[u] = j;
                        pj4[u] = X.pos[sub][j];
                        spj4[u] = X.span[sub][j];
                        ij4[u] = X.inv[sub][j];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (t0 + u > tmax) continue;
                        const uint32_t pj = pj4[u], spj = spj4[u];
                        uint32_t m;
                        if (!wide) {
                            m = min(min(absdiff_u32(pk[r], pj), absdiff_u32(ek32, pj + spj)), absdiff_u32(ck32, pj + (spj >> 1)));
                        } else {
                            const uint64_t ej = (uint64_t)pj + spj, cj = centre_of(pj, spj);
                            const uint64_t m2 = ek > ej ? ek - ej : ej - ek, m3 = ck > cj ? ck - cj : cj - ck;
                            const uint64_t mm = m2 < m3 ? m2 : m3;
                            m = absdiff_u32(pk[r], pj);
                            m = mm < (uint64_t)m ? (uint32_t)mm : m;
                        }
                        const uint32_t sdif = absdiff_u32(spk[r], spj);
                        const double inv = spk[r] > spj ? ik : ij4[u];            // 1 / the larger span (the same number when they are equal)
                        const double dp = (double)m * p.invn, ds = (double)sdif * inv;
                        S[at(k, jj[u])] = quantise(dp + ds, p.scale);
                    }
                }
            }
        }
    }
    __syncthreads();
    CL_STAMP(0x21);
    // state: this lane's rows' roots (every row) and, for the clusters they name, alive / size in LDS + the group's alive set
    uint32_t root[R];
    bool alive[R];
    int rb_me[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        root[r] = sl + r * GROUP;
        alive[r] = sl + r * GROUP < nn_;
        rb_me[r] = rowbase(sl + r * GROUP);
    }
    BitSet<NW> live = group_ballot<GROUP, R, NW>(alive, sub);
    uint32_t na = nn_;                                       // clusters left in the group
    // this round's merges, one after the other, each one on every lane's own entries: {i, dst} += {i, src} -- sums of
    // integers, any order gives the same matrix; a row dies once, so all the rounds together take fewer than n of these steps
    auto merge_round = [&](const BitSet<NW> &dead, const bool (&dies)[R], const uint32_t (&into)[R]) {
        const uint32_t nd = dead.count();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t me = sl + r * GROUP;
            if (dies[r]) {
                const uint32_t at_ = dead.count_below(me);
                X.mdst[sub][at_] = (uint8_t)into[r];
                X.msrc[sub][at_] = (uint8_t)me;
            }
            if (alive[r]) X.rep[sub][me] = (uint8_t)(dies[r] ? into[r] : me);
        }
        __syncthreads();
        // one pass of LDS adds (round 6; the tight groups' step below has the argument): every entry {x, s} with s a cluster that goes adds
        // itself to {x's cluster, s's cluster} -- the lane of x does it, of two clusters that both go the smaller one's; entries that
        // receive have two clusters that stay for indices, entries that give one that goes.  (Rounds 3-5: one merge after the other, each
        // a dependent read-add-write with a barrier behind it -- as many steps as clusters go in the round.)
        uint32_t nd_max = nd;
#pragma unroll
        for (int d = 32; d >= GROUP && d > 0; d >>= 1) nd_max = max(nd_max, (uint32_t)__shfl_xor((int)nd_max, d, 64));
        for (uint32_t t = 0; t < nd_max; t += 4) {
            const uint32_t d4 = *reinterpret_cast<const uint32_t *>(&X.mdst[sub][t]), s4 = *reinterpret_cast<const uint32_t *>(&X.msrc[sub][t]);
            double v[4][R];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t dst = (d4 >> (8 * u)) & 0xFFu, src = (s4 >> (8 * u)) & 0xFFu;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t me = sl + r * GROUP;
                    const uint32_t mine_c = dies[r] ? into[r] : me;              // my cluster after the round
                    const bool give = t + u < nd && alive[r] && me != src && mine_c != dst && !(dies[r] && me > src);
                    v[u][r] = S[give ? at(me, src) : NT + 1];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t dst = (d4 >> (8 * u)) & 0xFFu, src = (s4 >> (8 * u)) & 0xFFu;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t me = sl + r * GROUP;
                    const uint32_t mine_c = dies[r] ? into[r] : me;
                    const bool on = t + u < nd && alive[r];
                    if (on && me != src && mine_c != dst && !(dies[r] && me > src)) unsafeAtomicAdd(&S[at(mine_c, dst)], v[u][r]);
                    if (on && me == dst) X.size[sub][dst] = (uint8_t)((uint32_t)X.size[sub][dst] + (uint32_t)X.size[sub][src]);
                }
            }
        }
        __syncthreads();
        // every row follows its cluster; the clusters that are left, ascending, with their sizes
        BitSet<NW> after = live;
        for (int i = 0; i < NW; ++i) after.w[i] &= ~dead.w[i];
        const bool any_dead = dead.any();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (sl + r * GROUP < nn_) root[r] = X.rep[sub][root[r]];
            alive[r] = alive[r] && !dies[r] && any_dead;     // (a group without a merge in a round is finished: its state no longer changes)
        }
        live = after;
        na = after.count();
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t me = sl + r * GROUP;
            if (alive[r]) {
                const uint32_t at_ = after.count_below(me);
                X.alist[sub][at_] = (uint8_t)me;
                X.asize[sub][at_] = X.size[sub][me];
            }
        }
        __syncthreads();
    };
    uint32_t n_all = nn_;
#pragma unroll
    for (int d = 32; d >= GROUP && d > 0; d >>= 1) n_all = max(n_all, (uint32_t)__shfl_xor((int)n_all, d, 64));
    if (p.mergeable) {
        // Tight groups first.  A connected component of the graph {q <= threshold / 2} (or / 4) in which every pair is that
        // close is a cluster of the serial rule: while two of its clusters are left their mean is at most that level, and the
        // mean of anything of it with anything outside exceeds the level (no pair across is that close), so the rule finishes the
        // component before it touches its surroundings.  Exact integers: no guard band.  On SV-like data this takes the marks
        // of one SV in one step (identical marks included) and leaves a handful of clusters to the rounds below.
        BitSet<NW> g1[R], g2[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { g1[r].clear(); g2[r].clear(); }
        constexpr double kHalf = (double)(kQOne / 2), kQuarter = (double)(kQOne / 4);
        // (32 columns at a time, the bits of a mask word collected in one register, four entries' loads side by side)
        for (uint32_t k0 = 0; k0 < n_all; k0 += 32) {
            uint32_t a1[R], a2[R];
#pragma unroll
            for (int r = 0; r < R; ++r) a1[r] = a2[r] = 0;
            const uint32_t k1 = min(n_all, k0 + 32u);
            for (uint32_t k = k0; k < k1; k += 4) {
                double sv[4][R];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t kk = k + u;
                    const int rb_k = rowbase(kk);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t me = sl + r * GROUP;
                        int idx = me < kk ? rb_me[r] + (int)kk : rb_k + (int)me;
                        idx = (me == kk || kk >= nn_ || !alive[r]) ? NT : idx;
                        sv[u][r] = S[idx];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        a1[r] |= (sv[u][r] <= kHalf ? 1u : 0u) << (k + u - k0);
                        a2[r] |= (sv[u][r] <= kQuarter ? 1u : 0u) << (k + u - k0);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                for (int i = 0; i < NW; ++i) {
                    const bool here = NW == 1 || (k0 >> 6) == (uint32_t)i;
                    g1[r].w[i] |= here ? (uint64_t)a1[r] << (k0 & 63u) : 0ull;
                    g2[r].w[i] |= here ? (uint64_t)a2[r] << (k0 & 63u) : 0ull;
                }
            }
        }
        bool dies[R];
        uint32_t into[R];
        bool ok1[R], ok2[R];
        auto clique_rows = [&](BitSet<NW> (&g)[R], bool (&ok)[R]) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t me = sl + r * GROUP;
                g[r].set_if(alive[r], me);
                if (alive[r])
                    for (int i = 0; i < NW; ++i) s_mask[me][i] = g[r].w[i];
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                ok[r] = false;
                if (alive[r]) {
                    const uint32_t f = g[r].first();
                    BitSet<NW> o;
                    for (int i = 0; i < NW; ++i) o.w[i] = s_mask[f][i];
                    ok[r] = o.equals(g[r]);
                }
            }
            // a row's component is a clique <=> the row and all its neighbours have the neighbourhood of their smallest member
            const BitSet<NW> pass = group_ballot<GROUP, R, NW>(ok, sub);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                bool c = ok[r];
                for (int i = 0; i < NW; ++i) c = c && (g[r].w[i] & ~pass.w[i]) == 0ull;
                ok[r] = c;
            }
        };
        clique_rows(g1, ok1);
        clique_rows(g2, ok2);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t me = sl + r * GROUP;
            const uint32_t f = ok1[r] ? g1[r].first() : (ok2[r] ? g2[r].first() : me);
            dies[r] = alive[r] && f != me;
            into[r] = f;
        }
        const BitSet<NW> dead = group_ballot<GROUP, R, NW>(dies, sub);
        if (__ballot(dead.any())) {
            // the groups' sums in ONE pass of LDS adds (round 6): every unordered pair {i, j} of rows whose groups differ adds its entry to its
            // groups' entry {head(i), head(j)} -- unless it IS that entry.  An entry that receives has two heads for indices, an entry
            // that gives has a row that goes: nobody reads what somebody else writes, no barrier inside, and the adds are sums of integers
            // below 2^53 in binary64 (exact, any order).  The pairs are dealt as in the pair pass above.  (Rounds 3-5 went group after
            // group: per group a walk over its members with two barriers -- 5.9 us per unit of cl_fast_all at 1.0 M marks, the largest
            // single item of its 16.)
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t me = sl + r * GROUP;
                if (alive[r]) X.rep[sub][me] = (uint8_t)into[r];
            }
            __syncthreads();
            {
                const uint32_t half = nn_ >> 1;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t me = sl + r * GROUP;
                    if (alive[r]) {
                        const uint32_t A = into[r];
                        const uint32_t tmax = (!(nn_ & 1u) && me >= half) ? half - 1 : half;
                        for (uint32_t t0 = 1; t0 <= tmax; t0 += 4) {
                            uint32_t jj[4], B4[4];
                            double v4[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                uint32_t j = me + min(t0 + u, tmax);
                                j = j >= nn_ ? j - nn_ : j;
                                jj[u] = j;
                                B4[u] = X.rep[sub][j];
                                v4[u] = S[at(me, j)];
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (t0 + u <= tmax && A != B4[u] && !(A == me && B4[u] == jj[u])) unsafeAtomicAdd(&S[at(A, B4[u])], v4[u]);
                        }
                        if (into[r] == me) X.size[sub][me] = (uint8_t)(ok1[r] ? g1[r].count() : (ok2[r] ? g2[r].count() : 1u));
                    }
                }
            }
            __syncthreads();
            // every row follows its group; the clusters that are left, ascending, with their sizes
            BitSet<NW> after = live;
            for (int i = 0; i < NW; ++i) after.w[i] &= ~dead.w[i];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (sl + r * GROUP < nn_) root[r] = X.rep[sub][root[r]];
                alive[r] = alive[r] && !dies[r];
            }
            live = after;
            na = after.count();
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t me = sl + r * GROUP;
                if (alive[r]) {
                    const uint32_t at_ = after.count_below(me);
                    X.alist[sub][at_] = (uint8_t)me;
                    X.asize[sub][at_] = X.size[sub][me];
                }
            }
            __syncthreads();
        }
    }
    CL_STAMP(0x22);
    for (uint32_t round = 0; round < 2u * NMAX && p.mergeable; ++round) {
        CL_STAMP(0x23);
        // nearest neighbour of every cluster: smallest mean, ties to the smallest index (the list is ascending).  All lanes of
        // a group walk the same list; lane `me` looks at entry {me, k}
        double bs[R], bn[R];
        uint32_t bk[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { bs[r] = kNothing; bn[r] = 1.0; bk[r] = 0xFFu; }
        uint32_t na_max = na;
#pragma unroll
        for (int d = 32; d >= GROUP && d > 0; d >>= 1) na_max = max(na_max, (uint32_t)__shfl_xor((int)na_max, d, 64));
        for (uint32_t t = 0; t < na_max; t += 4) {
            const uint32_t ks = *reinterpret_cast<const uint32_t *>(&X.alist[sub][t < NMAX ? t : 0u]);
            const uint32_t ns = *reinterpret_cast<const uint32_t *>(&X.asize[sub][t < NMAX ? t : 0u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t k = (ks >> (8 * u)) & 0xFFu;
                const double nk = (double)((ns >> (8 * u)) & 0xFFu);
                const int rb_k = rowbase(k);
                const bool in = t + u < na;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t me = sl + r * GROUP;
                    int idx = me < k ? rb_me[r] + (int)k : rb_k + (int)me;
                    idx = (me == k || !in || !alive[r]) ? NT : idx;
                    double sv = S[idx];
                    sv = sv <= kFar ? sv : kNothing;
                    const bool better = sv * bn[r] < bs[r] * nk;
                    bs[r] = better ? sv : bs[r];
                    bn[r] = better ? nk : bn[r];
                    bk[r] = better ? k : bk[r];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t me = sl + r * GROUP;
            if (alive[r]) {
                const bool within = bs[r] <= (double)kQOne * (bn[r] * (double)X.size[sub][me]);
                bk[r] = within ? bk[r] : 0xFFu;
                X.nn[sub][me] = (uint8_t)bk[r];
            }
        }
        __syncthreads();
        // mutual nearest neighbours: the larger index goes into the smaller
        bool dies[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t me = sl + r * GROUP, k = bk[r];
            dies[r] = alive[r] && k < me && X.nn[sub][k < NMAX ? k : 0u] == me;        // (k = 0xFF: nobody)
        }
        const BitSet<NW> dead = group_ballot<GROUP, R, NW>(dies, sub);
        if (!__ballot(dead.any())) break;
        merge_round(dead, dies, bk);
    }
    // the clusters as bit sets: every row adds its bit at its root
    CL_STAMP(0x24);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t k = sl + r * GROUP;
        if (k < nn_)
            for (int i = 0; i < NW; ++i) s_mask[k][i] = 0ull;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t k = sl + r * GROUP;
        if (k < nn_) atomicOr((unsigned long long *)&s_mask[root[r]][NW == 1 ? 0 : (k >> 6)], 1ull << (k & 63u));
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t k = sl + r * GROUP;
        if (k < nn_)
            for (int i = 0; i < NW; ++i) F[r].w[i] = s_mask[root[r]][i];
    }
}

// ---------------------------------------------------------------------------------------------
// agglomeration, fast path: partitions whose clusters can be read off the threshold graph
// ---------------------------------------------------------------------------------------------
//
// What stage A0 emits depends only on the FINAL clusters of a partition (clusters by smallest member, members
// in sorted order, floor means).  Two facts about average linkage over exact means pin them down without running it:
//   * marks in different connected components of the graph {d(i,j) <= max_dist} are never merged: the mean of cross
//     distances that all exceed the threshold exceeds it;
//   * a component in which EVERY pair has d <= max_dist ends as exactly one cluster: while two of its clusters remain,
//     their mean is within the threshold, so the rule keeps merging.
// The fast pass evaluates d in binary32 (relative error < 4e-7; the fixed point's own step is 1.5e-8 of the threshold:
// hence the 1e-5 guard band around max_dist), builds each mark's closed neighbourhood as a bit mask, and accepts the
// partition when no pair falls inside the guard band and every neighbourhood equals the neighbourhood of its smallest
// member (<=> every component is a clique).  Everything else gets the exact linkage (link_unit).  Both paths produce
// the oracle's clusters; tests/test_gpu_cluster.py and tools/stress.py cover both.

// The threshold graph of one (GROUP == 64) or two (GROUP == 32) partitions of a wave with every UNORDERED pair evaluated once
// (tight_unit's sym_pass has the scheme and its cost): row i looks at row i + t (mod n) in step t, the compare mask rotated by
// t within the partition's n bits is the result of the pair that ENDS at each lane.  -> this lane's row's neighbours at the
// threshold as a column mask, and whether any of its pairs sits inside the guard band.  ps: the partition's rows in LDS
// (pos, span, end, centre); n: this lane's partition's rows (0: none).
template <int GROUP>
__device__ __forceinline__ void sym_graph0(const ClParams &p, const uint4 *ps, uint32_t pk, uint32_t spk, uint32_t ek, uint32_t ck, uint32_t n,
                                           uint32_t sl, uint64_t &cols, bool &amb)
{
    static_assert(GROUP == 64 || GROUP == 32, "one or two partitions per wave");
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n0 = (uint32_t)__builtin_amdgcn_readlane((int)n, 0), n1 = GROUP == 64 ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)n, 32);
    const uint32_t h0 = n0 >> 1, h1 = n1 >> 1, hmax = max(h0, h1);
    cols = 0;
    amb = false;
    if (hmax == 0u) return;
    const bool rowv = sl < n;
    uint32_t o = 0, r = 0;
    unsigned long long amb_all = 0;
    uint32_t j = sl;
    for (uint32_t t = 1; t <= hmax; ++t) {
        j += 1u;
        j = j >= n ? j - n : j;
        const uint4 q = ps[rowv ? j : 0u];
        const uint32_t m = min(min(absdiff_u32(pk, q.x), absdiff_u32(ek, q.z)), absdiff_u32(ck, q.w));
        const float fm = rowv ? (float)max(max(spk, q.y), 1u) : __builtin_nanf(""), fs = (float)absdiff_u32(spk, q.y);
        const float dp = (float)m * p.inv_norm;
        const float2v dp2 = {dp, dp}, fm2 = {fm, fm};
        const float2v b0 = (p.t_hl[0] - dp2) * fm2;
        unsigned long long m0 = __ballot(fs <= b0.x), ml = __ballot(fs <= b0.y), rm0, ram;
        if (GROUP == 64) {
            const unsigned long long am = m0 ^ ml;
            rm0 = (m0 << t) | (m0 >> (n0 - t));                  // (0 < t < n; bits at n and beyond belong to lanes without a row)
            ram = (am << t) | (am >> (n0 - t));
            amb_all |= am | ram;
        } else {
            const unsigned long long vm = (t <= h0 ? 0x00000000FFFFFFFFull : 0ull) | (t <= h1 ? 0xFFFFFFFF00000000ull : 0ull);
            m0 &= vm;
            const unsigned long long am = m0 ^ (ml & vm);
            auto rot = [&](unsigned long long x) -> unsigned long long {
                const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
                const uint32_t rlo = (lo << t) | (lo >> ((n0 - t) & 31u)), rhi = (hi << t) | (hi >> ((n1 - t) & 31u));
                return (unsigned long long)rlo | ((unsigned long long)rhi << 32);
            };
            rm0 = rot(m0);
            amb_all |= am | rot(am);
        }
        asm("v_addc_co_u32_e64 %0, vcc, %1, %1, %2" : "=v"(o) : "v"(o), "s"(m0) : "vcc");        // 2 x + this lane's bit
        asm("v_addc_co_u32_e64 %0, vcc, %1, %1, %2" : "=v"(r) : "v"(r), "s"(rm0) : "vcc");
    }
    // step t sits at bit hmax - t of both masks; own: column i + t, received: column i - t (mod n)
    const uint64_t X = (uint64_t)(__builtin_bitreverse32(o) >> (32u - hmax)) << 1;
    const uint64_t Y = n >= hmax ? (uint64_t)r << (n - hmax) : (uint64_t)r >> (hmax - n);
    const uint64_t Z = X | Y;
    const uint64_t A = sl == 0u ? Z : ((Z << sl) | (Z >> ((n - sl) & 63u)));
    cols = rowv ? A & (n >= 64u ? ~0ull : (1ull << n) - 1ull) : 0ull;
    amb = rowv && ((amb_all >> lane) & 1ull) != 0ull;
}

template <int GROUP, int R>
struct FastSmem {
    static constexpr int SUBS = 64 / GROUP, NMAX = GROUP * R, NW = NMAX > 64 ? 2 : 1;
    uint4 ps[SUBS][NMAX];                                    // (pos, span, end, centre): one 16-byte broadcast read per pair
    uint64_t mask[SUBS][NMAX][NW];
    // (emit_prep's sums live in the linkage's scratch, which is idle by then: 1 KB less per workgroup is the eighth
    // workgroup per CU for cl_fast_all)
};

// what a unit does: the threshold graph first and the exact linkage, in the same wavefront, for what that does not
// settle -- or the exact linkage right away (the partitions of more than 64 marks: nearly none of them is a set of cliques)
enum { kFastThenLink = 1, kLinkOnly = 2, kLinkUnfit = 3 };    // kLinkUnfit: only the partitions tight_unit does not vouch for

// one wave's worth of partitions (64 / GROUP of them, list[base ...]) of one size class
template <int GROUP, int R, int NCAP, int MODE>
__device__ __forceinline__ void fast_unit(const ClParams &p, const WorkList &list, uint32_t base, unsigned char *smem_fast,
                                          unsigned char *smem_link)
{
    constexpr int NMAX = GROUP * R, NW = NMAX > 64 ? 2 : 1;
    static_assert(NMAX <= 128 && NCAP <= NMAX && (R == 1 || (GROUP * R) % 64 == 0 || GROUP * R <= 64), "unsupported shape");
    FastSmem<GROUP, R> &S = *reinterpret_cast<FastSmem<GROUP, R> *>(smem_fast);
    const uint32_t lane = threadIdx.x, sub = lane / GROUP, sl = lane % GROUP;
    constexpr unsigned long long gm = GROUP == 64 ? ~0ull : ((1ull << (GROUP & 63)) - 1ull);
    auto group_any = [&](bool x) -> bool { return ((__ballot(x) >> (sub * GROUP)) & gm) != 0ull; };
    const uint32_t li = base + sub;
    const bool has = li < list.size();
    const uint32_t part = has ? list[li] : 0u;
    const uint32_t s = has ? p.part_start[part] : 0u;
    const uint32_t n = has ? p.part_start[part + 1] - s : 0u;
    CL_STAMP(0x10);
    __syncthreads();
    uint32_t pk[R], spk[R], ek[R], ck[R], mk[R], rd[R];
    bool bad = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t k = sl + r * GROUP;
        pk[r] = spk[r] = mk[r] = rd[r] = 0;
        if (k < n) {
            if (p.gather_rows) {
                mk[r] = mark_at(p, s + k);
                const uint3 q = load_rec(p, mk[r]);
                pk[r] = q.x;
                spk[r] = q.y;
                rd[r] = q.z;
            } else {
                const uint4 q = p.srec[s + k];           // (the rows in sorted order: cl_box laid them out, or the sort carried them)
                pk[r] = q.x;
                spk[r] = q.y;
                rd[r] = q.z;
                mk[r] = q.w & p.idx_mask;
            }
        }
        ek[r] = pk[r] + spk[r];
        ck[r] = pk[r] + (spk[r] >> 1);
        if (k < n) S.ps[sub][k] = make_uint4(pk[r], spk[r], ek[r], ck[r]);
        bad = bad || ek[r] < pk[r];                      // end does not fit 32 bits: leave it to the exact path
    }
    __syncthreads();
    CL_STAMP(0x11);
    BitSet<NW> F[R];                                     // the final cluster of each of this lane's marks
    bool solved = n < 2;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        F[r].clear();
        F[r].set_if(sl + r * GROUP < n, sl + r * GROUP);
    }
    if (MODE == kFastThenLink) {
        // closed neighbourhoods at the threshold; amb: some pair sits inside the guard band
        BitSet<NW> N0[R];
        bool amb = false;
#pragma unroll
        for (int r = 0; r < R; ++r) N0[r].clear();
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
                    amb = amb || e_hi != e_lo;
                    acc[r] |= (e_hi ? 1u : 0u) << (j - 32u * W);
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) N0[r].w[W >> 1] |= (uint64_t)acc[r] << (32u * (W & 1u));
        };
        bool by_columns = true;
        if constexpr ((GROUP == 64 || GROUP == 32) && R == 1) {
            if (p.sym) {
                // (every unordered pair once; rows whose end does not fit 32 bits are in, as in the column loop: such a
                // partition is handed to the exact linkage whatever this finds)
                sym_graph0<GROUP>(p, S.ps[sub], pk[0], spk[0], ek[0], ck[0], n, sl, N0[0].w[0], amb);
                by_columns = false;
            }
        }
        if (by_columns) {
            level0(std::integral_constant<uint32_t, 0>{});
            if constexpr (NMAX > 32) level0(std::integral_constant<uint32_t, 1>{});
            if constexpr (NMAX > 64) {
                level0(std::integral_constant<uint32_t, 2>{});
                level0(std::integral_constant<uint32_t, 3>{});
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) N0[r].set_if(sl + r * GROUP < n, sl + r * GROUP);
        // every component of the graph is a clique <=> each mark's neighbourhood equals that of its smallest member
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t k = sl + r * GROUP;
            if (k < n)
                for (int i = 0; i < NW; ++i) S.mask[sub][k][i] = N0[r].w[i];
        }
        __syncthreads();
        bool differs = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t k = sl + r * GROUP;
            if (k < n) {
                const uint32_t f = N0[r].first();
                BitSet<NW> o;
                for (int i = 0; i < NW; ++i) o.w[i] = S.mask[sub][f][i];
                differs = differs || !o.equals(N0[r]);
            }
        }
        const bool unfit = group_any(bad) || !p.fast;
        const bool amb0 = group_any(amb), cl0 = !group_any(differs);       // collective: every lane takes part
        if (n >= 2 && !unfit && !amb0 && cl0) {
            solved = true;
#pragma unroll
            for (int r = 0; r < R; ++r) F[r] = N0[r];
        }
    }
    CL_STAMP(0x12);
    const bool wide = group_any(bad);
    const bool take = MODE == kLinkUnfit ? (wide || !p.fast) : true;
    const bool need = has && !solved && take;
    if (__ballot(need)) {
        LinkSmem<GROUP, R, NCAP> &X = *reinterpret_cast<LinkSmem<GROUP, R, NCAP> *>(smem_link);
        link_unit<GROUP, R, NCAP, NW>(p, need, n, sub, sl, pk, spk, wide, X, S.mask[sub], F);
    }
    static_assert(sizeof(LinkSmem<GROUP, R, NCAP>) >= sizeof(unsigned long long) * 2 * (64 / GROUP) * NMAX, "emit_prep's sums do not fit the linkage's scratch");
    unsigned long long (*sums)[2] = reinterpret_cast<unsigned long long (*)[2]>(smem_link) + (size_t)sub * NMAX;
    CL_STAMP(0x13);
    emit_prep<GROUP, R, NW, NMAX>(p, has && take, part, s, n, sub, sl, F, S.mask[sub], S.ps[sub], sums, mk, rd);
    CL_STAMP(0x14);
}

// ---------------------------------------------------------------------------------------------
// agglomeration, contracted: what the threshold graph settles, tight groups, the linkage over what is left
// ---------------------------------------------------------------------------------------------
//
// The round-by-round linkage above keeps a partition's whole triangle of sums in LDS -- 16 KB for 64 marks, 40 KB for a
// hundred -- and that, not its instruction count, is what bounds it: seven wavefronts per CU, each a chain of dependent LDS
// round trips (doubling the LDS a workgroup holds doubles the kernels' time: profiles/history, round 3).  On SV-like data the
// triangle is not needed.  Three facts, all about exact means of the integer distances q (rule 3):
//   (a) marks in different connected components of {q <= threshold} never merge, and a component that is a clique ends as one
//       cluster (the comment above fast_unit);
//   (b) a connected component of {q <= level}, level <= threshold, that is a clique is a NODE of the merge tree (the comment in
//       link_unit): the serial rule finishes it before it touches its surroundings;
//   (c) the mean between two unions of such nodes is the sum of the member-pair distances over the product of the sizes.
// So: level 0 as before; the rows of components that are not cliques ("open" rows) are grouped by (b) at threshold / 2 and / 4;
// the sums between GROUPS -- k x k numbers, k <= 16 -- are accumulated straight from the rows (each unordered open pair is
// evaluated once, in binary64 exactly as the oracle does, and added to its groups' cell with an LDS atomic: sums of integers
// below 2^53, any order gives the same number); the rounds of mutual nearest neighbours then run on the k x k matrix.
// A workgroup holds 5 KB instead of 21.  The pair tests are binary32 with a guard band; a row with a pair inside a band
// repeats its tests in binary64 on the exact q, so the bit masks are the exact ones.  Partitions that leave more than k groups
// (no tight structure: not SV-like) or whose coordinates or parameters are outside what the binary32 tests vouch for go on the
// class's second work list and get the full triangle from the launch behind this one.
// groups a partition may have left in the first tier, by rows per unit: what keeps the second lists at a few per cent on SV-like
// data at the least LDS (the second tier takes as many groups as rows)
constexpr int tier1_groups(int nmax) { return nmax <= 8 ? 4 : (nmax <= 16 ? 8 : (nmax <= 32 ? 16 : (nmax <= 64 ? 32 : 64))); }

template <int GROUP, int R, int KC_>
struct TightSmem {
    static constexpr int SUBS = 64 / GROUP, NMAX = GROUP * R, NW = NMAX > 64 ? 2 : 1, KC = KC_, KT = KC * (KC - 1) / 2;
    static_assert(KC >= 2 && KC <= 64 && KC <= NMAX, "groups: lanes 0 .. KC-1 of a unit own them");
    uint4 ps[SUBS][NMAX];                                    // (pos, span, end, centre)
    double inv[SUBS][NMAX];                                  // 1 / span (0 for span 0), binary64 as the oracle computes it
    uint64_t mask[SUBS][NMAX][NW];
    union {
        double D[SUBS][KT + 2];                              // sums between groups, upper triangle row by row; [KT] "nothing there", [KT + 1] scratch
        unsigned long long sum[SUBS][NMAX][2];               // (emit_prep, when the linkage is over)
    };
    uint8_t gof[SUBS][NMAX];                                 // an open row's group (0xFF: the row is not open)
    uint8_t ghead[SUBS][KC], gsize[SUBS][KC], nn[SUBS][KC], rep[SUBS][KC];
    uint8_t alist[SUBS][KC + 4], asize[SUBS][KC + 4];        // the clusters that are left, ascending, and their sizes (read four at a time)
};

// one wave's worth of partitions (64 / GROUP of them, list[base ...]) of one size class.  First tier (TIER2 false): at most
// KC_ groups, the partitions beyond that -- and those it does not vouch for -- go on the class's second list (over_items /
// over_counts); second tier: the second list, as many groups as rows (the partitions it does not vouch for have been dealt
// with by the full-triangle linkage in front of it: tier2_unit)
template <int GROUP, int R, int KC_, bool TIER2>
__device__ __forceinline__ bool tight_unit(const ClParams &p, const WorkList &list, uint32_t base, unsigned char *smem,
                                           uint32_t *over_items, uint32_t *over_counts)
{
    using SM = TightSmem<GROUP, R, KC_>;
    constexpr int NMAX = SM::NMAX, NW = SM::NW, KC = SM::KC, KT = SM::KT;
    constexpr bool kOnePass = GROUP == 64;               // classes whose partitions are mostly open: all three levels in one pass over the pairs
    SM &S = *reinterpret_cast<SM *>(smem);
    const uint32_t lane = threadIdx.x, sub = lane / GROUP, sl = lane % GROUP;
    constexpr unsigned long long gm = GROUP == 64 ? ~0ull : ((1ull << (GROUP & 63)) - 1ull);
    auto group_any = [&](bool x) -> bool { return ((__ballot(x) >> (sub * GROUP)) & gm) != 0ull; };
    auto wave_max = [&](uint32_t x) -> uint32_t {          // over the wave's groups (x is uniform within a group): a scalar
#pragma unroll
        for (int d = 32; d >= GROUP && d > 0; d >>= 1) x = max(x, (uint32_t)__shfl_xor((int)x, d, 64));
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
    };
    const uint32_t li = base + sub;
    const bool has = li < list.size();
    const uint32_t part = has ? list[li] : 0u;
    const uint32_t s = has ? p.part_start[part] : 0u;
    const uint32_t n = has ? p.part_start[part + 1] - s : 0u;
    CL_STAMP(0x30);
    __syncthreads();
    uint32_t pk[R], spk[R], ek[R], ck[R], mk[R], rd[R];
    double ik[R];
    bool bad = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t k = sl + r * GROUP;
        pk[r] = spk[r] = mk[r] = rd[r] = 0;
        if (k < n) {
            if (p.gather_rows) {
                mk[r] = mark_at(p, s + k);
                const uint3 q = load_rec(p, mk[r]);
                pk[r] = q.x;
                spk[r] = q.y;
                rd[r] = q.z;
            } else {
                const uint4 q = p.srec[s + k];           // (the rows in sorted order: cl_box laid them out, or the sort carried them)
                pk[r] = q.x;
                spk[r] = q.y;
                rd[r] = q.z;
                mk[r] = q.w & p.idx_mask;
            }
        }
        ek[r] = pk[r] + spk[r];
        ck[r] = pk[r] + (spk[r] >> 1);
        ik[r] = spk[r] ? 1.0 / (double)spk[r] : 0.0;
        if (k < n) {
            S.ps[sub][k] = make_uint4(pk[r], spk[r], ek[r], ck[r]);
            S.inv[sub][k] = ik[r];
        }
        bad = bad || ek[r] < pk[r];                      // end does not fit 32 bits
    }
    __syncthreads();
    CL_STAMP(0x31);
    const bool unfit = group_any(bad) || !p.fast;
    const bool act = has && n >= 2 && !unfit;            // (uniform within a group)
    const uint32_t na = act ? n : 0u;
    const uint32_t n_all = wave_max(na);
    // q of rule 3 for (this lane's row r, row j), exactly the oracle's operations (ends fit 32 bits here)
    auto q_exact = [&](int r, uint32_t j) -> double {
        const uint4 q = S.ps[sub][j];
        const uint32_t m = min(min(absdiff_u32(pk[r], q.x), absdiff_u32(ek[r], q.z)), absdiff_u32(ck[r], q.w));
        const double inv = spk[r] > q.y ? ik[r] : S.inv[sub][j];         // 1 / the larger span (the same number when they are equal)
        const double dp = (double)m * p.invn, ds = (double)absdiff_u32(spk[r], q.y) * inv;
        return quantise(dp + ds, p.scale);
    };
    BitSet<NW> F[R];                                     // the final cluster of each of this lane's marks
    BitSet<NW> N0[R], g1[R], g2[R];                      // closed neighbourhoods at the threshold, at / 2, at / 4
    bool row[R], ambr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        F[r].clear();
        F[r].set_if(sl + r * GROUP < n, sl + r * GROUP);
        N0[r].clear();
        g1[r].clear();
        g2[r].clear();
        row[r] = sl + r * GROUP < na;
        ambr[r] = false;
    }
    // The pair tests in binary32 (relative error < 4e-7; the fixed point's step is 1.5e-8 of the threshold: hence the 1e-5 guard
    // band around each level), 32 columns at a time: the bits of one mask word are collected in one register.  L0 / L12: which
    // levels; cols: the columns that count for L12 (the open rows)
    auto pair_pass = [&](auto wc, auto l0, auto l12, const BitSet<NW> &cols) {
        constexpr uint32_t W = decltype(wc)::value;
        constexpr bool L0 = decltype(l0)::value, L12 = decltype(l12)::value;
        const uint32_t j1 = min(n_all, 32u * (W + 1u));
        uint32_t a0[R], a1[R], a2[R], am[R];
#pragma unroll
        for (int r = 0; r < R; ++r) a0[r] = a1[r] = a2[r] = am[r] = 0;
        const uint32_t cw = (uint32_t)(cols.w[W >> 1] >> (32u * (W & 1u)));
        // (two columns per step, both loads in front; a column beyond a group's own count reads something stale inside the
        // array and is masked)
        auto column = [&](uint32_t j, const uint4 &q) {
            const bool in = j < na;
            const bool in12 = L0 ? in : (in && ((cw >> (j - 32u * W)) & 1u));
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t m = min(min(absdiff_u32(pk[r], q.x), absdiff_u32(ek[r], q.z)), absdiff_u32(ck[r], q.w));
                const float fm = (float)max(max(spk[r], q.y), 1u), fs = (float)absdiff_u32(spk[r], q.y);
                const float dp = (float)m * p.inv_norm;
                bool amb = false;
                const float2v dp2 = {dp, dp}, fm2 = {fm, fm};        // (both bounds of a level in one packed subtract and multiply)
                if constexpr (L0) {
                    const float2v b = (p.t_hl[0] - dp2) * fm2;
                    const bool h = fs <= b.x, l = fs <= b.y;
                    amb = in && h != l;
                    a0[r] |= ((in && h) ? 1u : 0u) << (j - 32u * W);
                }
                if constexpr (L12) {
                    const float2v b1 = (p.t_hl[1] - dp2) * fm2, b2 = (p.t_hl[2] - dp2) * fm2;
                    const bool h1 = fs <= b1.x, l1 = fs <= b1.y;
                    const bool h2 = fs <= b2.x, l2 = fs <= b2.y;
                    amb = amb || (in12 && (h1 != l1 || h2 != l2));
                    a1[r] |= ((in12 && h1) ? 1u : 0u) << (j - 32u * W);
                    a2[r] |= ((in12 && h2) ? 1u : 0u) << (j - 32u * W);
                }
                am[r] |= amb ? 1u : 0u;
            }
        };
        for (uint32_t j = 32u * W; j < j1; j += 2) {
            const uint4 q0 = S.ps[sub][j], q1 = S.ps[sub][j + 1u];
            column(j, q0);
            column(j + 1u, q1);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            ambr[r] = ambr[r] || am[r] != 0u;
            if constexpr (L0) N0[r].w[W >> 1] |= (uint64_t)a0[r] << (32u * (W & 1u));
            if constexpr (L12) {
                g1[r].w[W >> 1] |= (uint64_t)a1[r] << (32u * (W & 1u));
                g2[r].w[W >> 1] |= (uint64_t)a2[r] << (32u * (W & 1u));
            }
        }
    };
    auto all_words = [&](auto l0, auto l12, const BitSet<NW> &cols) {
        pair_pass(std::integral_constant<uint32_t, 0>{}, l0, l12, cols);
        if constexpr (NMAX > 32) pair_pass(std::integral_constant<uint32_t, 1>{}, l0, l12, cols);
        if constexpr (NMAX > 64) {
            pair_pass(std::integral_constant<uint32_t, 2>{}, l0, l12, cols);
            pair_pass(std::integral_constant<uint32_t, 3>{}, l0, l12, cols);
        }
    };
    // a row with a pair inside a guard band repeats its levels on the exact q (so does the pair's other row: the binary32
    // expressions are symmetric)
    auto exact_rows = [&](auto l0, auto l12, const bool (&on)[R], const BitSet<NW> &cols) {
        constexpr bool L0 = decltype(l0)::value, L12 = decltype(l12)::value;
        bool any = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            ambr[r] = ambr[r] && on[r];
            any = any || ambr[r];
        }
        if (__ballot(any)) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (ambr[r]) {
                    if constexpr (L0) N0[r].clear();
                    if constexpr (L12) { g1[r].clear(); g2[r].clear(); }
                }
            for (uint32_t j = 0; j < n_all; ++j) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (ambr[r] && j < na) {
                        const double qq = q_exact(r, j);
                        if constexpr (L0) N0[r].set_if(qq <= (double)kQOne, j);
                        if constexpr (L12) {
                            const bool c = L0 || cols.test(j);
                            g1[r].set_if(c && qq <= (double)(kQOne / 2), j);
                            g2[r].set_if(c && qq <= (double)(kQOne / 4), j);
                        }
                    }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) ambr[r] = false;
    };
    // One partition per wave, one row per lane: every UNORDERED pair once.  In step t row i looks at row i + t (mod n), and what
    // it finds is also row i + t's result for row i - t': the compare results are lane masks, and rotating such a mask by t
    // within the partition's n bits hands every lane the result of the pair that ends at it -- on the scalar unit, nothing
    // crosses lanes.  Both results enter the lanes' masks in "rotated" order (one add-with-carry each: mask = 2 mask + bit) and
    // are turned to column order once at the end.  n / 2 steps of ~33 vector instructions where the column loop below takes
    // n steps of 34 (the binary32 expressions are symmetric in the two rows, so both rows get the same answer, guard band
    // included).
    auto sym_pass = [&]() {
        static_assert(GROUP != 64 || R != 1 || NW == 1, "one mask word");
        const uint32_t n_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)na);
        if (n_s < 2u) return;
        const uint32_t half = n_s >> 1;
        const bool actv = lane < n_s;
        const unsigned long long nmask = n_s >= 64u ? ~0ull : (1ull << n_s) - 1ull;
        uint32_t o0 = 0, o1 = 0, o2 = 0, r0 = 0, r1 = 0, r2 = 0;
        unsigned long long amb_all = 0;
        auto push = [](uint32_t x, unsigned long long bit_of_lane) -> uint32_t {     // 2 x + (this lane's bit of the mask)
            uint32_t d;
            asm("v_addc_co_u32_e64 %0, vcc, %1, %1, %2" : "=v"(d) : "v"(x), "s"(bit_of_lane) : "vcc");
            return d;
        };
        uint32_t j = lane;
        for (uint32_t t = 1; t <= half; ++t) {
            j += 1u;
            j = j >= n_s ? j - n_s : j;
            const uint4 q = S.ps[sub][actv ? j : 0u];
            const uint32_t m = min(min(absdiff_u32(pk[0], q.x), absdiff_u32(ek[0], q.z)), absdiff_u32(ck[0], q.w));
            // (a lane without a row compares against NaN bounds: false on every level, so the masks hold bits of rows only and
            // a rotation brings nothing in from beyond them)
            const float fm = actv ? (float)max(max(spk[0], q.y), 1u) : __builtin_nanf(""), fs = (float)absdiff_u32(spk[0], q.y);
            const float dp = (float)m * p.inv_norm;
            const float2v dp2 = {dp, dp}, fm2 = {fm, fm};
            const float2v b0 = (p.t_hl[0] - dp2) * fm2, b1 = (p.t_hl[1] - dp2) * fm2, b2 = (p.t_hl[2] - dp2) * fm2;
            const unsigned long long m0 = __ballot(fs <= b0.x), m1 = __ballot(fs <= b1.x), m2 = __ballot(fs <= b2.x);
            // inside a guard band: the two bounds of a level disagree (scalar arithmetic on the lane masks)
            const unsigned long long am = (m0 ^ __ballot(fs <= b0.y)) | (m1 ^ __ballot(fs <= b1.y)) | (m2 ^ __ballot(fs <= b2.y));
            // (bits at n and beyond are not looked at: the lanes there have no row)
            auto rot = [&](unsigned long long x) -> unsigned long long { return (x << t) | (x >> (n_s - t)); };              // (0 < t < n)
            amb_all |= am | rot(am);
            o0 = push(o0, m0); o1 = push(o1, m1); o2 = push(o2, m2);
            r0 = push(r0, rot(m0)); r1 = push(r1, rot(m1)); r2 = push(r2, rot(m2));
        }
        // step t sits at bit half - t of both masks; own: column i + t, received: column i - t (mod n)
        auto columns = [&](uint32_t o, uint32_t r) -> uint64_t {
            const uint64_t X = (uint64_t)(__builtin_bitreverse32(o) >> (32u - half)) << 1;      // bit t: step t
            const uint64_t Y = (uint64_t)r << (n_s - half);                                       // bit n - t: step t
            const uint64_t Z = X | Y;
            const uint64_t A = lane == 0u ? Z : ((Z << lane) | (Z >> ((n_s - lane) & 63u)));
            return actv ? A & nmask : 0ull;
        };
        N0[0].w[0] = columns(o0, r0);
        g1[0].w[0] = columns(o1, r1);
        g2[0].w[0] = columns(o2, r2);
        ambr[0] = actv && ((amb_all >> lane) & 1ull) != 0ull;
    };
    // The same for two partitions per wave (GROUP == 32: lanes 0..31 and 32..63, n0 and n1 rows): the compare masks are rotated
    // half by half, each by t within its own partition's bits; the wave steps to the larger n / 2 and a partition that is done
    // contributes zeros.  Levels as pair_pass: L0 the threshold, L12 threshold / 2 and / 4 (the columns that do not count for
    // L12 -- rows that are not open -- are masked out by the caller afterwards, as are their rows).
    auto sym_pass32 = [&](auto l0, auto l12) {
        constexpr bool L0 = decltype(l0)::value, L12 = decltype(l12)::value;
        static_assert(L0 != L12, "one pass per kind");
        const uint32_t n0 = (uint32_t)__builtin_amdgcn_readlane((int)na, 0), n1 = (uint32_t)__builtin_amdgcn_readlane((int)na, 32);
        const uint32_t h0 = n0 >> 1, h1 = n1 >> 1, hmax = max(h0, h1);
        if (hmax == 0u) return;
        const bool rowv = sl < na;
        uint32_t oA = 0, oB = 0, rA = 0, rB = 0;
        unsigned long long amb_all = 0;
        auto push = [](uint32_t x, unsigned long long bit_of_lane) -> uint32_t {     // 2 x + (this lane's bit of the mask)
            uint32_t d;
            asm("v_addc_co_u32_e64 %0, vcc, %1, %1, %2" : "=v"(d) : "v"(x), "s"(bit_of_lane) : "vcc");
            return d;
        };
        uint32_t j = sl;
        for (uint32_t t = 1; t <= hmax; ++t) {
            j += 1u;
            j = j >= na ? j - na : j;
            const uint4 q = S.ps[sub][rowv ? j : 0u];
            const uint32_t m = min(min(absdiff_u32(pk[0], q.x), absdiff_u32(ek[0], q.z)), absdiff_u32(ck[0], q.w));
            const float fm = rowv ? (float)max(max(spk[0], q.y), 1u) : __builtin_nanf(""), fs = (float)absdiff_u32(spk[0], q.y);
            const float dp = (float)m * p.inv_norm;
            const float2v dp2 = {dp, dp}, fm2 = {fm, fm};
            // (a partition that has done its n / 2 steps is out: its half of every mask is cleared)
            const unsigned long long vm = (t <= h0 ? 0x00000000FFFFFFFFull : 0ull) | (t <= h1 ? 0xFFFFFFFF00000000ull : 0ull);
            auto rot = [&](unsigned long long x) -> unsigned long long {
                const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
                const uint32_t rlo = (lo << t) | (lo >> ((n0 - t) & 31u)), rhi = (hi << t) | (hi >> ((n1 - t) & 31u));
                return (unsigned long long)rlo | ((unsigned long long)rhi << 32);
            };
            if constexpr (L0) {
                const float2v b0 = (p.t_hl[0] - dp2) * fm2;
                const unsigned long long m0 = __ballot(fs <= b0.x) & vm;
                const unsigned long long am = (m0 ^ (__ballot(fs <= b0.y) & vm));
                amb_all |= am | rot(am);
                oA = push(oA, m0);
                rA = push(rA, rot(m0));
            } else {
                const float2v b1 = (p.t_hl[1] - dp2) * fm2, b2 = (p.t_hl[2] - dp2) * fm2;
                const unsigned long long m1 = __ballot(fs <= b1.x) & vm, m2 = __ballot(fs <= b2.x) & vm;
                const unsigned long long am = (m1 ^ (__ballot(fs <= b1.y) & vm)) | (m2 ^ (__ballot(fs <= b2.y) & vm));
                amb_all |= am | rot(am);
                oA = push(oA, m1); oB = push(oB, m2);
                rA = push(rA, rot(m1)); rB = push(rB, rot(m2));
            }
        }
        // step t sits at bit hmax - t of both masks; own: column i + t, received: column i - t (mod n)
        auto columns = [&](uint32_t o, uint32_t r) -> uint64_t {
            const uint64_t X = (uint64_t)(__builtin_bitreverse32(o) >> (32u - hmax)) << 1;      // bit t: step t
            const uint64_t Y = na >= hmax ? (uint64_t)r << (na - hmax) : (uint64_t)r >> (hmax - na);     // bit n - t: step t
            const uint64_t Z = X | Y;
            const uint64_t A = sl == 0u ? Z : ((Z << sl) | (Z >> ((na - sl) & 63u)));
            return rowv ? A & ((1ull << na) - 1ull) : 0ull;
        };
        if constexpr (L0) {
            N0[0].w[0] = columns(oA, rA);
        } else {
            g1[0].w[0] = columns(oA, rA);
            g2[0].w[0] = columns(oB, rB);
        }
        ambr[0] = ambr[0] || (rowv && ((amb_all >> lane) & 1ull) != 0ull);
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    BitSet<NW> nocols;
    nocols.clear();
    constexpr bool kSym = GROUP == 64 && R == 1;
    constexpr bool kSym32 = GROUP == 32 && R == 1;
    if constexpr (kSym) {
        if (p.sym) {
            sym_pass();
        } else {
            all_words(T_{}, T_{}, nocols);
        }
        exact_rows(T_{}, T_{}, row, nocols);
    } else if constexpr (kOnePass) {
        all_words(T_{}, T_{}, nocols);
        exact_rows(T_{}, T_{}, row, nocols);
    } else {
        if constexpr (kSym32) {
            if (p.sym) sym_pass32(T_{}, F_{});
            else all_words(T_{}, F_{}, nocols);
        } else {
            all_words(T_{}, F_{}, nocols);
        }
        exact_rows(T_{}, F_{}, row, nocols);
    }
    CL_STAMP(0x32);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        N0[r].set_if(row[r], sl + r * GROUP);
        if (!row[r]) N0[r].clear();
    }
    // a row's component of the graph given by the masks g is a clique <=> the row and all its neighbours have the neighbourhood
    // of their smallest member (rows with on == false stay out of it)
    auto clique_rows = [&](BitSet<NW> (&g)[R], const bool (&on)[R], bool (&ok)[R]) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t me = sl + r * GROUP;
            if (on[r])
                for (int i = 0; i < NW; ++i) S.mask[sub][me][i] = g[r].w[i];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            ok[r] = false;
            if (on[r]) {
                const uint32_t f = g[r].first();
                BitSet<NW> o;
                for (int i = 0; i < NW; ++i) o.w[i] = S.mask[sub][f][i];
                ok[r] = o.equals(g[r]);
            }
        }
        const BitSet<NW> pass = group_ballot<GROUP, R, NW>(ok, sub);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            bool c = ok[r];
            for (int i = 0; i < NW; ++i) c = c && (g[r].w[i] & ~pass.w[i]) == 0ull;
            ok[r] = c;
        }
    };
    bool ok0[R], open[R];
    clique_rows(N0, row, ok0);
    CL_STAMP(0x33);
    bool any_open = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (row[r] && ok0[r]) F[r] = N0[r];              // (a): the component is a clique, hence a cluster
        open[r] = row[r] && !ok0[r];
        any_open = any_open || open[r];
    }
    bool too_many = false;
    if (__ballot(any_open)) {
        const BitSet<NW> omask = group_ballot<GROUP, R, NW>(open, sub);
        // the open rows' neighbourhoods among themselves at threshold / 2 and / 4 (the rows within threshold / 2 of an open
        // row are open themselves: they are in its component)
        if constexpr (!kOnePass) {
            if constexpr (kSym32) {
                if (p.sym) sym_pass32(F_{}, T_{});
                else all_words(F_{}, T_{}, omask);
            } else {
                all_words(F_{}, T_{}, omask);
            }
            exact_rows(F_{}, T_{}, open, omask);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            for (int i = 0; i < NW; ++i) {
                g1[r].w[i] &= omask.w[i];
                g2[r].w[i] &= omask.w[i];
            }
            g1[r].set_if(open[r], sl + r * GROUP);
            g2[r].set_if(open[r], sl + r * GROUP);
            if (!open[r]) { g1[r].clear(); g2[r].clear(); }
        }
        CL_STAMP(0x34);
        bool ok1[R], ok2[R], head[R];
        clique_rows(g1, open, ok1);
        clique_rows(g2, open, ok2);
        CL_STAMP(0x35);
        // (b): the groups, named by their smallest member
        uint32_t into[R], gsz[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t me = sl + r * GROUP;
            into[r] = ok1[r] ? g1[r].first() : (ok2[r] ? g2[r].first() : me);
            gsz[r] = ok1[r] ? g1[r].count() : (ok2[r] ? g2[r].count() : 1u);
            head[r] = open[r] && into[r] == me;
        }
        const BitSet<NW> heads = group_ballot<GROUP, R, NW>(head, sub);
        const uint32_t k = heads.count();
        too_many = k > (TIER2 ? (uint32_t)KC : min(p.kc, (uint32_t)KC));
        const uint32_t kk = too_many ? 0u : k;               // (a group that hands its partition on sits the rest out)
        uint32_t gi[R];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            gi[r] = open[r] ? heads.count_below(into[r]) : 0xFFu;
            if (row[r]) S.gof[sub][sl + r * GROUP] = (uint8_t)gi[r];         // (0xFF: not an open row)
            if (head[r] && kk) {
                S.ghead[sub][gi[r]] = (uint8_t)(sl + r * GROUP);
                S.gsize[sub][gi[r]] = (uint8_t)gsz[r];
                S.alist[sub][gi[r]] = (uint8_t)gi[r];
                S.asize[sub][gi[r]] = (uint8_t)gsz[r];
            }
        }
        double *D = S.D[sub];
        // entry {a, b}, a < b, sits at rowbase(a) + b
        auto rowbase = [](uint32_t i) -> int { return (int)(__umul24(i, 2u * KC - i - 1u) >> 1) - (int)i - 1; };
        for (uint32_t i = sl; i < (uint32_t)KT + 2u; i += GROUP) D[i] = i == (uint32_t)KT ? kNothing : 0.0;
        __syncthreads();
        CL_STAMP(0x36);
        // (c): every unordered pair of rows once -- row i takes i+1 .. i+n/2 (mod n), for even n the distance-n/2 pairs only from
        // the lower half -- and the pairs of open rows of different groups go into their groups' cell
        {
            const uint32_t no = kk ? na : 0u, half = no >> 1, half_all = wave_max(half);
            uint32_t tmax[R];
#pragma unroll
            for (int r = 0; r < R; ++r) tmax[r] = (!open[r] || !no) ? 0u : ((!(no & 1u) && sl + r * GROUP >= half) ? half - 1u : half);
            for (uint32_t t = 1; t <= half_all; ++t) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    uint32_t j = sl + r * GROUP + t;
                    j = j >= no ? j - no : j;
                    j = t <= tmax[r] ? j : 0u;
                    const uint32_t gj = S.gof[sub][j];
                    if (t <= tmax[r] && gj != 0xFFu && gj != gi[r]) {
                        const double qq = q_exact(r, j);
                        const uint32_t lo = min(gi[r], gj), hi = max(gi[r], gj);
                        __hip_atomic_fetch_add(&D[rowbase(lo) + (int)hi], qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        __syncthreads();
        CL_STAMP(0x37);
        // the rounds of link_unit on the groups: lane a < k owns group a (a cluster keeps the index of its smallest group,
        // which is the rank of its smallest member: the oracle's tie order)
        const uint32_t a = sl;
        const bool mine = a < kk;
        bool aliveA = mine;
        uint32_t rootA = a;
        const int rb_a = rowbase(a < (uint32_t)KC ? a : 0u);
        constexpr unsigned long long km = KC == 64 ? ~0ull : ((1ull << (KC & 63)) - 1ull);
        uint32_t nl = kk;                                    // clusters left
        for (uint32_t round = 0; round < (uint32_t)KC; ++round) {
            CL_STAMP(0x38);
            // nearest neighbour of every cluster: smallest mean, ties to the smallest index -- the list of what is left is
            // ascending, read four entries at a time
            const double sizeA = aliveA ? (double)S.gsize[sub][a] : 1.0;
            double bs = kNothing, bn = 1.0;
            uint32_t bk = 0xFFu;
            const uint32_t nl_all = wave_max(nl);
            for (uint32_t t = 0; t < nl_all; t += 4) {
                const uint32_t ks = *reinterpret_cast<const uint32_t *>(&S.alist[sub][t < (uint32_t)KC ? t : 0u]);
                const uint32_t ns = *reinterpret_cast<const uint32_t *>(&S.asize[sub][t < (uint32_t)KC ? t : 0u]);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint32_t b = (ks >> (8 * w)) & 0xFFu;
                    const double nk = (double)((ns >> (8 * w)) & 0xFFu);
                    const bool in = aliveA && t + w < nl && b != a;
                    const int idx = !in ? KT : (a < b ? rb_a + (int)b : rowbase(b) + (int)a);
                    double sv = D[idx];
                    sv = sv <= kFar ? sv : kNothing;
                    const bool better = sv * bn < bs * nk;
                    bs = better ? sv : bs;
                    bn = better ? nk : bn;
                    bk = better ? b : bk;
                }
            }
            const bool within = bs <= (double)kQOne * (bn * sizeA);
            bk = (aliveA && within) ? bk : 0xFFu;
            if (mine) S.nn[sub][a] = (uint8_t)bk;
            __syncthreads();
            // mutual nearest neighbours: the larger index goes into the smaller
            const bool dies = aliveA && bk < a && S.nn[sub][bk < (uint32_t)KC ? bk : 0u] == a;
            unsigned long long deadK = (__ballot(dies) >> (sub * GROUP)) & km;
            if (!__ballot(dies)) break;
            // this round's merges one after the other, each on every lane's own entries: {x, dst} += {x, src} (link_unit's merge_round)
            const uint32_t nd_all = wave_max((uint32_t)__popcll(deadK));
            for (uint32_t t = 0; t < nd_all; ++t) {
                const bool on = deadK != 0ull;
                const uint32_t src = on ? (uint32_t)__ffsll((long long)deadK) - 1u : 0u;
                deadK &= deadK - 1ull;
                const uint32_t dst = on ? S.nn[sub][src] : 0u;
                const bool upd = on && aliveA && a != dst && a != src;
                const int i_d = a < dst ? rb_a + (int)dst : rowbase(dst) + (int)a;
                const int i_s = a < src ? rb_a + (int)src : rowbase(src) + (int)a;
                const double v = D[upd ? i_d : KT + 1] + D[upd ? i_s : KT + 1];
                D[upd ? i_d : KT + 1] = v;
                __syncthreads();
            }
            if (dies) S.gsize[sub][bk] = (uint8_t)((uint32_t)S.gsize[sub][bk] + (uint32_t)S.gsize[sub][a]);
            if (mine) S.rep[sub][a] = (uint8_t)(dies ? bk : a);
            aliveA = aliveA && !dies;
            const unsigned long long liveK = (__ballot(aliveA) >> (sub * GROUP)) & km;
            nl = (uint32_t)__popcll(liveK);
            __syncthreads();
            if (mine) rootA = S.rep[sub][rootA];
            if (aliveA) {
                const uint32_t at_ = (uint32_t)__popcll(liveK & ((1ull << a) - 1ull));
                S.alist[sub][at_] = (uint8_t)a;
                S.asize[sub][at_] = S.gsize[sub][a];
            }
            __syncthreads();
        }
        // the clusters as bit sets: every open row adds its bit at the smallest member of its group's cluster
        CL_STAMP(0x39);
        __syncthreads();
        if (mine) S.nn[sub][a] = (uint8_t)rootA;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t me = sl + r * GROUP;
            if (open[r])
                for (int i = 0; i < NW; ++i) S.mask[sub][me][i] = 0ull;
        }
        __syncthreads();
        uint32_t hr[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t me = sl + r * GROUP;
            hr[r] = me;
            if (open[r] && kk) {
                hr[r] = S.ghead[sub][S.nn[sub][gi[r]]];
                atomicOr((unsigned long long *)&S.mask[sub][hr[r]][NW == 1 ? 0 : (me >> 6)], 1ull << (me & 63u));
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (open[r] && kk)
                for (int i = 0; i < NW; ++i) F[r].w[i] = S.mask[sub][hr[r]][i];
    }
    // hand on what this unit does not vouch for
    const bool over = has && n >= 2 && (unfit || too_many);
    if (!TIER2 && over && sl == 0) over_append(p, over_items, over_counts, part, s);
    CL_STAMP(0x3A);
    emit_prep<GROUP, R, NW, NMAX>(p, has && !over, part, s, n, sub, sl, F, S.mask[sub], S.ps[sub], S.sum[sub], mk, rd);
    CL_STAMP(0x3B);
    return __ballot(has && n >= 2 && unfit) != 0ull;     // (some partition of this wave was not vouched for)
}
// one size class per launch: what large inputs use (the fused kernel needs the registers of all variants at once; with
// millions of partitions per class there is nothing to gain from fusing)
template <int GROUP, int R, int KC>
__global__ __launch_bounds__(64) void cl_tight_one(const ClParams p, uint32_t *items /* the class's lists */, const uint32_t *counts /* [kShards] */,
                                                   uint32_t *over_counts /* [kShards] */)
{
    CL_STAMP_INIT(GROUP == 64 ? 4 : (GROUP == 32 ? 5 : 0));       // (large inputs: cl_fast_all's and cl_link_one's areas are free / shared)
    __shared__ __align__(16) unsigned char smem[sizeof(TightSmem<GROUP, R, KC>)];
    __shared__ uint32_t s_pref[kShards + 1];
    worklist_prefix(counts, s_pref);
    __syncthreads();
    const WorkList list{items, s_pref, p.tps * kScanTile, 0u};
    const uint32_t L = list.size();
    for (uint32_t base = blockIdx.x * (64 / GROUP); base < L; base += gridDim.x * (64 / GROUP))
        tight_unit<GROUP, R, KC, false>(p, list, base, smem, items, over_counts);
}

// The partitions of more than 64 marks without a list: a wavefront looks at 64 partitions' sizes at a time and takes the ones
// it finds (a handful in a million marks; the launch that listed them first cost their chain 10 us).
template <int KC>
__global__ __launch_bounds__(64) void cl_tight_big(const ClParams p, uint32_t *over_items, uint32_t *over_counts /* [kShards] */,
                                                   uint32_t *found /* how many there were (statistics) */)
{
    CL_STAMP_INIT(3);
    __shared__ __align__(16) unsigned char smem[sizeof(TightSmem<64, 2, KC>)];
    const uint32_t n_parts = *p.n_parts, lane = threadIdx.x;
    const WorkList all{nullptr, nullptr, n_parts, 0u};
    for (uint32_t b0 = blockIdx.x * 64u; b0 < n_parts; b0 += gridDim.x * 64u) {
        const uint32_t part = b0 + lane;
        const bool big = part < n_parts && p.part_start[part + 1] - p.part_start[part] > 64u;
        unsigned long long m = __ballot(big);
        while (m) {
            const uint32_t at = (uint32_t)__ffsll((long long)m) - 1u;
            m &= m - 1ull;
            tight_unit<64, 2, KC, false>(p, all, b0 + at, smem, over_items, over_counts);
            if (lane == 0) atomicAdd(found, 1u);
        }
    }
}

constexpr size_t cmax(size_t a, size_t b) { return a > b ? a : b; }
constexpr int kK8 = tier1_groups(8), kK16 = tier1_groups(16), kK32 = tier1_groups(32), kK64 = tier1_groups(64);

// the classes of up to 64 marks in one launch, largest partitions first (they are the longest chains): a virtual block is
// one wave's worth of partitions of one class
constexpr size_t kTightSmemBytes = cmax(cmax(sizeof(TightSmem<64, 1, kK64>), sizeof(TightSmem<32, 1, kK32>)),
                                        cmax(sizeof(TightSmem<16, 1, kK16>), sizeof(TightSmem<8, 1, kK8>)));

__global__ __launch_bounds__(64, 6) void cl_tight_all(const ClParams p, uint32_t *lists, const uint32_t *cnts /* [kClasses][kShards] */,
                                                   uint32_t *over_cnts /* [kClasses][kShards] */)
{
    CL_STAMP_INIT(0);
    __shared__ __align__(16) unsigned char smem[kTightSmemBytes];
    __shared__ uint32_t s_pref[4][kShards + 1];
#pragma unroll
    for (int c = 0; c < 4; ++c) worklist_prefix(cnts + c * kShards, s_pref[c]);
    __syncthreads();
    const size_t M = p.M;
    const uint32_t span = p.tps * kScanTile;
    const WorkList l0{lists, s_pref[0], span, 0u}, l1{lists + M, s_pref[1], span, 0u}, l2{lists + 2 * M, s_pref[2], span, 0u},
        l3{lists + 3 * M, s_pref[3], span, 0u};
    const uint32_t c0 = l0.size(), c1 = l1.size(), c2 = l2.size(), c3 = l3.size();
    const uint32_t b3 = c3, b2 = b3 + (c2 + 1) / 2, b1 = b2 + (c1 + 3) / 4, b0 = b1 + (c0 + 7) / 8;
    for (uint32_t vb = blockIdx.x; vb < b0; vb += gridDim.x) {
        if (vb < b3) tight_unit<64, 1, kK64, false>(p, l3, vb, smem, lists + 3 * M, over_cnts + 3 * kShards);
        else if (vb < b2) tight_unit<32, 1, kK32, false>(p, l2, (vb - b3) * 2, smem, lists + 2 * M, over_cnts + 2 * kShards);
        else if (vb < b1) tight_unit<16, 1, kK16, false>(p, l1, (vb - b2) * 4, smem, lists + M, over_cnts + kShards);
        else tight_unit<8, 1, kK8, false>(p, l0, (vb - b1) * 8, smem, lists, over_cnts);
    }
}

// Small inputs (one launch's worth of partitions: the chip is not full and a partition's own chain of dependent steps is what
// takes the time): the threshold graph and, in the same wavefront, the full-triangle linkage for what it does not settle --
// one launch, nothing handed on.  Measured at 1.0 M marks: 86 us against 70 + 60 us for the two tiers above.
constexpr size_t kFastSmemBytes32 = cmax(sizeof(FastSmem<32, 1>), cmax(sizeof(FastSmem<16, 1>), sizeof(FastSmem<8, 1>)));
constexpr size_t kLinkSmemBytes32 = cmax(sizeof(LinkSmem<32, 1, 32>), cmax(sizeof(LinkSmem<16, 1, 16>), sizeof(LinkSmem<8, 1, 8>)));
constexpr size_t kFastSmemBytes = cmax(sizeof(FastSmem<64, 1>), kFastSmemBytes32);
constexpr size_t kLinkSmemBytes = cmax(sizeof(LinkSmem<64, 1, 64>), kLinkSmemBytes32);

// C64: the class of 33..64 marks is this launch's too (else the four-wavefront units of cl_wide_list have it: the launch holds the scratch of
// 32 rows, 10 KB instead of 19, and all its workgroups are resident at once at 1.0 M marks)
template <bool C64>
__global__ __launch_bounds__(64) void cl_fast_all(const ClParams p, const uint32_t *lists, const uint32_t *cnts /* [kClasses][kShards] */)
{
    CL_STAMP_INIT(4);
    __shared__ __align__(16) unsigned char smem[C64 ? kFastSmemBytes : kFastSmemBytes32];
    __shared__ __align__(16) unsigned char smem_link[C64 ? kLinkSmemBytes : kLinkSmemBytes32];
    __shared__ uint32_t s_pref[kShards + 1];               // the running sums of ONE class's shard counters at a time
    if (p.box_done_flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(p.box_done_flag, p.fork_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    // the classes' totals (one wavefront: a lane per shard)
    uint32_t tot[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint32_t x = cnts[c * kShards + (threadIdx.x & 63u)];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) x += (uint32_t)__shfl_xor((int)x, d, 64);
        tot[c] = (((p.fast_classes >> c) & 1u) && (C64 || c < 3)) ? x : 0u;
    }
    const size_t M = p.M;
    const uint32_t span = p.tps * kScanTile;
    const uint32_t b3 = tot[3], b2 = b3 + (tot[2] + 1) / 2, b1 = b2 + (tot[1] + 3) / 4, b0 = b1 + (tot[0] + 7) / 8;
    int loaded = -1;
    for (uint32_t vb = blockIdx.x; vb < b0; vb += gridDim.x) {
        const int c = vb < b3 ? 3 : (vb < b2 ? 2 : (vb < b1 ? 1 : 0));
        if (c != loaded) {
            __syncthreads();
            worklist_prefix(cnts + c * kShards, s_pref);
            __syncthreads();
            loaded = c;
        }
        const WorkList l{lists + (size_t)c * M, s_pref, span, 0u};
        if (c == 3) {
            if constexpr (C64) fast_unit<64, 1, 64, kFastThenLink>(p, l, vb, smem, smem_link);
        } else if (c == 2) fast_unit<32, 1, 32, kFastThenLink>(p, l, (vb - b3) * 2, smem, smem_link);
        else if (c == 1) fast_unit<16, 1, 16, kFastThenLink>(p, l, (vb - b2) * 4, smem, smem_link);
        else fast_unit<8, 1, 8, kFastThenLink>(p, l, (vb - b1) * 8, smem, smem_link);
    }
}

// The second lists: the contracted linkage again, now with as many groups as rows; what it does not vouch for either
// (coordinates beyond 32 bits, parameters outside the binary32 tests' range) gets the round-by-round linkage on the full
// triangle.  The two use the same scratch one after the other.
constexpr size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
template <int GROUP, int R, int NCAP>
struct Tier2Smem {
    static constexpr size_t link_at = align16(sizeof(FastSmem<GROUP, R>));
    static constexpr size_t bytes = cmax(link_at + sizeof(LinkSmem<GROUP, R, NCAP>), sizeof(TightSmem<GROUP, R, GROUP * R>));
};
template <int GROUP, int R, int NCAP>
__device__ __forceinline__ void tier2_unit(const ClParams &p, const WorkList &list, uint32_t base, unsigned char *smem)
{
    if (tight_unit<GROUP, R, GROUP * R, true>(p, list, base, smem, nullptr, nullptr))
        fast_unit<GROUP, R, NCAP, kLinkUnfit>(p, list, base, smem, smem + Tier2Smem<GROUP, R, NCAP>::link_at);
}

constexpr size_t kTier2SmemBytes = cmax(cmax(Tier2Smem<64, 1, 64>::bytes, Tier2Smem<32, 1, 32>::bytes), cmax(Tier2Smem<16, 1, 16>::bytes, Tier2Smem<8, 1, 8>::bytes));

__global__ __launch_bounds__(64) void cl_tier2_all(const ClParams p, const uint32_t *lists, const uint32_t *over_cnts /* [kClasses][kShards] */,
                                                   uint32_t with64 /* 0: the class of 33..64 marks has a launch of its own */)
{
    CL_STAMP_INIT(0);
    __shared__ __align__(16) unsigned char smem[kTier2SmemBytes];
    __shared__ uint32_t s_pref[4][kShards + 1];
#pragma unroll
    for (int c = 0; c < 4; ++c) worklist_prefix(over_cnts + c * kShards, s_pref[c]);
    __syncthreads();
    const size_t M = p.M;
    const uint32_t span = p.tps * kScanTile;
    const WorkList l0{lists, s_pref[0], span, p.M}, l1{lists + M, s_pref[1], span, p.M}, l2{lists + 2 * M, s_pref[2], span, p.M},
        l3{lists + 3 * M, s_pref[3], span, p.M};
    const uint32_t c0 = l0.size(), c1 = l1.size(), c2 = l2.size(), c3 = with64 ? l3.size() : 0u;
    const uint32_t b3 = c3, b2 = b3 + (c2 + 1) / 2, b1 = b2 + (c1 + 3) / 4, b0 = b1 + (c0 + 7) / 8;
    for (uint32_t vb = blockIdx.x; vb < b0; vb += gridDim.x) {
        if (vb < b3) tier2_unit<64, 1, 64>(p, l3, vb, smem);
        else if (vb < b2) tier2_unit<32, 1, 32>(p, l2, (vb - b3) * 2, smem);
        else if (vb < b1) tier2_unit<16, 1, 16>(p, l1, (vb - b2) * 4, smem);
        else tier2_unit<8, 1, 8>(p, l0, (vb - b1) * 8, smem);
    }
}

template <int GROUP, int R, int NCAP>
__global__ __launch_bounds__(64) void cl_tier2_one(const ClParams p, const uint32_t *items, const uint32_t *over_counts /* [kShards] */)
{
    CL_STAMP_INIT(0);
    __shared__ __align__(16) unsigned char smem[Tier2Smem<GROUP, R, NCAP>::bytes];
    __shared__ uint32_t s_pref[kShards + 1];
    worklist_prefix(over_counts, s_pref);
    __syncthreads();
    const WorkList list{items, s_pref, p.tps * kScanTile, p.M};
    const uint32_t L = list.size();
    for (uint32_t base = blockIdx.x * (64 / GROUP); base < L; base += gridDim.x * (64 / GROUP))
        tier2_unit<GROUP, R, NCAP>(p, list, base, smem);
}

// (the partitions of more than 64 marks: their first tier already takes 64 groups; what it hands on gets the full triangle)
template <int GROUP, int R, int NCAP>
__global__ __launch_bounds__(64) void cl_link_one(const ClParams p, const uint32_t *items, const uint32_t *over_counts /* [kShards] */)
{
    CL_STAMP_INIT(5);
    __shared__ __align__(16) unsigned char smem[sizeof(FastSmem<GROUP, R>)];
    __shared__ __align__(16) unsigned char smem_link[sizeof(LinkSmem<GROUP, R, NCAP>)];
    __shared__ uint32_t s_pref[kShards + 1];
    worklist_prefix(over_counts, s_pref);
    __syncthreads();
    const WorkList list{items, s_pref, p.tps * kScanTile, p.M};
    const uint32_t L = list.size();
    for (uint32_t base = blockIdx.x * (64 / GROUP); base < L; base += gridDim.x * (64 / GROUP))
        fast_unit<GROUP, R, NCAP, kLinkOnly>(p, list, base, smem, smem_link);
}

// ---------------------------------------------------------------------------------------------
// agglomeration, small inputs: one partition on FOUR (EIGHT) wavefronts (round 6)
// ---------------------------------------------------------------------------------------------
//
// On a small input the chip is not full and a launch lasts as long as its longest partition: a 64-mark partition on one wavefront is a
// chain of 70 us (cl_fast_all), a 100-mark one 90 us (cl_tight_big + cl_link_one) -- pair pass, group sums and a dozen rounds, each a run
// of dependent LDS trips (profiles/history/r05_agglomeration_chain_logs_config2.txt).  Here NCAP x 4 threads take ONE partition, four
// lanes per row, and every phase is a handful of loads that leave together:
//   * the sums live in a FULL symmetric matrix (row stride NCAP + 2): a lane's sixteen columns of its row are eight 16-byte loads side
//     by side, with the columns' sizes in one more -- no index arithmetic, no dependent trips (the first version of this kernel kept
//     link_unit's triangle and walked columns one at a time: 4 dependent trips per column, 85 us per 64-mark partition);
//   * a CONTRACTION -- disjoint groups of clusters become one cluster each, rep[c] = the group's head -- is two passes of LDS atomics
//     (sums of integers below 2^53 in binary64: exact, order-free): every row adds its entries at dying columns to the heads' columns,
//     then every dying row adds its entries at surviving columns to its head's row.  Targets are never sources within a pass;
//   * the tight groups of link_unit (components of {q <= threshold / 2} or {/ 4} that are cliques: nodes of the merge tree, its
//     comment has the argument) are one contraction; every round of mutual nearest neighbours is another, with groups of one or two;
//   * nearest neighbours: a row's four lanes scan their sixteen columns each and meet by shuffles (smaller mean, ties to the smaller
//     index: what one lane walking the columns in order finds).
// Exact sums of integers: any evaluation that merges provably-first groups arrives at the oracle's clusters (oracle/cluster_oracle.c,
// rule 4 and the comment above link_unit).  Four barriers per round.
template <int NCAP>
struct WideSmem {
    static constexpr int NW = NCAP > 64 ? 2 : 1, LD = NCAP + 2;
    double S[NCAP * LD];                                     // S[a * LD + b]: the sum over the member pairs of clusters a and b (symmetric; the diagonal is never read)
    double inv[NCAP];
    uint2 ps[NCAP];
    uint32_t mk[NCAP], rd[NCAP];
    uint64_t m1[NCAP][NW], m2[NCAP][NW];                     // neighbourhoods at threshold / 2 and / 4; m1 later: the clusters' rows, at their heads
    unsigned long long sum[NCAP][2];
    uint64_t pass[2][NW], heads[NW], amask[2][NW];           // amask: the clusters that stay, as the step decides it
    uint32_t deaths, na[2];
    uint16_t acl[2][NCAP];                                   // the clusters that are left, ascending: id | size << 8 (a step reads one list and writes the other)
    uint8_t size[NCAP];                                      // by cluster id (what a cluster that is gone leaves here is never read)
    uint8_t rep[NCAP];                                       // this step: the cluster a cluster goes into (itself: it stays)
    uint8_t nn[NCAP], root[NCAP];
};

template <int NW>
__device__ __forceinline__ BitSet<NW> bits_load(const uint64_t (&m)[NW])
{
    BitSet<NW> b;
    for (int i = 0; i < NW; ++i) b.w[i] = m[i];
    return b;
}

// part: a partition of up to NCAP rows, on NCAP * 4 threads; gather: its rows come through the sort permutation (no cl_box before this launch)
template <int NCAP>
__device__ __forceinline__ void wide_unit(const ClParams &p, uint32_t part, bool gather, WideSmem<NCAP> &X)
{
    constexpr int NW = WideSmem<NCAP>::NW, LD = WideSmem<NCAP>::LD, LPR = 4, CH = NCAP / LPR, T = NCAP * LPR;
    static_assert(CH % 16 == 0, "a lane takes its columns sixteen at a time");
    const uint32_t tid = threadIdx.x, row0 = tid / LPR, q0 = tid % LPR;
    const uint32_t s = p.part_start[part], n = p.part_start[part + 1] - s;
    auto shfl64 = [](uint64_t v, int d) -> uint64_t {
        return ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, d, 64);
    };
    double *S = X.S;
    CL_STAMP(0x40);
    __syncthreads();                                         // (the unit before this one is through with the scratch)
    if (tid < (uint32_t)NCAP) {
        uint32_t pk = 0, sp = 0, rdv = 0, mkv = 0;
        if (tid < n) {
            if (gather) {
                mkv = mark_at(p, s + tid);
                const uint3 r3 = load_rec(p, mkv);
                pk = r3.x; sp = r3.y; rdv = r3.z;
            } else {
                const uint4 r4 = p.srec[s + tid];
                pk = r4.x; sp = r4.y; rdv = r4.z; mkv = r4.w & p.idx_mask;
            }
        }
        X.ps[tid] = make_uint2(pk, sp);
        X.rd[tid] = rdv;
        X.mk[tid] = mkv;
        X.inv[tid] = sp ? 1.0 / (double)sp : 0.0;
        X.size[tid] = 1;
        X.acl[0][tid] = (uint16_t)(tid | 0x100u);
        X.rep[tid] = (uint8_t)tid;
        X.root[tid] = (uint8_t)tid;
    }
    if (tid == 0) {
        X.deaths = 0;
        X.na[0] = n;
        for (int l = 0; l < 2; ++l)
            for (int i = 0; i < NW; ++i) { X.pass[l][i] = 0; X.amask[l][i] = 0; }
    }
    __syncthreads();
    CL_STAMP(0x41);
    const bool live = row0 < n;
    // every unordered pair once (link_unit's scheme), a row's columns dealt to its lanes; 64-bit ends and centres throughout
    if (live) {
        const uint32_t half = n >> 1, tmax = (!(n & 1u) && row0 >= half) ? half - 1u : half;
        const uint2 me = X.ps[row0];
        const double ik = X.inv[row0];
        const uint64_t ek = (uint64_t)me.x + me.y, ck = centre_of(me.x, me.y);
        for (uint32_t t = 1u + q0; t <= tmax; t += LPR) {
            uint32_t j = row0 + t;
            j = j >= n ? j - n : j;
            const uint2 o = X.ps[j];
            const double ij = X.inv[j];
            const uint64_t ej = (uint64_t)o.x + o.y, cj = centre_of(o.x, o.y);
            const uint64_t m2 = ek > ej ? ek - ej : ej - ek, m3 = ck > cj ? ck - cj : cj - ck;
            const uint64_t mm = m2 < m3 ? m2 : m3;
            uint32_t m = absdiff_u32(me.x, o.x);
            m = mm < (uint64_t)m ? (uint32_t)mm : m;
            const uint32_t sdif = absdiff_u32(me.y, o.y);
            const double inv = me.y > o.y ? ik : ij;                             // 1 / the larger span
            const double dp = (double)m * p.invn, ds = (double)sdif * inv;
            const double v = quantise(dp + ds, p.scale);
            S[row0 * LD + j] = v;
            S[j * LD + row0] = v;
        }
    }
    __syncthreads();
    CL_STAMP(0x42);
    uint32_t L = 0, seen = 0;
    // A step works on the clusters that are left (list L, na of them): as many lanes per cluster as the threads allow (four when all
    // NCAP are there, up to 64), a cluster's columns -- the list again -- dealt to its lanes one by one.  What a step costs falls with
    // the square of what is left.
    // Once kSolo clusters or fewer are left the first wavefront goes on ALONE (four lanes per cluster and more): what is left of a round is its
    // chain of dependent LDS trips, and without the other wavefronts its barriers are no more than waits for the wave's own LDS traffic
    // (3 us per round with the barriers, whatever is left)
    constexpr uint32_t kSolo = 16;
    uint32_t na = n, lg = 0, row = 0, q = 0, chd = 0, te = T;
    bool has = false, solo = false;
    auto sync = [&]() {
        if (solo) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            __syncthreads();
        }
    };
    auto deal = [&]() {
        na = X.na[L];
        lg = min(6u, 31u - (uint32_t)__clz((int)(te / max(na, 1u))));
        const uint32_t ri = tid >> lg;
        q = tid & ((1u << lg) - 1u);
        has = ri < na;
        row = has ? (uint32_t)X.acl[L][ri] & 0xFFu : 0u;
        chd = (na + (1u << lg) - 1u) >> lg;
    };
    // the contraction that X.rep describes (sizes and X.amask[L ^ 1] say what the clusters that stay look like afterwards)
    auto contract = [&]() {
        const uint16_t *acl = X.acl[L];
        if (has) {
            for (uint32_t u0 = 0; u0 < chd; u0 += 4) {
                uint32_t col[4], h[4];
                double v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t ci = q + ((u0 + k) << lg);
                    col[k] = ci < na ? (uint32_t)acl[ci] & 0xFFu : row;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { h[k] = X.rep[col[k]]; v[k] = S[row * LD + col[k]]; }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (h[k] != col[k] && q + ((u0 + k) << lg) < na) unsafeAtomicAdd(&S[row * LD + h[k]], v[k]);
            }
            if (q == 0 && X.rep[row] == row) {
                const BitSet<NW> stay = bits_load<NW>(X.amask[L ^ 1u]);
                X.acl[L ^ 1u][stay.count_below(row)] = (uint16_t)(row | ((uint32_t)X.size[row] << 8));
            }
        }
        if (tid == 0) X.na[L ^ 1u] = bits_load<NW>(X.amask[L ^ 1u]).count();
        for (uint32_t r = tid; r < n; r += te) X.root[r] = X.rep[X.root[r]];
        sync();
        const uint32_t head = has ? X.rep[row] : row;
        if (head != row) {
            for (uint32_t u0 = 0; u0 < chd; u0 += 4) {
                uint32_t col[4], h[4];
                double v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t ci = q + ((u0 + k) << lg);
                    col[k] = ci < na ? (uint32_t)acl[ci] & 0xFFu : row;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { h[k] = X.rep[col[k]]; v[k] = S[row * LD + col[k]]; }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (h[k] == col[k] && q + ((u0 + k) << lg) < na) unsafeAtomicAdd(&S[head * LD + col[k]], v[k]);     // (col = row: its head is not itself)
            }
        }
        sync();
        L ^= 1u;
    };
    if (p.mergeable) {
        // tight groups (link_unit): neighbourhoods at the two levels; every row is there, four lanes each, sixteen columns at a time
        constexpr double kHalf = (double)(kQOne / 2), kQuarter = (double)(kQOne / 4);
        BitSet<NW> g1, g2;
        g1.clear();
        g2.clear();
        if (live) {
#pragma unroll
            for (int c = 0; c < CH; c += 16) {
                const uint32_t c0 = q0 * CH + c;
                if (c0 >= n) continue;
                double2 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const double2 *>(&S[row0 * LD + c0 + 2 * k]);
                uint32_t b1 = 0, b2 = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint32_t j = c0 + k;
                    const double sv = (k & 1) ? v[k >> 1].y : v[k >> 1].x;
                    const bool in = j < n && j != row0;
                    b1 |= (in && sv <= kHalf) ? 1u << k : 0u;
                    b2 |= (in && sv <= kQuarter) ? 1u << k : 0u;
                }
                for (int i = 0; i < NW; ++i) {
                    g1.w[i] |= (NW == 1 || (c0 >> 6) == (uint32_t)i) ? (uint64_t)b1 << (c0 & 63u) : 0ull;
                    g2.w[i] |= (NW == 1 || (c0 >> 6) == (uint32_t)i) ? (uint64_t)b2 << (c0 & 63u) : 0ull;
                }
            }
        }
#pragma unroll
        for (int d = 1; d < LPR; d <<= 1) {
            for (int i = 0; i < NW; ++i) {
                g1.w[i] |= shfl64(g1.w[i], d);
                g2.w[i] |= shfl64(g2.w[i], d);
            }
        }
        g1.set_if(live, row0);
        g2.set_if(live, row0);
        if (live && q0 == 0) {
            for (int i = 0; i < NW; ++i) { X.m1[row0][i] = g1.w[i]; X.m2[row0][i] = g2.w[i]; }
        }
        __syncthreads();
        bool ok1 = false, ok2 = false;
        if (live) {
            ok1 = bits_load<NW>(X.m1[g1.first()]).equals(g1);
            ok2 = bits_load<NW>(X.m2[g2.first()]).equals(g2);
            if (q0 == 0) {
                if (ok1) atomicOr((unsigned long long *)&X.pass[0][NW == 1 ? 0 : (row0 >> 6)], 1ull << (row0 & 63u));
                if (ok2) atomicOr((unsigned long long *)&X.pass[1][NW == 1 ? 0 : (row0 >> 6)], 1ull << (row0 & 63u));
            }
        }
        __syncthreads();
        if (live) {
            // a row's component is a clique <=> the row and all its neighbours have the neighbourhood of their smallest member
            const BitSet<NW> p1 = bits_load<NW>(X.pass[0]), p2 = bits_load<NW>(X.pass[1]);
            for (int i = 0; i < NW; ++i) {
                ok1 = ok1 && (g1.w[i] & ~p1.w[i]) == 0ull;
                ok2 = ok2 && (g2.w[i] & ~p2.w[i]) == 0ull;
            }
            const uint32_t f = ok1 ? g1.first() : (ok2 ? g2.first() : row0);
            if (q0 == 0) {
                X.rep[row0] = (uint8_t)f;
                if (f == row0) {
                    X.size[row0] = (uint8_t)(ok1 ? g1.count() : (ok2 ? g2.count() : 1u));
                    atomicOr((unsigned long long *)&X.amask[1][NW == 1 ? 0 : (row0 >> 6)], 1ull << (row0 & 63u));
                } else {
                    atomicAdd(&X.deaths, 1u);
                }
            }
        }
        __syncthreads();
        seen = X.deaths;
        CL_STAMP(0x43);
        if (seen) {
            deal();
            contract();
        }
        CL_STAMP(0x44);
    }
    // rounds of mutual nearest neighbours
    for (uint32_t round = 0; round < 2u * NCAP && p.mergeable; ++round) {
        CL_STAMP(0x45);
        if (!solo && X.na[L] <= kSolo) {
            if (tid >= 64u) break;                                               // (the other wavefronts wait in front of the emission)
            solo = true;
            te = 64u;
        }
        deal();
        if (na < 2u) break;
        const uint16_t *acl = X.acl[L];
        double bs = kNothing, bn = 1.0;
        uint32_t bk = 0xFFu;
        if (has) {
            for (uint32_t u0 = 0; u0 < chd; u0 += 4) {
                uint32_t e[4];
                double v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t ci = q + ((u0 + k) << lg);
                    e[k] = ci < na ? (uint32_t)acl[ci] : row;                    // (beyond the list: the cluster itself, which is skipped)
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = S[row * LD + (e[k] & 0xFFu)];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t j = e[k] & 0xFFu;
                    const double sv = (j != row && v[k] <= kFar) ? v[k] : kNothing;
                    const double nk = (double)(e[k] >> 8);
                    const bool better = sv * bn < bs * nk;
                    bs = better ? sv : bs;
                    bn = better ? nk : bn;
                    bk = better ? j : bk;
                }
            }
        }
        CL_STAMP(0x48);
        for (uint32_t d = 1; d < (1u << lg); d <<= 1) {
            const double os = __shfl_xor(bs, (int)d, 64), on = __shfl_xor(bn, (int)d, 64);
            const uint32_t ok = (uint32_t)__shfl_xor((int)bk, (int)d, 64);
            const double l = os * bn, r = bs * on;
            const bool take = l < r || (l == r && ok < bk);
            bs = take ? os : bs;
            bn = take ? on : bn;
            bk = take ? ok : bk;
        }
        CL_STAMP(0x49);
        if (has && q == 0) {
            const bool within = bs <= (double)kQOne * (bn * (double)X.size[row]);
            X.nn[row] = (uint8_t)(within ? bk : 0xFFu);
        }
        if (tid == 0)
            for (int i = 0; i < NW; ++i) X.amask[L ^ 1u][i] = 0;
        sync();
        CL_STAMP(0x4A);
        // mutual nearest neighbours merge, the larger index into the smaller
        if (has && q == 0) {
            const uint32_t k = X.nn[row];
            const uint32_t pm = (k != 0xFFu && X.nn[k] == row) ? k : 0xFFu;
            const bool dies = pm < row;                                           // (0xFF is not below any row)
            X.rep[row] = (uint8_t)(dies ? pm : row);
            if (dies) {
                atomicAdd(&X.deaths, 1u);
            } else {
                if (pm != 0xFFu) X.size[row] = (uint8_t)((uint32_t)X.size[row] + (uint32_t)X.size[pm]);      // (pm goes: it writes nothing, nobody else reads this)
                atomicOr((unsigned long long *)&X.amask[L ^ 1u][NW == 1 ? 0 : (row >> 6)], 1ull << (row & 63u));
            }
        }
        sync();
        CL_STAMP(0x4B);
        const uint32_t d = X.deaths;
        if (d == seen) break;
        seen = d;
        contract();
        CL_STAMP(0x4C);
    }
    // the clusters' rows and sums at their heads (the head is the cluster's smallest row)
    CL_STAMP(0x46);
    __syncthreads();                                         // (everybody again)
    if (tid < (uint32_t)NCAP) {
        for (int i = 0; i < NW; ++i) X.m1[tid][i] = 0;
        X.sum[tid][0] = 0;
        X.sum[tid][1] = 0;
    }
    if (tid == 0)
        for (int i = 0; i < NW; ++i) X.heads[i] = 0;
    __syncthreads();
    if (tid < n) {
        const uint32_t r = X.root[tid];
        const uint2 me = X.ps[tid];
        atomicOr((unsigned long long *)&X.m1[r][NW == 1 ? 0 : (tid >> 6)], 1ull << (tid & 63u));
        atomicAdd(&X.sum[r][0], (unsigned long long)me.x);
        atomicAdd(&X.sum[r][1], (unsigned long long)me.y);
        if (r == tid) atomicOr((unsigned long long *)&X.heads[NW == 1 ? 0 : (tid >> 6)], 1ull << (tid & 63u));
    }
    __syncthreads();
    // what cl_emit needs (emit_prep): every mark's place in the output, the clusters' records
    if (tid < n) {
        const uint32_t rt = X.root[tid];
        const BitSet<NW> F = bits_load<NW>(X.m1[rt]), H = bits_load<NW>(X.heads);
        BitSet<NW> below = H;
        uint32_t before = 0;
        while (below.any()) {
            const uint32_t h = below.pop_first();
            if (h >= rt) break;
            before += bits_load<NW>(X.m1[h]).count();
        }
        const uint32_t size = F.count(), rank = before + F.count_below(tid);
        p.order[s + rank] = X.mk[tid];
        if (p.sv_mark_out) p.sv_mark_out[s + rank] = X.rd[tid];
        if (rt == tid) {
            const uint32_t ci = H.count_below(rt);
            uint4 *at_ = ci ? p.e_rec + (s + ci) : p.e_first + part;
            *at_ = make_uint4(rank | ((before + size) << 8), (uint32_t)((double)X.sum[tid][0] / (double)size),
                              (uint32_t)((double)X.sum[tid][1] / (double)size), p.rec_mode ? p.srec[s].w >> p.idx_bits : 0u);
        }
        if (tid == 0) p.pc[part] = H.count();
    }
    CL_STAMP(0x47);
}

// the partitions of more than 64 marks, found from the partition starts alone (cl_tight_big's scheme: the launches start beside cl_box) and
// listed first: a workgroup per listed partition (one that found two in its own stretch of partitions took them one after the other --
// 70 us for the launch where a unit takes 35)
__global__ __launch_bounds__(256) void cl_find_big(const ClParams p, uint32_t *big_list, uint32_t *count /* zero on entry */)
{
    const uint32_t n_parts = *p.n_parts;
    for (uint32_t part = blockIdx.x * 256u + threadIdx.x; part < n_parts; part += gridDim.x * 256u)
        if (p.part_start[part + 1] - p.part_start[part] > 64u) big_list[atomicAdd(count, 1u)] = part;
}
__global__ __launch_bounds__(512) void cl_wide_big(const ClParams p, const uint32_t *big_list, const uint32_t *count)
{
    CL_STAMP_INIT(3);
    __shared__ WideSmem<128> X;
    const uint32_t cnt = *count;
    for (uint32_t i = blockIdx.x; i < cnt; i += gridDim.x) wide_unit<128>(p, big_list[i], p.gather_rows != 0u, X);
}

// the listed partitions of the classes c_lo .. 3 (up to 64 marks; their rows where cl_box laid them out).  Tests only (DUET_DBG_CLUSTER_WIDE_ALL):
// measured at 1.0 M marks the four-wavefront units LOSE on these partitions, whatever sizes they take (profiles/history/r06_wide_units.txt)
__global__ __launch_bounds__(256) void cl_wide_list(const ClParams p, const uint32_t *lists, const uint32_t *cnts /* [kClasses][kShards] */, uint32_t c_lo)
{
    CL_STAMP_INIT(5);
    __shared__ WideSmem<64> X;
    __shared__ uint32_t s_pref[kShards + 1];
    const uint32_t span = p.tps * kScanTile;
    for (uint32_t c = 3; c + 1u > c_lo; --c) {
        __syncthreads();
        if (threadIdx.x < 64u) worklist_prefix(cnts + c * kShards, s_pref);
        __syncthreads();
        const WorkList l{lists + (size_t)c * p.M, s_pref, span, 0u};
        const uint32_t L = l.size();
        for (uint32_t i = blockIdx.x; i < L; i += gridDim.x) wide_unit<64>(p, l[i], false, X);
        if (c == 0u) break;
    }
}

// One LANE PER CLUSTER: a wavefront takes 64 consecutive partitions (their count lives on the device: the grid strides over an upper
// bound), and their clusters -- records dense from each partition's start, candidates cbase[part] ... -- are dealt to the lanes
// 64 at a time: a lane finds its cluster's partition among the wave's 64 by bisection over the lanes' candidate bases (six
// bpermutes, no memory), loads the record and the depth bin it selects, and writes its candidate -- neighbouring lanes write
// neighbouring candidates, and a partition of ten clusters costs what one of one does (round 3: one thread per partition
// walking its clusters four at a time, 76 us at 2e7 marks).
// A fork without an event (round 6).  hipEventRecord on the main stream + hipStreamWaitEvent on the side stream cost the MAIN stream
// 7-14 us each time (the record's barrier packet sits in front of its next kernel: timelines of rounds 4-6).  Instead the main stream
// runs a one-lane kernel that writes the run's epoch to a device word, and the side stream -- whose launches were queued long before --
// starts with a one-lane kernel that waits for that word.  The main stream never waits for the side stream here (the JOIN stays an
// event), so a side queue that the device schedules late only starts late: nothing can deadlock; the wait is bounded all the same
// (2 s, then the kernel traps: a loud failure, not a hang).  Kernel boundaries on both queues do the cache maintenance around it.
__global__ void cl_signal(uint32_t *flag, uint32_t epoch)
{
    if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void cl_gate(const uint32_t *flag, uint32_t epoch)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    // (epochs only grow; a later run's value serves as well: its main stream is behind this run's)
    while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > 200000000ull) __builtin_trap();
    }
}

// A kernel that waits for another queue's kernel needs the device to run the two side by side.  Where it takes ONE kernel at a time -- a profiler
// collecting counters (rocprofv3 --pmc serialises the dispatches: round 6's last collection hung there), AMD_SERIALIZE_KERNEL, a debugger -- a gate at the
// head of its queue can be picked before the kernel that opens it, deep in the other queue, and then waits for ever.  So a context tries it out
// once, bounded, before its first gated run: two kernels on one side stream, then the gate at the head of ANOTHER (up to 20 ms), then the kernel that
// opens it behind the first two.  Only a context that saw the gate open forks through gates; the others fork behind events.
__global__ void cl_gate_try(const uint32_t *flag, uint32_t epoch, uint32_t *opened)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    uint32_t ok = 1;
    while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > 2000000ull) { ok = 0; break; }
    }
    *opened = ok;
}

// (the join of two side streams in one launch)
__global__ void cl_gate2(const uint32_t *flag_a, const uint32_t *flag_b, uint32_t epoch)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while ((int32_t)(__hip_atomic_load(flag_a, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0 ||
           (int32_t)(__hip_atomic_load(flag_b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > 200000000ull) __builtin_trap();
    }
}

// clusters per partition -> their sums over every 64 partitions and over every 2048: what cl_emit needs to number the candidates itself (rounds 1-5: a
// generic exclusive scan in two launches, then cl_emit -- three launches, 5 us apiece at 1.0 M marks whatever they do).  gate (or null): the launch is also the
// JOIN of the side stream -- every workgroup waits for the word the side stream's last kernel writes (cl_gate's bounded wait; one launch less on the main
// stream, and at 1.0 M marks the side chain is through 25 us before this starts).
__global__ __launch_bounds__(256) void cl_pc_sums(const uint32_t *pc, const uint32_t *n_parts_p, uint32_t *csum, uint32_t *tsum, const uint32_t *gate, uint32_t epoch)
{
    __shared__ uint32_t s_w[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_parts = *n_parts_p, base = blockIdx.x * 2048u;       // (the main stream's own: it leaves beside the gate's first load)
    if (base >= n_parts) return;
    if (gate) {
        if (tid == 0) {
            const unsigned long long t0 = wall_clock64();
            while ((int32_t)(__hip_atomic_load(gate, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
                __builtin_amdgcn_s_sleep(16);
                if (wall_clock64() - t0 > 200000000ull) __builtin_trap();
            }
        }
        __syncthreads();
    }
    uint32_t tot = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint32_t i = base + (wave * 8u + c) * 64u + lane;
        uint32_t v = i < n_parts ? pc[i] : 0u;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
        if (lane == 0) csum[blockIdx.x * 32u + wave * 8u + c] = v;
        tot += v;
    }
    if (lane == 0) s_w[wave] = tot;
    __syncthreads();
    if (tid == 0) tsum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(256) void cl_emit(const ClParams p)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) p.cand_off[0] = 0;
    // Step E/F's plan (fused pipeline): the first candidate of every contig is where the candidates' contig changes -- the
    // candidate there writes the offsets of the contigs it opens (its own and the empty ones before it), the last candidate
    // those behind it; the per-contig seed counts and the status words are zeroed here.  (A launch of its own for K + 1
    // binary searches cost 6 us at 1.0 M marks.)
    const bool plan = p.ef_ctg_off != nullptr;
    if (plan && blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < p.n_contigs + 8u; i += blockDim.x) p.ef_zero[i] = 0u;
    const uint32_t n_parts = *p.n_parts, lane = threadIdx.x & 63u;
    // the candidates' number: every wave adds up the tiles' sums (a few hundred at 2e7 marks); the first one leaves it for step E/F
    uint32_t n_cands = 0;
    for (uint32_t i = lane, nt = (n_parts + 2047u) >> 11; i < nt; i += 64u) n_cands += p.tsum[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) n_cands += (uint32_t)__shfl_xor((int)n_cands, d, 64);
    if (blockIdx.x == 0 && threadIdx.x == 0) *p.n_cands_w = n_cands;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t b0 = wave * 64u; b0 < n_parts; b0 += n_waves * 64u) {
        const uint32_t part = b0 + lane;
        const bool has = part < n_parts;
        const uint32_t s = has ? p.part_start[part] : 0u, nc = has ? p.pc[part] : 0u;
        // the first candidate of every partition of the wave: the tiles in front, the 64s in front within the tile, the lanes in front
        uint32_t c0;
        {
            const uint32_t tile = b0 >> 11, cin = (b0 >> 6) & 31u;
            uint32_t acc = lane < cin ? p.csum[(tile << 5) + lane] : 0u;
            for (uint32_t i = lane; i < tile; i += 64u) acc += p.tsum[i];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) acc += (uint32_t)__shfl_xor((int)acc, d, 64);
            uint32_t x = nc;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = (uint32_t)__shfl_up((int)x, d, 64);
                if ((int)lane >= d) x += y;
            }
            c0 = acc + x - nc;
        }
        uint32_t hi = 0;                                        // contig | type: in the cluster records (record sort), else from the sorted key
        if (has && !p.rec_mode) hi = (uint32_t)((p.skeys[s] & key_mask(p.key_bits)) >> p.centre_bits);
        // (the plan: the contig of the partition in front of the wave's first one, through lane 0)
        uint32_t k_before = 0xFFFFFFFFu;
        if (plan && lane == 0 && b0 > 0) {
            k_before = (p.rec_mode ? p.e_first[b0 - 1u].w
                                   : (uint32_t)((p.skeys[p.part_start[b0 - 1u]] & key_mask(p.key_bits)) >> p.centre_bits)) >> p.type_bits;
        }
        const uint32_t base = (uint32_t)__shfl((int)c0, 0, 64);
        const uint32_t n_has = min(64u, n_parts - b0);
        const uint32_t tot = (uint32_t)__shfl((int)(c0 + nc), (int)n_has - 1, 64) - base;
        const uint32_t rel = has ? c0 - base : 0xFFFFFFFFu;     // (ascending over the lanes: every partition has a cluster)
        for (uint32_t k0 = 0; k0 < tot; k0 += 64u) {
            const uint32_t c = k0 + lane;
            uint32_t lo = 0;                                    // the last lane whose first cluster is <= c
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) {
                const uint32_t at = lo + (uint32_t)step;
                const uint32_t v = (uint32_t)__shfl((int)rel, (int)min(at, 63u), 64);
                lo = (at < 64u && v <= c) ? at : lo;
            }
            const uint32_t ps = (uint32_t)__shfl((int)s, (int)lo, 64), pr = (uint32_t)__shfl((int)rel, (int)lo, 64);
            const uint32_t ph = (uint32_t)__shfl((int)hi, (int)lo, 64);
            const bool on = c < tot;
            uint4 rec = make_uint4(0u, 0u, 0u, 0u);
            if (on) rec = c == pr ? p.e_first[b0 + lo] : p.e_rec[ps + (c - pr)];
            const uint32_t ct = p.rec_mode ? rec.w : ph;
            const uint32_t k = ct >> p.type_bits, type = ct & ((1u << p.type_bits) - 1u);
            if (plan) {
                uint32_t kp = (uint32_t)__shfl_up((int)k, 1, 64);
                if (lane == 0) kp = k_before;
                k_before = (uint32_t)__shfl((int)k, 63, 64);    // (the next 64 clusters, if there are any: this lot was full)
                const uint32_t cand = base + c;
                if (on && k != kp)
                    for (uint32_t kk = kp + 1u; kk <= k; ++kk) p.ef_ctg_off[kk] = cand;
                if (on && cand + 1u == n_cands)
                    for (uint32_t kk = k + 1u; kk <= p.n_contigs; ++kk) p.ef_ctg_off[kk] = n_cands;
            }
            if (!on) continue;
            const uint32_t info = rec.x, cand = base + c;
            p.cand_off[cand + 1] = ps + (info >> 8);
            p.cand_contig[cand] = (uint16_t)k;
            p.cand_type[cand] = (uint8_t)type;
            p.cand_pos[cand] = rec.y;
            p.cand_span[cand] = rec.z;
            if (p.sv_svread) {
                // what a caller VCF would have carried: support = members, reference reads = depth(contig, pos) - support
                const uint32_t d_lo = p.sv_depth_off[k], nb = p.sv_depth_off[k + 1] - d_lo;
                uint32_t bin = rec.y / p.sv_depth_bin;
                bin = bin < nb ? bin : nb - 1;
                const uint32_t d = nb ? p.sv_depth[d_lo + bin] : 0u;
                const uint32_t support = (info >> 8) - (info & 0xFFu);                  // a cluster's end - its first member's rank
                p.sv_svread[cand] = support;
                p.sv_refread[cand] = d > support ? d - support : 0u;
                p.sv_gt[cand] = 1;
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
    uint32_t *ef_ctg_off, *ef_zero;       // step E/F's plan, written by cl_emit (null: E/F plans for itself)
    uint32_t n_contigs;
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
    int rc;
    if ((rc = duet_reserve(ctx, ctx->cl_ws[11], 4 * ((size_t)64 + 2 * kClasses * kShards + M / 64 + 16)))) return rc;
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
    // The record travels with the key (duet_recsort.hip.h) when contig, type and mark index fit one word of it
    const uint32_t idx_bits = bits_for(M - 1);
    // ... from 1.25 M marks on: below, every pass is launch-latency bound and 8-byte keys are the cheaper thing to move (measured
    // on the fused pipeline, round 4: 1.0 M marks, one contig 0.277 ms with the key sort against 0.287; 2 M marks over 24 contigs 0.369
    // against 0.316, 4 M 0.483 / 0.455, 8 M 0.946 / 0.869, 2e7 2.35 / 2.22; round 6, 24 contigs, key sort / record sort: 0.9 M 0.252 / 0.259,
    // 1.1 M 0.253 / 0.253, 1.2 M 0.229 / 0.227, 1.3 M 0.283 / 0.249, 1.5 M 0.276 / 0.240 -- rounds 4-5 drew the line at 1.5 M);
    // DUET_DBG_CLUSTER_RECSORT takes it at any size
    const bool rec_mode = contig_bits + p.type_bits + idx_bits <= 32u &&
                          (M >= 1250000u || (ctx->dbg & (DUET_DBG_CLUSTER_RECSORT | DUET_DBG_CLUSTER_LARGE))) &&
                          !(ctx->dbg & (DUET_DBG_CLUSTER_PAIRS | DUET_DBG_CLUSTER_LSD | DUET_DBG_CLUSTER_KEYSORT));
    p.rec_mode = rec_mode ? 1u : 0u;
    p.idx_bits = idx_bits;
    p.idx_mask = rec_mode ? (uint32_t)((1ull << idx_bits) - 1ull) : 0xFFFFFFFFu;
    // the digits of the record sort: LSD passes of up to kRsMaxW bits over the key's top bits, so many of them that the marks
    // that agree in them (a GROUP: one type within 2^lo centres) are a few dozen; the low bits are ordered group by group
    uint32_t rs_lo = 0, rs_np = 0, rs_w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (rec_mode) {
        // (measured at 2e7 marks, 34-bit keys: 20 top bits in two 10-bit passes leave groups of 53 marks on average and the local
        // stage takes 285 + 33 us; 21 bits 235 us; 22 bits, two 11-bit passes, 212 us but 33 us more in the second scatter and
        // its offsets: 21 it is.  1.0 M marks: 16 bits in two 8-bit passes, groups of 32)
        uint32_t T = std::max(bits_for(M >> 4), 8u);
        T = std::min(T, key_bits);
        if (key_bits - T > 31u) T = key_bits - 31u;            // (the local stage holds the low bits in 32-bit words)
        rs_lo = key_bits - T;
        rs_np = (T + kRsMaxW - 1u) / kRsMaxW;
        for (uint32_t i = 0; i < rs_np; ++i) rs_w[i] = T / rs_np + (i < T % rs_np ? 1u : 0u);
    }
    const uint32_t nb_rs = (M + kRsTile - 1) / kRsTile, rs_chunks = (nb_rs + kRsChunk - 1) / kRsChunk;
    {
        const size_t hist_legacy = (size_t)256 * nb_rx * 4, hist_rec = ((size_t)nb_rs + rs_chunks + 1) * (4u << kRsMaxW);
        const size_t sizes[16] = {rec_mode ? 16 : (size_t)M * 8, rec_mode ? 16 : (size_t)M * 8, (size_t)M * 4, rec_mode ? (size_t)M * 2 + 16 : (size_t)M * 4,
                                  rec_mode ? hist_rec : hist_legacy,
                                  ((size_t)nb_sc + 1) * sizeof(PartSum) + 16 + (size_t)nb_sc * kScanThreads, ((size_t)nb_sc + 2) * 4, ((size_t)(nb_sc > nb_hs ? nb_sc : nb_hs) + 1) * 4,
                                  ((size_t)M + 1) * 4, (size_t)M * 16, (size_t)M * 4, 4 * ((size_t)64 + 2 * kClasses * kShards + M / 64 + 16), (size_t)M * 4 * kClasses,
                                  ((size_t)M + 1) * 4 * 2 + 16 + (size_t)M * ((sv || rec_mode) ? 16 : 8), (size_t)M * 16, (size_t)M * 16};
        for (int i = 0; i < 16; ++i)
            if ((rc = duet_reserve(ctx, ctx->cl_ws[i], sizes[i]))) return rc;
    }
    scal = (uint32_t *)ctx->cl_ws[11].ptr;
    uint64_t *keysA = (uint64_t *)ctx->cl_ws[0].ptr, *keysB = (uint64_t *)ctx->cl_ws[1].ptr;
    uint32_t *valsA = (uint32_t *)ctx->cl_ws[2].ptr, *valsB = (uint32_t *)ctx->cl_ws[3].ptr;
    uint32_t *hist = (uint32_t *)ctx->cl_ws[4].ptr;
    uint32_t *tmpA = (uint32_t *)ctx->cl_ws[5].ptr;             // the partition scan's tile summaries
    uint32_t *tile_first = (uint32_t *)ctx->cl_ws[6].ptr;       // [tiles + 1] first partition starting in each scan tile
    uint32_t *spart = (uint32_t *)ctx->cl_ws[7].ptr;
    uint32_t *part_start = (uint32_t *)ctx->cl_ws[8].ptr;
    uint32_t *cbase = (uint32_t *)ctx->cl_ws[13].ptr + (M + 1);      // (the first M + 1 words hold label8 / comp8)
    uint32_t *pc = (uint32_t *)ctx->cl_ws[10].ptr;

    // outputs (the agglomeration kernels write the marks' output order themselves)
    if (sv) {
        p.sv_mark_in = sv->mark_in; p.sv_depth = sv->depth; p.sv_depth_off = sv->depth_off; p.sv_depth_bin = sv->depth_bin;
        p.sv_mark_out = sv->mark_out; p.sv_svread = sv->svread; p.sv_refread = sv->refread; p.sv_gt = sv->gt;
        p.ef_ctg_off = sv->ef_ctg_off; p.ef_zero = sv->ef_zero; p.n_contigs = sv->n_contigs;
    }
    p.order = res->order; p.cand_off = res->cand_off; p.cand_pos = res->cand_pos; p.cand_span = res->cand_span;
    p.cand_contig = res->cand_contig; p.cand_type = res->cand_type;

    const dim3 g256((M + 255) / 256), b256(256);
    unsigned char *recs = (unsigned char *)ctx->cl_ws[13].ptr + ((8 * ((size_t)M + 1) + 15) & ~(size_t)15);
    if (sv) p.rec4 = (const uint4 *)recs;
    else p.ps = (const uint2 *)recs;
    const bool big_sort = (ctx->dbg & DUET_DBG_CLUSTER_LARGE) != 0;     // (tests: the tile-offset path of > 4 M keys)
    const bool rx_totals = nb_rx <= kRxTotalsTiles && !big_sort;
    // Small inputs whose keys would take four passes or more: global passes over the top 16 bits only, then the groups of keys
    // that agree in them -- the marks of one type within 2^(key_bits - 16) centres, a few dozen -- are put in order by their low
    // bits locally (rx_local: a rank count in LDS; rx_big for a group of more than 256 keys).  8 launches instead of 13.
    // (The first attempt at this, earlier in round 3, ordered the groups with three ballot-ranked LDS radix passes and cost
    // what the global passes it replaced did: 91 against 88 us at 1 M marks; on the uneven groups of 2e7 marks over a whole
    // genome it was slower, 1010 against 745 us, and large inputs keep the plain LSD passes.)
    const bool small_in = M <= (4u << 20) && !(ctx->dbg & DUET_DBG_CLUSTER_LARGE);
    // (the global passes take the top 16 bits of a small input's keys, the top 24 of a large one's: groups of a few keys to a
    // few dozen either way on a genome; used where that saves two passes or more)
    const uint32_t top_bits = M <= (2u << 20) ? 16u : 24u;     // (16 bits leave groups of a hundred keys and more beyond 2 M marks)
    // (rx_local and rx_big hold the low bits in 32-bit words: wider keys than top_bits + 31 take plain LSD passes)
    const bool hybrid = !rec_mode && p.idx_packed && (key_bits + 7u) / 8u >= top_bits / 8u + 2u && key_bits - top_bits < 32u && !(ctx->dbg & DUET_DBG_CLUSTER_LSD);
    const uint32_t top_shift = hybrid ? key_bits - top_bits : 0u;
    uint32_t *big_count = scal + 40, *big_list = valsA;          // (the value buffers are idle when the index rides in the key)
    const uint32_t loc_cap = (ctx->dbg & DUET_DBG_CLUSTER_SMALLCAP) ? 3u : (uint32_t)kLocHalo;
    // large inputs, record sort: cl_box is also the partition scan's last stage (no part_apply launch)
    // (small inputs keep part_apply: there the chain of the partitions of more than 64 marks is the critical path and has to start
    // beside the box test, not behind it -- measured: 315 us against 287 at 1.0 M marks)
    // (cl_box<.., true> reads tiles[tile] as the EXCLUSIVE carry part_spine leaves: it applies exactly where the spine launch runs,
    // i.e. never together with the spine-less part_apply<true> of `nb_sc <= kSelfSpine && !big_sort`)
    static_assert((4u << 20) == kSelfSpine * (uint32_t)kScanTile, "small_in's bound is the spine-less scan's reach: box_applies needs part_spine's carries");
    const bool use_spine = nb_sc > kSelfSpine || big_sort;
    const bool box_applies = rec_mode && (!small_in || big_sort) && use_spine;
    PartSum *tiles = (PartSum *)tmpA;                             // the partition scan's tile summaries: 3 words per 2048 marks
    uint8_t *hbits = (uint8_t *)tmpA + ((((size_t)nb_sc + 1) * sizeof(PartSum) + 15) & ~(size_t)15);      // ... and the head flags, a bit per mark
    p.e_rec = (uint4 *)ctx->cl_ws[9].ptr;
    p.e_first = (uint4 *)ctx->cl_ws[15].ptr;
    if (rec_mode) {
        // the record sort (duet_recsort.hip.h): the first pass reads the caller's arrays, every pass moves 16-byte records
        RsSrc src;
        src.contig = pr->mark_contig; src.type = pr->mark_type; src.pos = pr->mark_pos; src.span = pr->mark_span;
        src.read = sv ? sv->mark_in : nullptr;
        src.type_bits = p.type_bits; src.idx_bits = idx_bits; src.centre_bits = p.centre_bits;
        uint4 *buf[2] = {(uint4 *)ctx->cl_ws[14].ptr, (uint4 *)recs};
        const bool rs_small = nb_rs <= kRsSmallTiles && !big_sort;
        uint32_t *dtot = ctx->rx_dtot;                             // [kRsDtotCopies][2048]
        // (every pass's rs_scatter zeroes the totals behind itself; a run cut short between a histogram and its scatter -- a launch
        // failure, an early return on a HIP error -- would leave them dirty for every later sort on this context: ~1 us for not
        // depending on that)
        HIP_TRY(ctx, hipMemsetAsync(dtot, 0, (size_t)kRsDtotCopies * 2048u * sizeof(uint32_t), st));
        uint32_t *partial = hist + ((size_t)nb_rs << kRsMaxW);
        uint16_t *dig = (uint16_t *)valsB;                        // the next pass's digit of every record (rs_scatter -> rs_hist_dig)
        const uint4 *rin = nullptr;
        int at = 0;                                               // the buffer the next launch writes
        uint32_t shift = rs_lo;
        for (uint32_t ps = 0; ps < rs_np; ++ps) {
            const uint32_t w = rs_w[ps];
            uint32_t *htot = rs_small ? dtot : (uint32_t *)nullptr;
            if (ps == 0) hipLaunchKernelGGL(rs_hist<true>, dim3(nb_rs), dim3(kRsHistThreads), 0, st, src, (const uint4 *)nullptr, M, shift, w, hist, htot, big_count);
            else hipLaunchKernelGGL(rs_hist_dig, dim3(nb_rs), dim3(kRsDigThreads), 0, st, (const uint16_t *)dig, M, w, hist, htot);
            uint16_t *dig_out = ps + 1 < rs_np ? dig : (uint16_t *)nullptr;
            const uint32_t nshift = shift + w, nmask = ps + 1 < rs_np ? (1u << rs_w[ps + 1]) - 1u : 0u;
            if (rs_small) {
                hipLaunchKernelGGL(rs_offsets_small, dim3(std::max(1u, (1u << w) / 64u)), dim3(1024), 0, st, hist, nb_rs, w, (const uint32_t *)dtot);
            } else {
                hipLaunchKernelGGL(rs_col_reduce, dim3(rs_chunks), dim3(256), 0, st, (const uint32_t *)hist, nb_rs, w, partial, dtot);
                hipLaunchKernelGGL(rs_offsets_small, dim3(std::max(1u, (1u << w) / 64u)), dim3(1024), 0, st, partial, rs_chunks, w, (const uint32_t *)dtot);
            }
            if (ps == 0) {
                if (w <= 10u) hipLaunchKernelGGL((rs_scatter<10, true>), dim3(nb_rs), dim3(kRsThreads), 0, st, src, (const uint4 *)nullptr, M, shift, w, nb_rs, (const uint32_t *)hist, buf[at], dtot, dig_out, nshift, nmask, rs_small ? (const uint32_t *)nullptr : (const uint32_t *)partial);
                else hipLaunchKernelGGL((rs_scatter<kRsMaxW, true>), dim3(nb_rs), dim3(kRsThreads), 0, st, src, (const uint4 *)nullptr, M, shift, w, nb_rs, (const uint32_t *)hist, buf[at], dtot, dig_out, nshift, nmask, rs_small ? (const uint32_t *)nullptr : (const uint32_t *)partial);
            } else {
                if (w <= 10u) hipLaunchKernelGGL((rs_scatter<10, false>), dim3(nb_rs), dim3(kRsThreads), 0, st, src, rin, M, shift, w, nb_rs, (const uint32_t *)hist, buf[at], dtot, dig_out, nshift, nmask, rs_small ? (const uint32_t *)nullptr : (const uint32_t *)partial);
                else hipLaunchKernelGGL((rs_scatter<kRsMaxW, false>), dim3(nb_rs), dim3(kRsThreads), 0, st, src, rin, M, shift, w, nb_rs, (const uint32_t *)hist, buf[at], dtot, dig_out, nshift, nmask, rs_small ? (const uint32_t *)nullptr : (const uint32_t *)partial);
            }
            rin = buf[at];
            at ^= 1;
            shift += w;
        }
        if (rs_lo > 0) {
            const KeyOfRec keyof{p.centre_bits, idx_bits};
            if (small_in)
                hipLaunchKernelGGL((rx_local<1024, uint4, KeyOfRec>), dim3((M + kLocTile - 1) / kLocTile), dim3(1024), 0, st, rin, buf[at], M, rs_lo, keyof, loc_cap, big_list, big_count);
            else
                hipLaunchKernelGGL((rx_local<256, uint4, KeyOfRec>), dim3((M + kLocTile - 1) / kLocTile), dim3(256), 0, st, rin, buf[at], M, rs_lo, keyof, loc_cap, big_list, big_count);
            hipLaunchKernelGGL((rx_big<uint4, KeyOfRec>), dim3(256), dim3(256), 0, st, (uint4 *)rin, buf[at], M, rs_lo, keyof, (const uint32_t *)big_list, (const uint32_t *)big_count);
            rin = buf[at];
        }
        p.srec = (uint4 *)rin;
        p.rec4 = nullptr; p.ps = nullptr;
        const LoadHead<uint4, KeyOfRec> heads{rin, p.centre_bits, p.part_gap, KeyOfRec{p.centre_bits, idx_bits}};
        hipLaunchKernelGGL((part_reduce<uint4, KeyOfRec>), dim3(nb_sc), dim3(kScanThreads), 0, st, heads, M, p.part_max, tiles, scal + 64, (uint32_t)(2 * kClasses * kShards), hbits);
        if (nb_sc <= kSelfSpine && !big_sort) {
            if (!box_applies)
                hipLaunchKernelGGL(part_apply<true>, dim3(nb_sc), dim3(kScanThreads), 0, st, (const uint8_t *)hbits, M, p.part_max, (const PartSum *)tiles, (uint32_t *)nullptr,
                                   part_start, scal, tile_first);
        } else {
            hipLaunchKernelGGL(part_spine, dim3(1), dim3(1024), 0, st, tiles, nb_sc, p.part_max);
            if (!box_applies)
                hipLaunchKernelGGL(part_apply<false>, dim3(nb_sc), dim3(kScanThreads), 0, st, (const uint8_t *)hbits, M, p.part_max, (const PartSum *)tiles, (uint32_t *)nullptr,
                                   part_start, scal, tile_first);
        }
    } else {
    hipLaunchKernelGGL(cl_keys, dim3(nb_rx), dim3(kRxHistThreads), 0, st, p, keysA, valsA, (uint2 *)recs, (uint4 *)recs, top_shift,
                       key_bits - top_shift >= 8u ? 255u : (1u << (key_bits - top_shift)) - 1u, nb_rx, hist, rx_totals ? ctx->rx_dtot : (uint32_t *)nullptr,
                       big_count);
    uint64_t *kin = nullptr, *kout = nullptr;
    uint32_t *vin = nullptr;
    if (p.idx_packed) radix_sort_pairs(keysA, keysB, nullptr, nullptr, M, key_bits, hist, spart, ctx->rx_dtot, st, &kin, nullptr, &kout, big_sort, true, top_shift);
    else radix_sort_pairs(keysA, keysB, valsA, valsB, M, key_bits, hist, spart, ctx->rx_dtot, st, &kin, &vin, &kout, big_sort, true);
    const KeyOfU64 keyof{key_mask(key_bits)};
    if (hybrid) {
        if (small_in)
            hipLaunchKernelGGL((rx_local<1024, uint64_t, KeyOfU64>), dim3((M + kLocTile - 1) / kLocTile), dim3(1024), 0, st, (const uint64_t *)kin, kout, M, top_shift, keyof, loc_cap,
                               big_list, big_count);
        else
            hipLaunchKernelGGL((rx_local<256, uint64_t, KeyOfU64>), dim3((M + kLocTile - 1) / kLocTile), dim3(256), 0, st, (const uint64_t *)kin, kout, M, top_shift, keyof, loc_cap,
                               big_list, big_count);
        hipLaunchKernelGGL((rx_big<uint64_t, KeyOfU64>), dim3(256), dim3(256), 0, st, kin, kout, M, top_shift, keyof, (const uint32_t *)big_list, (const uint32_t *)big_count);
        uint64_t *t = kin; kin = kout; kout = t;
    }
    p.sorted = vin;
    p.skeys = kin;
    // partitions: one composite scan straight off the sorted keys -> each position's partition id, the partition start
    // list and their number (scal[0]); it also zeroes the work-list counters
    {
        const LoadHead<uint64_t, KeyOfU64> heads{(const uint64_t *)kin, p.centre_bits, p.part_gap, keyof};
        hipLaunchKernelGGL((part_reduce<uint64_t, KeyOfU64>), dim3(nb_sc), dim3(kScanThreads), 0, st, heads, M, p.part_max, tiles, scal + 64, (uint32_t)(2 * kClasses * kShards), hbits);
        if (nb_sc <= kSelfSpine && !big_sort) {
            hipLaunchKernelGGL(part_apply<true>, dim3(nb_sc), dim3(kScanThreads), 0, st, (const uint8_t *)hbits, M, p.part_max, (const PartSum *)tiles, (uint32_t *)nullptr,
                               part_start, scal, tile_first);
        } else {
            hipLaunchKernelGGL(part_spine, dim3(1), dim3(1024), 0, st, tiles, nb_sc, p.part_max);
            hipLaunchKernelGGL(part_apply<false>, dim3(nb_sc), dim3(kScanThreads), 0, st, (const uint8_t *)hbits, M, p.part_max, (const PartSum *)tiles, (uint32_t *)nullptr,
                               part_start, scal, tile_first);
        }
    }
    p.srec = (uint4 *)ctx->cl_ws[14].ptr;
    }
    p.part_start = part_start; p.n_parts = scal; p.pc = pc;
    const uint32_t grid = M < 16384u ? M : 16384u;               // partitions <= marks; kernels stride over them
    uint32_t *lists = (uint32_t *)ctx->cl_ws[12].ptr;            // [kClasses][M] partitions by size class, each list in kShards pieces
    uint32_t *cnts = scal + 64;                                  // [kClasses][kShards]
    p.tps = (nb_sc + kShards - 1) / kShards;
    p.inv_norm = (float)(1.0 / pr->normalizer);
    for (int l = 0; l < 3; ++l) {
        p.t_lo[l] = (float)(pr->max_dist / (double)(1 << l) * (1.0 - 1e-5));
        p.t_hi[l] = (float)(pr->max_dist / (double)(1 << l) * (1.0 + 1e-5));
        p.t_hl[l] = float2v{p.t_hi[l], p.t_lo[l]};
    }
    p.fast = (pr->max_dist >= 1e-6 && pr->max_dist <= 1e6 && pr->normalizer >= 1e-3 && pr->normalizer <= 1e9) ? 1u : 0u;
    if (ctx->dbg & DUET_DBG_CLUSTER_EXACT) p.fast = 0;
    p.box = (ctx->dbg & DUET_DBG_CLUSTER_NOBOX) ? 0u : 1u;
    p.invn = 1.0 / pr->normalizer;
    p.scale = (double)kQOne / pr->max_dist;
    p.mergeable = pr->max_dist >= 0 ? 1u : 0u;
    p.kc = (ctx->dbg & DUET_DBG_CLUSTER_KC2) ? 2u : 64u;
    p.sym = (ctx->dbg & DUET_DBG_CLUSTER_NOSYM) ? 0u : 1u;
    p.stamps = ctx->d_stamps;
    const uint32_t gridw = std::min(32768u, std::max(1024u, M / 256u));     // (a wavefront per virtual block, striding)
    const bool small = M <= (4u << 20) && !(ctx->dbg & DUET_DBG_CLUSTER_LARGE);
    const bool cap100 = p.part_max <= 100u;          // no unit has more rows than part_max
    uint32_t *over = cnts + kClasses * kShards;                  // [kClasses][kShards] the second lists' counters
    uint32_t *l4 = lists + 4 * (size_t)M, *c4 = cnts + 4 * kShards, *o4 = over + 4 * kShards;
    // The partitions of more than 64 marks on the side stream: few, long chains.  They are listed from the partition starts
    // alone and read their rows through the sort permutation themselves (or where the record sort left them), so their launch
    // starts beside cl_box, not after it (1.0 M marks: the chain ended 30-40 us after everything else when it started behind
    // cl_box) -- except where cl_box itself numbers the partitions (large inputs: the chain is far from critical there).
    // (small inputs only: at 2e7 marks the forks' gaps are 15 of 2,100 us, and a side chain that starts 7 us earlier takes compute units
    // from the first tier's first launch -- measured: 2,187 against 2,127 us)
    if (small && ctx->cl_gates == 0) {
        const uint32_t e0 = ++ctx->cl_epoch;
        uint32_t opened = 0;
        hipLaunchKernelGGL(cl_signal, dim3(1), dim3(64), 0, ctx->cl_side[1], ctx->cl_flags + 13, e0);
        hipLaunchKernelGGL(cl_signal, dim3(1), dim3(64), 0, ctx->cl_side[1], ctx->cl_flags + 13, e0);
        hipLaunchKernelGGL(cl_gate_try, dim3(1), dim3(64), 0, ctx->cl_side[0], (const uint32_t *)(ctx->cl_flags + 14), e0, ctx->cl_flags + 15);
        hipLaunchKernelGGL(cl_signal, dim3(1), dim3(64), 0, ctx->cl_side[1], ctx->cl_flags + 14, e0);
        HIP_TRY(ctx, hipMemcpyAsync(&opened, ctx->cl_flags + 15, 4, hipMemcpyDeviceToHost, ctx->cl_side[0]));     // (not the null stream: it would wait for the caller's)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->cl_side[1]));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->cl_side[0]));
        ctx->cl_gates = opened ? 1 : -1;
    }
    const bool gate_forks = small && ctx->cl_gates > 0 && !(ctx->dbg & DUET_DBG_CLUSTER_EVENT_FORKS);
    const uint32_t epoch = ++ctx->cl_epoch;
    const bool tiers = !small || (ctx->dbg & DUET_DBG_CLUSTER_TIERS);
    // small inputs: every partition of more than 64 marks on its own workgroup of eight wavefronts (wide_unit), on the side stream beside cl_box and
    // cl_fast_all; DUET_DBG_CLUSTER_WIDE_ALL: every listed partition on four wavefronts as well (tests)
    // (up to 2 M marks: a wide unit holds 143 KB of LDS, i.e. a CU, for 50 us -- at 4 M marks the 80-odd of them take a third of the chip from
    // cl_fast_all for that long and the pipeline is 1 % slower with them, 0.423 against 0.419 ms; level at 2 M)
    const bool wide = !tiers && M <= (2u << 20) && !(ctx->dbg & DUET_DBG_CLUSTER_WIDE_OFF);
    const bool wide_all = wide && (ctx->dbg & DUET_DBG_CLUSTER_WIDE_ALL);
    // (measured, profiles/history/r06_wide_units.txt: for the listed partitions of up to 64 marks the four-wavefront units LOSE -- cl_fast_all is
    // bound by the chip's throughput there, not by a chain -- so only the partitions of more than 64 marks take them outside the tests)
    const bool wide_list = wide_all;
    auto join_signal = [&]() -> int {
        // the join the same way on small inputs: the side stream says when it is through, a gate on the main stream waits for it (the
        // side stream's launches are queued in front of that gate, so even one shared hardware queue would run them first)
        if (gate_forks) hipLaunchKernelGGL(cl_signal, dim3(1), dim3(64), 0, ctx->cl_side[0], ctx->cl_flags + 4, epoch);
        else HIP_TRY(ctx, hipEventRecord(ctx->cl_join[0], ctx->cl_side[0]));
        return DUET_OK;
    };
    auto launch_big = [&](bool next_kernel_signals) -> int {
        if (gate_forks) {
            if (!next_kernel_signals) hipLaunchKernelGGL(cl_signal, dim3(1), dim3(64), 0, st, ctx->cl_flags + 0, epoch);
            hipLaunchKernelGGL(cl_gate, dim3(1), dim3(64), 0, ctx->cl_side[0], (const uint32_t *)(ctx->cl_flags + 0), epoch);
        } else {
            HIP_TRY(ctx, hipEventRecord(ctx->cl_fork, st));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->cl_side[0], ctx->cl_fork, 0));
        }
        ClParams pb = p;
        pb.gather_rows = rec_mode ? 0u : 1u;                     // (rec_mode: the sorted rows are there already)
        if (wide) {
            uint32_t *big_parts = scal + 64 + 2 * kClasses * kShards;           // [M / 65 + 1] the partitions of more than 64 marks; o4[0] counts them
            hipLaunchKernelGGL(cl_find_big, dim3(std::max(1u, std::min(512u, (M / 8u + 255u) / 256u))), dim3(256), 0, ctx->cl_side[0], pb, big_parts, o4);
            hipLaunchKernelGGL(cl_wide_big, dim3(256), dim3(512), 0, ctx->cl_side[0], pb, (const uint32_t *)big_parts, (const uint32_t *)o4);
            return join_signal();
        }
        const uint32_t gb = std::min(4096u, std::max(256u, (M / 64u + 63u) / 64u));      // (partitions <= marks; 64 of them per wavefront and step)
        hipLaunchKernelGGL((cl_tight_big<64>), dim3(gb), dim3(64), 0, ctx->cl_side[0], pb, l4, o4, c4);
        const uint32_t gl = std::min(gb, 1024u);
        if (cap100) hipLaunchKernelGGL((cl_link_one<64, 2, 100>), dim3(gl), dim3(64), 0, ctx->cl_side[0], pb, (const uint32_t *)l4, (const uint32_t *)o4);
        else hipLaunchKernelGGL((cl_link_one<64, 2, 128>), dim3(gl), dim3(64), 0, ctx->cl_side[0], pb, (const uint32_t *)l4, (const uint32_t *)o4);
        return join_signal();
    };
    // (small inputs: cl_box itself signals as it starts -- a signal kernel of its own was 5.9 us on the main stream)
    p.fork_flag = (!box_applies && gate_forks) ? ctx->cl_flags + 0 : nullptr;
    p.fork_epoch = epoch;
    // (the side stream's launches are queued BEHIND cl_box's: were the two streams ever to share a hardware queue, a gate in front of the
    // kernel that opens it would never see it start; an event fork goes in front, as in rounds 1-5)
    if (!box_applies && !gate_forks && (rc = launch_big(false))) return rc;
    // the bounding-box test finishes the partitions it can (on SV-like data: most) and lists the others by size class
    if (box_applies)
        hipLaunchKernelGGL((cl_box<true, true>), dim3(nb_sc), dim3(kBoxThreads), 0, st, p, (const uint32_t *)nullptr, lists, cnts, (const uint8_t *)hbits, (const PartSum *)tiles, part_start, scal);
    else if (rec_mode)
        hipLaunchKernelGGL((cl_box<true, false>), dim3(nb_sc), dim3(kBoxThreads), 0, st, p, (const uint32_t *)tile_first, lists, cnts, (const uint8_t *)nullptr, (const PartSum *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
    else
        hipLaunchKernelGGL((cl_box<false, false>), dim3(nb_sc), dim3(kBoxThreads), 0, st, p, (const uint32_t *)tile_first, lists, cnts, (const uint8_t *)nullptr, (const PartSum *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
    if ((box_applies || gate_forks) && (rc = launch_big(!box_applies && gate_forks))) return rc;
    if (!tiers) {
        // one launch for the classes of up to 64 marks (32 beside the wide units), nothing handed on
        if (wide_list && !gate_forks) {
            HIP_TRY(ctx, hipEventRecord(ctx->cl_join[1], st));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->cl_side[1], ctx->cl_join[1], 0));
        }
        p.fast_classes = wide_all ? 0u : 15u;
        p.box_done_flag = (wide_list && gate_forks) ? ctx->cl_flags + 8 : nullptr;
        if (wide_list) hipLaunchKernelGGL(cl_fast_all<false>, dim3(gridw), dim3(64), 0, st, p, (const uint32_t *)lists, (const uint32_t *)cnts);
        else hipLaunchKernelGGL(cl_fast_all<true>, dim3(gridw), dim3(64), 0, st, p, (const uint32_t *)lists, (const uint32_t *)cnts);
        if (wide_list) {
            // the listed partitions of 33..64 marks on a side stream of their own: behind cl_box (cl_fast_all's first workgroup says when that
            // is), beside cl_fast_all and the partitions of more than 64
            if (gate_forks) hipLaunchKernelGGL(cl_gate, dim3(1), dim3(64), 0, ctx->cl_side[1], (const uint32_t *)(ctx->cl_flags + 8), epoch);
            const uint32_t gwl = std::min(8192u, std::max(256u, M / 512u));
            hipLaunchKernelGGL(cl_wide_list, dim3(gwl), dim3(256), 0, ctx->cl_side[1], p, (const uint32_t *)lists, (const uint32_t *)cnts, 0u);
            if (gate_forks) hipLaunchKernelGGL(cl_signal, dim3(1), dim3(64), 0, ctx->cl_side[1], ctx->cl_flags + 12, epoch);
            else HIP_TRY(ctx, hipEventRecord(ctx->cl_join[2], ctx->cl_side[1]));
        }
    } else if (small) {
        hipLaunchKernelGGL(cl_tight_all, dim3(gridw), dim3(64), 0, st, p, lists, (const uint32_t *)cnts, over);
    } else {
        // the first tier, one launch per size class: the registers and the LDS each variant needs, not the largest one's.
        // (Measured and dropped, profiles/history: the tiles in two or four pieces, a piece's box test on a side stream while the
        // first tier takes the piece before -- the box test, 32 KB of LDS per workgroup, then waits for room between the first
        // tier's many small workgroups and takes as long for a piece as it does alone for everything.)
        hipLaunchKernelGGL((cl_tight_one<64, 1, kK64>), dim3(grid), dim3(64), 0, st, p, lists + 3 * (size_t)M, (const uint32_t *)(cnts + 3 * kShards), over + 3 * kShards);
        // (what this class hands on -- the most of all classes -- has its second tier beside the other classes' first one)
        if (gate_forks) {
            hipLaunchKernelGGL(cl_signal, dim3(1), dim3(64), 0, st, ctx->cl_flags + 8, epoch);
            hipLaunchKernelGGL(cl_gate, dim3(1), dim3(64), 0, ctx->cl_side[1], (const uint32_t *)(ctx->cl_flags + 8), epoch);
        } else {
            HIP_TRY(ctx, hipEventRecord(ctx->cl_join[1], st));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->cl_side[1], ctx->cl_join[1], 0));
        }
        hipLaunchKernelGGL((cl_tier2_one<64, 1, 64>), dim3(std::min(grid, 4096u)), dim3(64), 0, ctx->cl_side[1], p, (const uint32_t *)(lists + 3 * (size_t)M), (const uint32_t *)(over + 3 * kShards));
        HIP_TRY(ctx, hipEventRecord(ctx->cl_join[2], ctx->cl_side[1]));
        hipLaunchKernelGGL((cl_tight_one<32, 1, kK32>), dim3(grid), dim3(64), 0, st, p, lists + 2 * (size_t)M, (const uint32_t *)(cnts + 2 * kShards), over + 2 * kShards);
        hipLaunchKernelGGL((cl_tight_one<16, 1, kK16>), dim3(grid), dim3(64), 0, st, p, lists + 1 * (size_t)M, (const uint32_t *)(cnts + 1 * kShards), over + 1 * kShards);
        hipLaunchKernelGGL((cl_tight_one<8, 1, kK8>), dim3(grid), dim3(64), 0, st, p, lists, (const uint32_t *)cnts, over);
    }
    // what they handed on
    if (tiers) hipLaunchKernelGGL(cl_tier2_all, dim3(std::min(gridw, 2048u)), dim3(64), 0, st, p, (const uint32_t *)lists, (const uint32_t *)over, small ? 1u : 0u);
    if (!small || (wide_list && !gate_forks)) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->cl_join[2], 0));
    // (the join inside cl_pc_sums: its waiting workgroups hold wave slots -- up to M / 2048 workgroups of four wavefronts when every mark is a partition of its
    // own -- so only up to 2 M marks, where they cannot take more than half of the chip's from the side stream's kernels; beyond, a one-lane gate kernel)
    const bool join_in_sums = gate_forks && !wide_list && M <= (2u << 20);
    if (gate_forks && wide_list) hipLaunchKernelGGL(cl_gate2, dim3(1), dim3(64), 0, st, (const uint32_t *)(ctx->cl_flags + 4), (const uint32_t *)(ctx->cl_flags + 12), epoch);
    else if (gate_forks && !join_in_sums) hipLaunchKernelGGL(cl_gate, dim3(1), dim3(64), 0, st, (const uint32_t *)(ctx->cl_flags + 4), epoch);
    else if (!gate_forks) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->cl_join[0], 0));
    // clusters per partition -> their sums per 64 and per 2048 partitions (cl_emit numbers the candidates from them); with gate forks the launch is
    // also the join of the side stream
    hipLaunchKernelGGL(cl_pc_sums, dim3((M + 2047u) / 2048u), dim3(256), 0, st, (const uint32_t *)pc, (const uint32_t *)scal, cbase, spart,
                       join_in_sums ? (const uint32_t *)(ctx->cl_flags + 4) : (const uint32_t *)nullptr, epoch);
    p.csum = cbase;
    p.tsum = spart;
    p.n_cands_w = res->n_cands;
    p.n_cands = res->n_cands;
    hipLaunchKernelGGL(cl_emit, dim3(std::min((M + 16383u) / 16384u * 8u, 4096u)), b256, 0, st, p);        // (a wave per 64 partitions, striding)
    HIP_TRY(ctx, hipGetLastError());
    if (getenv("DUET_CL_DEBUG")) {
        uint32_t h[64 + 2 * kClasses * kShards];
        HIP_TRY(ctx, hipMemcpyAsync(h, scal, sizeof(h), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        uint32_t c[2 * kClasses] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < 2 * kClasses; ++k)
            for (int sh = 0; sh < kShards; ++sh) c[k] += h[64 + k * kShards + sh];
        fprintf(stderr, "[duet_cluster] parts %u; past the box test, by size class: %u %u %u %u %u; handed on by the first tier: %u %u %u %u %u\n",
                h[0], c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9]);
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
    // (the copy is ordered on the stream it was issued on: a run on ANOTHER stream uploads again -- after waiting for that
    // stream, whose pageable copy may still be reading the host vector)
    if (ctx->sv_depth_off_at != (void *)d_depth_off || ctx->sv_depth_off.size() != (size_t)K + 1 || ctx->sv_depth_off_stream != st ||
        memcmp(ctx->sv_depth_off.data(), pr->depth_off, ((size_t)K + 1) * 4) != 0) {
        // (the previous copy's source is about to change: wait for THAT COPY -- an event recorded behind it, not the stream it
        // was issued on, which the caller may have destroyed in the meantime)
        if (!ctx->sv_depth_off_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->sv_depth_off_ev, hipEventDisableTiming));
        if (ctx->sv_depth_off_at) HIP_TRY(ctx, hipEventSynchronize(ctx->sv_depth_off_ev));
        ctx->sv_depth_off.assign(pr->depth_off, pr->depth_off + K + 1);
        HIP_TRY(ctx, hipMemcpyAsync(d_depth_off, ctx->sv_depth_off.data(), ((size_t)K + 1) * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipEventRecord(ctx->sv_depth_off_ev, st));
        ctx->sv_depth_off_at = (void *)d_depth_off;
        ctx->sv_depth_off_stream = st;
    }
    // clustering; its emit kernel also writes what a caller VCF would have carried (support, reference reads, GT)
    // and the marks' read indices in output order
    SvExtra sv;
    sv.mark_in = pr->mark_read; sv.depth = pr->depth; sv.depth_off = d_depth_off; sv.depth_bin = pr->depth_bin;
    sv.mark_out = (uint32_t *)ctx->sv_ws[4].ptr; sv.svread = (uint32_t *)ctx->sv_ws[1].ptr;
    sv.refread = (uint32_t *)ctx->sv_ws[2].ptr; sv.gt = (uint8_t *)ctx->sv_ws[3].ptr;
    sv.ef_ctg_off = nullptr; sv.ef_zero = nullptr; sv.n_contigs = K;
    // (fully asynchronous runs: cl_emit writes step E/F's plan into E/F's workspace, sized for the bound of M candidates)
    if (!n_cands_host && (rc = duet_ef_plan_on_device_prepare(ctx, K, M, st, &sv.ef_ctg_off, &sv.ef_zero))) return rc;
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
                                             out_pred, out_ps, st, true);
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

int duet_svim_phase_host(duet_ctx *ctx, const duet_svim_problem *pr, const duet_cluster_result *res, uint8_t *out_pred, uint32_t *out_ps)
{
    if (!ctx) return duet_fail(nullptr, DUET_ERR_INVALID, "null context");
    if (!pr || !res || !res->n_cands || !out_pred || !out_ps) return duet_fail(ctx, DUET_ERR_INVALID, "null argument");
    if (!pr->depth_off || pr->n_contigs == 0) return duet_fail(ctx, DUET_ERR_INVALID, "bad depth / contig description");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t M = pr->marks.n_marks;
    *res->n_cands = 0;
    if (M == 0) return DUET_OK;
    // every array is on the host here: what the device entry has to trust is checked -- a contig id beyond the depth description
    // would index sv_depth_off / the E/F plan outside their K + 1 entries
    if (!res->cand_off || !res->cand_contig || !res->cand_type || !res->cand_pos || !res->cand_span)
        return duet_fail(ctx, DUET_ERR_INVALID, "null result array");
    for (uint32_t k = 0; k < pr->n_contigs; ++k)
        if (pr->depth_off[k] > pr->depth_off[k + 1]) return duet_fail(ctx, DUET_ERR_INVALID, "depth_off must be non-decreasing");
    if (!pr->marks.mark_contig) return duet_fail(ctx, DUET_ERR_INVALID, "null array");
    for (uint32_t i = 0; i < M; ++i)
        if (pr->marks.mark_contig[i] >= pr->n_contigs)
            return duet_fail(ctx, DUET_ERR_INVALID, "a mark's contig id is not below n_contigs (the depth description's contig count)");
    hipStream_t s = ctx->own_stream;
    int rc;
    const size_t n_depth = pr->depth_off[pr->n_contigs];
    const void *src[7] = {pr->marks.mark_contig, pr->marks.mark_type, pr->marks.mark_pos, pr->marks.mark_span, pr->mark_read, pr->read_tag, pr->depth};
    const size_t ib[7] = {(size_t)M * 2, (size_t)M, (size_t)M * 4, (size_t)M * 4, (size_t)M * 4, (size_t)pr->n_reads * 8, n_depth * 4};
    DevBuf *in[7] = {&ctx->cl_in[0], &ctx->cl_in[1], &ctx->cl_in[2], &ctx->cl_in[3], &ctx->sv_in[0], &ctx->sv_in[1], &ctx->sv_in[2]};
    for (int i = 0; i < 7; ++i) {
        if (!src[i] && ib[i]) return duet_fail(ctx, DUET_ERR_INVALID, "null array");
        if ((rc = duet_reserve(ctx, *in[i], ib[i] + 16))) return rc;           // (+16: a readable word even for an empty table)
        if (ib[i]) HIP_TRY(ctx, hipMemcpyAsync(in[i]->ptr, src[i], ib[i], hipMemcpyHostToDevice, s));
    }
    const size_t ob[6] = {(size_t)M * 4, ((size_t)M + 1) * 4 + 16, (size_t)M * 2, (size_t)M, (size_t)M * 4, (size_t)M * 4};
    for (int i = 0; i < 6; ++i)
        if ((rc = duet_reserve(ctx, ctx->cl_out[i], ob[i]))) return rc;
    if ((rc = duet_reserve(ctx, ctx->sv_out[0], (size_t)M + 16))) return rc;
    if ((rc = duet_reserve(ctx, ctx->sv_out[1], (size_t)M * 4 + 16))) return rc;
    duet_svim_problem d = *pr;
    d.marks.mark_contig = (const uint16_t *)ctx->cl_in[0].ptr;
    d.marks.mark_type = (const uint8_t *)ctx->cl_in[1].ptr;
    d.marks.mark_pos = (const uint32_t *)ctx->cl_in[2].ptr;
    d.marks.mark_span = (const uint32_t *)ctx->cl_in[3].ptr;
    d.mark_read = (const uint32_t *)ctx->sv_in[0].ptr;
    d.read_tag = (const uint64_t *)ctx->sv_in[1].ptr;
    d.depth = (const uint32_t *)ctx->sv_in[2].ptr;
    duet_cluster_result r;
    r.order = (uint32_t *)ctx->cl_out[0].ptr;
    r.cand_off = (uint32_t *)ctx->cl_out[1].ptr;
    r.cand_contig = (uint16_t *)ctx->cl_out[2].ptr;
    r.cand_type = (uint8_t *)ctx->cl_out[3].ptr;
    r.cand_pos = (uint32_t *)ctx->cl_out[4].ptr;
    r.cand_span = (uint32_t *)ctx->cl_out[5].ptr;
    r.n_cands = (uint32_t *)((char *)ctx->cl_out[1].ptr + ((size_t)M + 1) * 4);        // spare word after cand_off
    uint32_t n = 0;
    if ((rc = duet_svim_phase_device(ctx, &d, &r, (uint8_t *)ctx->sv_out[0].ptr, (uint32_t *)ctx->sv_out[1].ptr, &n, s))) return rc;
    if ((rc = duet_ef_check(ctx, s))) return rc;                                        // (synchronises; DUET_ERR_DIV_ZERO comes out here)
    *res->n_cands = n;
    if (res->order) HIP_TRY(ctx, hipMemcpy(res->order, r.order, (size_t)M * 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(res->cand_off, r.cand_off, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(res->cand_contig, r.cand_contig, (size_t)n * 2, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(res->cand_type, r.cand_type, (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(res->cand_pos, r.cand_pos, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(res->cand_span, r.cand_span, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out_pred, ctx->sv_out[0].ptr, (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out_ps, ctx->sv_out[1].ptr, (size_t)n * 4, hipMemcpyDeviceToHost));
    return DUET_OK;
}


}  // extern "C"
