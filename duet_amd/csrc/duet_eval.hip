// duet_eval.hip -- gfx950 kernels and C ABI for the accuracy evaluator: a phased callset
// scored against a truth set, as src/scripts/evaluation.py:99-159 does it.
//
//   every call is matched to the NEAREST truth call of its contig and type (np.searchsorted; ties and the end-of-list
//   rule of :117-125), accepted within `refdist` and a length ratio of at least `ratio` (:126-127);
//   call_tp / base_tp / call_tp_gt / base_tp_gt are SETS of record ids (:128-133);
//   phasing is scored per (contig, phase set): the calls whose haplotype equals the truth's ("same") against those that
//   are its mirror image ("flip"), the labelling with more ids -- call ids plus truth ids -- wins, ties to "flip"
//   (:134-148); the winners are united over the phase sets.
//
// The host (duet_amd/evaluation.py) parses the two VCFs as upstream does and flattens them: truth calls per (contig, type)
// in position order, record ids and haplotype strings as small integers.  Sets become flag arrays indexed by id (a store of
// 1 is idempotent), the per-phase-set set sizes come from a hash set of (phase set, labelling, side, id) keys built with
// 64-bit compare-and-swap -- its CONTENT does not depend on the order of arrival.  The six counts come back; the ten numbers
// upstream prints are divisions of those counts in binary64 on the host, exactly as upstream divides them.
//
// Integer / index work, no floating point besides the one length-ratio division per call (binary64, as upstream).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <string>

#include "duet_ef.h"
#include "duet_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;

struct EvalParams {
    uint32_t n_calls, refdist;
    double ratio;
    const uint32_t *base_off, *base_pos, *base_len, *base_uid;
    const uint8_t *base_hp;
    const uint32_t *call_key, *call_pos, *call_len, *call_uid, *call_group;
    const uint8_t *call_hp;
    // per call: matched truth record (global index) or kNone; bit 0 tp (always set with a match), 1 gt, 2 same, 3 flip
    uint32_t *match;
    uint8_t *bits;
    uint8_t *f_call_tp, *f_base_tp, *f_call_gt, *f_base_gt, *f_call_hp, *f_base_hp;      // flag arrays, indexed by id
    unsigned long long *table;                    // hash set, `mask` + 1 slots, 0 = empty
    uint32_t mask;
    uint32_t *cnt_same, *cnt_flip;                // [n_groups]: ids (call side + truth side) per labelling
};

// haplotype codes fixed by the host: 0 '1|0', 1 '0|1', 2 '1|1', anything else >= 3 (compared for equality only)
__global__ __launch_bounds__(256) void eval_match(const EvalParams p)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.n_calls) return;
    const uint32_t key = p.call_key[c];
    uint32_t m = kNone;
    uint8_t bits = 0;
    if (key != kNone) {
        const uint32_t lo0 = p.base_off[key], n = p.base_off[key + 1] - lo0;
        const uint32_t *bp = p.base_pos + lo0;
        const uint32_t pos = p.call_pos[c];
        uint32_t lo = 0, hi = n;                                        // np.searchsorted(..., side='left')
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (bp[mid] < pos) lo = mid + 1; else hi = mid;
        }
        uint32_t j;                                                      // :118-125
        if (lo == n) j = lo - 1;
        else if (lo > 0 && llabs((long long)pos - (long long)bp[lo]) > llabs((long long)pos - (long long)bp[lo - 1])) j = lo - 1;
        else j = lo;
        const uint32_t b = lo0 + j;
        const uint32_t cl = p.call_len[c], bl = p.base_len[b];
        const uint32_t mn = cl < bl ? cl : bl, mx = cl < bl ? bl : cl;
        const bool ok = (uint64_t)llabs((long long)pos - (long long)bp[j]) <= (uint64_t)p.refdist &&
                        (double)mn / (double)mx >= p.ratio;             // :126-127
        if (ok) {
            m = b;
            const uint8_t ch = p.call_hp[c], bh = p.base_hp[b];
            const bool gt = (ch < 2 && bh < 2) || (ch == 2 && bh == 2);                                  // :130-133
            const bool same = ch == bh;                                                                 // :134-136
            const bool flip = (ch == 2 && bh == 2) || (ch == 1 && bh == 0) || (ch == 0 && bh == 1);      // :137-141
            bits = (uint8_t)(1u | (gt ? 2u : 0u) | (same ? 4u : 0u) | (flip ? 8u : 0u));
            p.f_call_tp[p.call_uid[c]] = 1;
            p.f_base_tp[p.base_uid[b]] = 1;
            if (gt) { p.f_call_gt[p.call_uid[c]] = 1; p.f_base_gt[p.base_uid[b]] = 1; }
        }
    }
    p.match[c] = m;
    p.bits[c] = bits;
}

// true when `key` was not in the set yet
__device__ __forceinline__ bool set_insert(unsigned long long *table, uint32_t mask, unsigned long long key)
{
    const unsigned long long stored = key + 1ull;                       // 0 marks an empty slot
    unsigned long long h = key * 0x9E3779B97F4A7C15ull;
    uint32_t slot = (uint32_t)(h >> 32) & mask;
    for (;;) {
        const unsigned long long old = atomicCAS(&table[slot], 0ull, stored);
        if (old == 0ull) return true;
        if (old == stored) return false;
        slot = (slot + 1u) & mask;
    }
}

// key = labelling (1 bit) | phase-set group (30 bits) | side (1 bit: 0 call, 1 truth) | id (32 bits)
__device__ __forceinline__ unsigned long long set_key(uint32_t labelling, uint32_t group, uint32_t side, uint32_t id)
{
    return ((unsigned long long)labelling << 63) | ((unsigned long long)group << 33) | ((unsigned long long)side << 32) | id;
}

__global__ __launch_bounds__(256) void eval_group_sets(const EvalParams p)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.n_calls) return;
    const uint8_t bits = p.bits[c];
    if (!(bits & 12u)) return;
    const uint32_t g = p.call_group[c], cu = p.call_uid[c], bu = p.base_uid[p.match[c]];
    if (bits & 4u) {
        uint32_t add = set_insert(p.table, p.mask, set_key(0, g, 0, cu)) ? 1u : 0u;
        add += set_insert(p.table, p.mask, set_key(0, g, 1, bu)) ? 1u : 0u;
        if (add) atomicAdd(&p.cnt_same[g], add);
    }
    if (bits & 8u) {
        uint32_t add = set_insert(p.table, p.mask, set_key(1, g, 0, cu)) ? 1u : 0u;
        add += set_insert(p.table, p.mask, set_key(1, g, 1, bu)) ? 1u : 0u;
        if (add) atomicAdd(&p.cnt_flip[g], add);
    }
}

__global__ __launch_bounds__(256) void eval_group_choose(const EvalParams p)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.n_calls) return;
    const uint8_t bits = p.bits[c];
    if (!(bits & 12u)) return;
    const uint32_t g = p.call_group[c];
    const bool use_same = p.cnt_same[g] > p.cnt_flip[g];                // :143-148, ties to the mirrored labelling
    if (bits & (use_same ? 4u : 8u)) {
        p.f_call_hp[p.call_uid[c]] = 1;
        p.f_base_hp[p.base_uid[p.match[c]]] = 1;
    }
}

// out[which] += number of set flags
__global__ __launch_bounds__(256) void eval_count(const uint8_t *flags, uint32_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += flags[i] ? 1u : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63u) == 0 && acc) atomicAdd(out, acc);
}

}  // namespace

extern "C" {

int duet_eval_run_host(duet_ctx *ctx, const duet_eval_problem *pr, duet_eval_counts *out)
{
    if (!ctx) return duet_fail(nullptr, DUET_ERR_INVALID, "null context");
    if (!pr || !out) return duet_fail(ctx, DUET_ERR_INVALID, "null argument");
    memset(out, 0, sizeof(*out));
    const uint32_t nb = pr->n_base, nc = pr->n_calls, ng = pr->n_groups, nk = pr->n_keys;
    if (nc == 0) return DUET_OK;
    if (ng >= (1u << 30)) return duet_fail(ctx, DUET_ERR_INVALID, "too many phase sets");
    if (!pr->base_off || !pr->call_key || !pr->call_pos || !pr->call_len || !pr->call_hp || !pr->call_uid || !pr->call_group ||
        (nb && (!pr->base_pos || !pr->base_len || !pr->base_hp || !pr->base_uid)))
        return duet_fail(ctx, DUET_ERR_INVALID, "null array");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->own_stream;
    // one arena: inputs, then the scratch that has to start at zero
    uint32_t slots = 64;
    while (slots < 8u * nc && slots < (1u << 31)) slots <<= 1;
    const size_t in_bytes[11] = {((size_t)nk + 1) * 4, (size_t)nb * 4, (size_t)nb * 4, (size_t)nb * 4, (size_t)nb,
                                 (size_t)nc * 4, (size_t)nc * 4, (size_t)nc * 4, (size_t)nc * 4, (size_t)nc * 4, (size_t)nc};
    const void *in_src[11] = {pr->base_off, pr->base_pos, pr->base_len, pr->base_uid, pr->base_hp, pr->call_key, pr->call_pos,
                              pr->call_len, pr->call_uid, pr->call_group, pr->call_hp};
    size_t off[11], total = 0;
    for (int i = 0; i < 11; ++i) { off[i] = total; total += (in_bytes[i] + 255) & ~(size_t)255; }
    const size_t o_match = total;  total += ((size_t)nc * 4 + 255) & ~(size_t)255;
    const size_t o_bits = total;   total += ((size_t)nc + 255) & ~(size_t)255;
    const size_t zero_from = total;
    const uint32_t ncu = pr->n_call_uid, nbu = pr->n_base_uid;
    const size_t o_flags[6] = {total, total + ncu, total + 2 * (size_t)ncu, total + 3 * (size_t)ncu, total + 3 * (size_t)ncu + nbu,
                               total + 3 * (size_t)ncu + 2 * (size_t)nbu};      // call tp/gt/hp, truth tp/gt/hp
    total += (3 * ((size_t)ncu + nbu) + 255) & ~(size_t)255;
    const size_t o_cnt = total;    total += ((size_t)ng * 8 + 255) & ~(size_t)255;
    const size_t o_out = total;    total += 256;
    const size_t o_table = total;  total += (size_t)slots * 8;
    int rc;
    if ((rc = duet_reserve(ctx, ctx->eval_ws, total))) return rc;
    char *base = (char *)ctx->eval_ws.ptr;
    for (int i = 0; i < 11; ++i)
        if (in_bytes[i]) HIP_TRY(ctx, hipMemcpyAsync(base + off[i], in_src[i], in_bytes[i], hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(base + zero_from, 0, total - zero_from, st));
    EvalParams p;
    p.n_calls = nc; p.refdist = pr->refdist; p.ratio = pr->ratio;
    p.base_off = (const uint32_t *)(base + off[0]); p.base_pos = (const uint32_t *)(base + off[1]);
    p.base_len = (const uint32_t *)(base + off[2]); p.base_uid = (const uint32_t *)(base + off[3]);
    p.base_hp = (const uint8_t *)(base + off[4]); p.call_key = (const uint32_t *)(base + off[5]);
    p.call_pos = (const uint32_t *)(base + off[6]); p.call_len = (const uint32_t *)(base + off[7]);
    p.call_uid = (const uint32_t *)(base + off[8]); p.call_group = (const uint32_t *)(base + off[9]);
    p.call_hp = (const uint8_t *)(base + off[10]);
    p.match = (uint32_t *)(base + o_match); p.bits = (uint8_t *)(base + o_bits);
    p.f_call_tp = (uint8_t *)(base + o_flags[0]); p.f_call_gt = (uint8_t *)(base + o_flags[1]); p.f_call_hp = (uint8_t *)(base + o_flags[2]);
    p.f_base_tp = (uint8_t *)(base + o_flags[3]); p.f_base_gt = (uint8_t *)(base + o_flags[4]); p.f_base_hp = (uint8_t *)(base + o_flags[5]);
    p.cnt_same = (uint32_t *)(base + o_cnt); p.cnt_flip = p.cnt_same + ng;
    p.table = (unsigned long long *)(base + o_table); p.mask = slots - 1u;
    const dim3 grid((nc + 255) / 256), block(256);
    hipLaunchKernelGGL(eval_match, grid, block, 0, st, p);
    hipLaunchKernelGGL(eval_group_sets, grid, block, 0, st, p);
    hipLaunchKernelGGL(eval_group_choose, grid, block, 0, st, p);
    uint32_t *d_out = (uint32_t *)(base + o_out);
    const uint8_t *fl[6] = {p.f_call_tp, p.f_base_tp, p.f_call_gt, p.f_base_gt, p.f_call_hp, p.f_base_hp};
    const uint32_t fn[6] = {ncu, nbu, ncu, nbu, ncu, nbu};
    for (int i = 0; i < 6; ++i)
        if (fn[i]) hipLaunchKernelGGL(eval_count, dim3((fn[i] + 2047) / 2048 < 1024u ? (fn[i] + 2047) / 2048 : 1024u), block, 0, st, fl[i], fn[i], d_out + i);
    HIP_TRY(ctx, hipGetLastError());
    uint32_t h[6];
    HIP_TRY(ctx, hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    out->call_tp = h[0]; out->base_tp = h[1]; out->call_gt = h[2]; out->base_gt = h[3]; out->call_hp = h[4]; out->base_hp = h[5];
    return DUET_OK;
}

}  // extern "C"
