// duet_rows.hip -- gfx950 kernels and C ABI for the last step of Duet's step E/F on the device: from the
// per-candidate (pred, ps) to the data rows of phased_sv.vcf, as text.
//
// What it restates (reference file:line):
//   emission order       src/duet/sv_phasing_fn.py:204-228   contig order, PS-class 0/1/2, file order; pred 0 dropped
//   the sort             src/duet/sv_phasing_fn.py:229        stable, key (CHROM as text, POS as int)
//   the rows             src/duet/write_file.py:6-17          CHROM POS Duet.<i> REF ALT . PASS SVLEN=<signed>;SVTYPE=<T> HP:PS <hp>:<ps>
//   the sign of SVLEN    src/duet/sv_phasing_fn.py:225        positive iff SVTYPE is exactly INS or DUP
//
// Pipeline (one stream):
//   scan + compaction     kept = pred != 0; the scan's store writes the kept candidates' indices
//   rows_keys             per kept candidate: PS-class (distinct PS over its tagged marks, capped at 2), contig (binary
//                         search in the contig offsets) -> sort key  rank(CHROM text) | POS | contig | class
//   radix sort            stable LSD over exactly the bits in use; ties keep file order, so the result is the reference's
//   scan of row lengths   decimal digit counts + text lengths -> row offsets, total
//   rows_write            one wavefront per row: lane 0 formats the numeric pieces into LDS, all lanes copy the pieces
//
// The texts (CHROM, REF, ALT, SVTYPE of every candidate) come in one pool with 4 offsets per candidate; CHROM's byte-order
// rank among the distinct CHROM texts is computed by the host once per input (there are a few dozen of them).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>

#include "duet_ef.h"
#include "duet_internal.h"

namespace {

#include "duet_prims.hip.h"

constexpr uint32_t kAbsentRead = 0xFFFFFFFFu;

struct RowParams {
    uint32_t C, K, n_rows;
    const uint8_t *pred;
    const uint32_t *ps;
    const uint32_t *cand_pos, *cand_svlen;
    const uint8_t *cand_plus;
    const uint16_t *chrom_rank;
    const char *pool;
    const uint32_t *str_off;                 // [4 * C + 1]
    const uint32_t *ctg_off;                 // device copy, [K + 1]
    const uint32_t *cand_off, *mark_read;    // the E/F problem's CSR
    const uint64_t *read_tag;
    uint32_t sh_rank, sh_pos;                // key = rank << sh_rank | pos << sh_pos | contig << 2 | class
    const uint32_t *sorted;                  // candidate of each row
    const uint32_t *row_off;                 // [n_rows] byte offset of each row
    char *out;
    uint64_t cap;
    uint32_t *overflow;
};

struct LoadKeep {
    const uint8_t *pred;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return pred[i] != 0 ? 1u : 0u; }
};
struct StoreCompact {
    uint32_t *idx;
    __device__ __forceinline__ void operator()(uint32_t i, uint32_t v, uint32_t in) const { if (in) idx[v] = i; }
};

__device__ __forceinline__ uint32_t digits_u32(uint32_t v)
{
    uint32_t d = 1;
    d += v >= 10u; d += v >= 100u; d += v >= 1000u; d += v >= 10000u; d += v >= 100000u;
    d += v >= 1000000u; d += v >= 10000000u; d += v >= 100000000u; d += v >= 1000000000u;
    return d;
}

__global__ void rows_keys(const RowParams p, const uint32_t *idx, uint64_t *keys, uint32_t *vals)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= p.n_rows) return;
    const uint32_t c = idx[j];
    // PS-class (sv_phasing_fn.py:191-194): distinct PS over ALL tagged marks, 0 / 1 / more
    uint32_t n_ps = 0, first = 0;
    for (uint32_t m = p.cand_off[c]; m < p.cand_off[c + 1]; ++m) {
        const uint32_t r = p.mark_read[m];
        if (r == kAbsentRead) continue;
        const uint32_t ps = (uint32_t)p.read_tag[r];
        if (n_ps == 0) { n_ps = 1; first = ps; }
        else if (ps != first) { n_ps = 2; break; }
    }
    // contig: the last k with ctg_off[k] <= c
    uint32_t lo = 0, hi = p.K;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (p.ctg_off[mid] <= c) lo = mid; else hi = mid;
    }
    keys[j] = ((uint64_t)p.chrom_rank[c] << p.sh_rank) | ((uint64_t)p.cand_pos[c] << p.sh_pos) | ((uint64_t)lo << 2) | n_ps;
    vals[j] = c;
}

// fixed pieces of a row
__device__ __constant__ const char kPieceB[] = "\t.\tPASS\tSVLEN=";       // 14
__device__ __constant__ const char kPieceB2[] = ";SVTYPE=<";               // 9
__device__ __constant__ const char kPieceC[] = ">\tHP:PS\t";              // 8

struct LoadRowLen {
    RowParams p;
    __device__ __forceinline__ uint32_t operator()(uint32_t j) const
    {
        const uint32_t c = p.sorted[j];
        const uint32_t *o = p.str_off + 4 * (size_t)c;
        const uint32_t text = o[4] - o[0];                                  // CHROM + REF + ALT + SVTYPE
        const uint32_t mag = p.cand_svlen[c];
        const uint32_t neg = (!p.cand_plus[c] && mag != 0) ? 1u : 0u;
        //      \t pos \t Duet. row \t | \t | piece B [-] mag piece B2 | piece C hp : ps \n
        return text + 1 + digits_u32(p.cand_pos[c]) + 1 + 5 + digits_u32(j + 1) + 1 + 1 + 14 + neg + digits_u32(mag) + 9 + 8 + 3 + 1 +
               digits_u32(p.ps[c]) + 1;
    }
};

__device__ __forceinline__ uint32_t put_u32(char *dst, uint32_t v)
{
    const uint32_t n = digits_u32(v);
    for (uint32_t i = n; i-- > 0;) {
        dst[i] = (char)('0' + v % 10u);
        v /= 10u;
    }
    return n;
}

__global__ __launch_bounds__(256) void rows_write(const RowParams p)
{
    __shared__ char s_a[4][32], s_b[4][40], s_c[4][32];
    __shared__ uint32_t s_len[4][3];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t j = blockIdx.x * 4 + wave; j < p.n_rows; j += gridDim.x * 4) {
        const uint32_t c = p.sorted[j];
        const uint32_t *o = p.str_off + 4 * (size_t)c;
        const uint32_t o0 = o[0], o1 = o[1], o2 = o[2], o3 = o[3], o4 = o[4];
        if (lane == 0) {
            char *a = s_a[wave];
            uint32_t n = 0;
            a[n++] = '\t';
            n += put_u32(a + n, p.cand_pos[c]);
            a[n++] = '\t'; a[n++] = 'D'; a[n++] = 'u'; a[n++] = 'e'; a[n++] = 't'; a[n++] = '.';
            n += put_u32(a + n, j + 1);
            a[n++] = '\t';
            s_len[wave][0] = n;
            char *b = s_b[wave];
            n = 0;
            for (int i = 0; i < 14; ++i) b[n++] = kPieceB[i];
            const uint32_t mag = p.cand_svlen[c];
            if (!p.cand_plus[c] && mag != 0) b[n++] = '-';
            n += put_u32(b + n, mag);
            for (int i = 0; i < 9; ++i) b[n++] = kPieceB2[i];
            s_len[wave][1] = n;
            char *q = s_c[wave];
            n = 0;
            for (int i = 0; i < 8; ++i) q[n++] = kPieceC[i];
            const uint32_t hp = p.pred[c] & 3u;                              // 1: 1|0, 2: 0|1, 3: 1|1
            q[n++] = hp == 2 ? '0' : '1';
            q[n++] = '|';
            q[n++] = hp == 1 ? '0' : '1';
            q[n++] = ':';
            n += put_u32(q + n, p.ps[c]);
            q[n++] = '\n';
            s_len[wave][2] = n;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t la = s_len[wave][0], lb = s_len[wave][1], lc = s_len[wave][2];
        uint64_t cur = p.row_off[j];
        const uint64_t total = (uint64_t)(o4 - o0) + la + 1 + lb + lc;
        if (cur + total > p.cap) {
            if (lane == 0) *p.overflow = 1;
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        char *out = p.out;
        for (uint32_t i = lane; i < o1 - o0; i += 64) out[cur + i] = p.pool[o0 + i];          // CHROM
        cur += o1 - o0;
        for (uint32_t i = lane; i < la; i += 64) out[cur + i] = s_a[wave][i];                 // \t POS \t Duet.N \t
        cur += la;
        for (uint32_t i = lane; i < o2 - o1; i += 64) out[cur + i] = p.pool[o1 + i];          // REF
        cur += o2 - o1;
        if (lane == 0) out[cur] = '\t';
        cur += 1;
        for (uint32_t i = lane; i < o3 - o2; i += 64) out[cur + i] = p.pool[o2 + i];          // ALT
        cur += o3 - o2;
        for (uint32_t i = lane; i < lb; i += 64) out[cur + i] = s_b[wave][i];                 // \t.\tPASS\tSVLEN=..;SVTYPE=<
        cur += lb;
        for (uint32_t i = lane; i < o4 - o3; i += 64) out[cur + i] = p.pool[o3 + i];          // SVTYPE
        cur += o4 - o3;
        for (uint32_t i = lane; i < lc; i += 64) out[cur + i] = s_c[wave][i];                 // >\tHP:PS\t hp:ps \n
        __builtin_amdgcn_wave_barrier();                                                     // before lane 0 rewrites the LDS pieces
    }
}

}  // namespace

extern "C" {

int duet_rows_run_device(duet_ctx *ctx, const duet_rows_problem *pr, char *out_text, uint64_t out_cap, uint64_t *out_len,
                         uint32_t *n_rows, void *stream_)
{
    if (!ctx) return duet_fail(nullptr, DUET_ERR_INVALID, "null context");
    if (!pr || !out_len || !n_rows) return duet_fail(ctx, DUET_ERR_INVALID, "null argument");
    *out_len = 0;
    *n_rows = 0;
    const uint32_t C = pr->n_cands, K = pr->n_contigs;
    if (C == 0) return DUET_OK;
    if (!pr->pred || !pr->ps || !pr->cand_pos || !pr->cand_svlen || !pr->cand_plus || !pr->cand_chrom_rank || !pr->pool ||
        !pr->str_off || !pr->cand_ctg_off || !pr->cand_off || !pr->mark_read || !out_text)
        return duet_fail(ctx, DUET_ERR_INVALID, "null array");
    if (K == 0 || K > 65535) return duet_fail(ctx, DUET_ERR_INVALID, "bad contig count");
    if (pr->n_chrom_texts == 0 || pr->n_chrom_texts > 65536) return duet_fail(ctx, DUET_ERR_INVALID, "bad CHROM text count");
    if (pr->pool_bytes + 96ull * C >= 0xFFFFFFFFull) return duet_fail(ctx, DUET_ERR_INVALID, "rows would exceed 4 GiB");
    hipStream_t st = (hipStream_t)stream_;
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    const uint32_t nb_rx = (C + kRxTile - 1) / kRxTile, nb_sc = (C + kScanTile - 1) / kScanTile;
    const uint32_t nb_hs = (256u * nb_rx + kScanTile - 1) / kScanTile;
    const size_t sizes[8] = {(size_t)C * 4, (size_t)C * 8, (size_t)C * 8, (size_t)C * 4, (size_t)C * 4, (size_t)256 * nb_rx * 4,
                             ((size_t)(nb_sc > nb_hs ? nb_sc : nb_hs) + 1) * 4, ((size_t)K + 1) * 4 + 64};
    int rc;
    for (int i = 0; i < 8; ++i)
        if ((rc = duet_reserve(ctx, ctx->rows_ws[i], sizes[i]))) return rc;
    uint32_t *idx = (uint32_t *)ctx->rows_ws[0].ptr;
    uint64_t *keysA = (uint64_t *)ctx->rows_ws[1].ptr, *keysB = (uint64_t *)ctx->rows_ws[2].ptr;
    uint32_t *valsA = (uint32_t *)ctx->rows_ws[3].ptr, *valsB = (uint32_t *)ctx->rows_ws[4].ptr;
    uint32_t *hist = (uint32_t *)ctx->rows_ws[5].ptr, *spart = (uint32_t *)ctx->rows_ws[6].ptr;
    uint32_t *d_ctg = (uint32_t *)ctx->rows_ws[7].ptr, *d_scal = d_ctg + (K + 1);     // scal: [0] rows, [1] bytes, [2] overflow

    HIP_TRY(ctx, hipMemcpyAsync(d_ctg, pr->cand_ctg_off, ((size_t)K + 1) * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(d_scal, 0, 16, st));
    // kept candidates, in file order
    launch_scan<0>(LoadKeep{pr->pred}, C, spart, StoreCompact{idx}, d_scal, st);
    uint32_t N = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&N, d_scal, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    *n_rows = N;
    if (N == 0) return DUET_OK;

    RowParams p;
    memset(&p, 0, sizeof(p));
    p.C = C; p.K = K; p.n_rows = N;
    p.pred = pr->pred; p.ps = pr->ps; p.cand_pos = pr->cand_pos; p.cand_svlen = pr->cand_svlen; p.cand_plus = pr->cand_plus;
    p.chrom_rank = pr->cand_chrom_rank; p.pool = pr->pool; p.str_off = pr->str_off; p.ctg_off = d_ctg;
    p.cand_off = pr->cand_off; p.mark_read = pr->mark_read; p.read_tag = pr->read_tag;
    const uint32_t contig_bits = bits_for(K - 1), pos_bits = bits_for(pr->max_pos ? pr->max_pos : 0xFFFFFFFFu);
    const uint32_t rank_bits = bits_for(pr->n_chrom_texts - 1);
    p.sh_pos = 2 + contig_bits;
    p.sh_rank = p.sh_pos + pos_bits;
    const uint32_t key_bits = p.sh_rank + rank_bits;
    if (key_bits > 64) return duet_fail(ctx, DUET_ERR_INVALID, "sort key does not fit 64 bits");
    hipLaunchKernelGGL(rows_keys, dim3((N + 255) / 256), dim3(256), 0, st, p, (const uint32_t *)idx, keysA, valsA);
    uint64_t *kin = nullptr;
    uint32_t *vin = nullptr;
    radix_sort_pairs(keysA, keysB, valsA, valsB, N, key_bits, hist, spart, ctx->rx_dtot, st, &kin, &vin, nullptr);
    p.sorted = vin;
    // row offsets (idx is free again) and the total
    uint32_t *row_off = idx;
    launch_scan<0>(LoadRowLen{p}, N, spart, StorePlain{row_off}, d_scal + 1, st);
    p.row_off = row_off;
    p.out = out_text; p.cap = out_cap; p.overflow = d_scal + 2;
    const uint32_t wb = (N + 3) / 4;
    hipLaunchKernelGGL(rows_write, dim3(wb < 8192u ? wb : 8192u), dim3(256), 0, st, p);
    HIP_TRY(ctx, hipGetLastError());
    uint32_t fin[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(fin, d_scal + 1, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    *out_len = fin[0];
    if (fin[1]) return duet_fail(ctx, DUET_ERR_INVALID, "output buffer too small for the rows");
    return DUET_OK;
}

int duet_ef_rows_run_host(duet_ctx *ctx, const duet_ef_problem *pr, const duet_rows_problem *rows, char *out_text, uint64_t out_cap,
                          uint64_t *out_len, uint32_t *n_rows)
{
    if (!ctx) return duet_fail(nullptr, DUET_ERR_INVALID, "null context");
    if (!pr || !rows || !out_len || !n_rows) return duet_fail(ctx, DUET_ERR_INVALID, "null argument");
    *out_len = 0;
    *n_rows = 0;
    const uint32_t C = pr->n_cands;
    if (C == 0) return DUET_OK;
    if (rows->n_cands != C || !rows->cand_plus || !rows->cand_chrom_rank || !rows->pool || !rows->str_off || !out_text)
        return duet_fail(ctx, DUET_ERR_INVALID, "rows description does not match the problem");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->own_stream;
    int rc;
    duet_ef_problem d;
    if ((rc = duet_ef_upload(ctx, pr, &d, s))) return rc;
    uint8_t *d_pred = (uint8_t *)ctx->h_out[0].ptr;
    uint32_t *d_ps = (uint32_t *)ctx->h_out[1].ptr;
    if ((rc = duet_ef_run_device(ctx, &d, d_pred, d_ps, s))) return rc;
    const uint64_t cap = rows->pool_bytes + 96ull * C + 64;
    const void *src[4] = {rows->pool, rows->str_off, rows->cand_chrom_rank, rows->cand_plus};
    const size_t bytes[4] = {(size_t)rows->pool_bytes, ((size_t)4 * C + 1) * 4, (size_t)C * 2, (size_t)C};
    for (int i = 0; i < 4; ++i) {
        if ((rc = duet_reserve(ctx, ctx->rows_in[i], bytes[i] ? bytes[i] : 16))) return rc;
        if (bytes[i]) HIP_TRY(ctx, hipMemcpyAsync(ctx->rows_in[i].ptr, src[i], bytes[i], hipMemcpyHostToDevice, s));
    }
    if ((rc = duet_reserve(ctx, ctx->rows_in[4], cap))) return rc;
    if ((rc = duet_ef_check(ctx, s))) return rc;                  // division by zero etc. surfaces here, before any row
    duet_rows_problem r = *rows;
    r.n_contigs = pr->n_contigs;
    r.cand_ctg_off = pr->cand_ctg_off;
    r.pred = d_pred; r.ps = d_ps;
    r.cand_pos = d.cand_pos; r.cand_svlen = d.cand_svlen;
    r.pool = (const char *)ctx->rows_in[0].ptr;
    r.str_off = (const uint32_t *)ctx->rows_in[1].ptr;
    r.cand_chrom_rank = (const uint16_t *)ctx->rows_in[2].ptr;
    r.cand_plus = (const uint8_t *)ctx->rows_in[3].ptr;
    r.cand_off = d.cand_off; r.mark_read = d.mark_read; r.read_tag = d.read_tag;
    uint64_t len = 0;
    if ((rc = duet_rows_run_device(ctx, &r, (char *)ctx->rows_in[4].ptr, cap, &len, n_rows, s))) return rc;
    if (len > out_cap) return duet_fail(ctx, DUET_ERR_INVALID, "output buffer too small for the rows");
    if (len) HIP_TRY(ctx, hipMemcpy(out_text, ctx->rows_in[4].ptr, len, hipMemcpyDeviceToHost));
    *out_len = len;
    return DUET_OK;
}

}  // extern "C"
