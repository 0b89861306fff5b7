// duet_prims.hip.h -- device primitives shared by the translation units of libduet_ef.so: block-tiled scans whose
// element sources / sinks are functors, and a stable LSD radix sort of (u64 key, u32 value) pairs.  Included inside
// each unit's anonymous namespace.
#ifndef DUET_PRIMS_HIP_H
#define DUET_PRIMS_HIP_H

constexpr int kRxThreads = 256;
constexpr int kRxItems = 16;
constexpr int kRxTile = kRxThreads * kRxItems;        // keys per radix block
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

// 1024 threads per 4096-key tile (the tile of rx_scatter): four keys per thread, so that a small input -- one tile per CU at
// 1 M keys -- still has 16 wavefronts per CU loading
constexpr int kRxHistThreads = 1024;
constexpr uint32_t kRxTotalsTiles = 1024;             // up to this many tiles rx_hist also accumulates the digit totals (see radix_sort_pairs)
constexpr int kDtotCopies = 8;                        // the digit totals are kept in 8 copies (by tile index): 8x fewer atomics per address
__global__ __launch_bounds__(kRxHistThreads) void rx_hist(const uint64_t *keys, uint32_t n, uint32_t shift, uint32_t dmask, uint32_t nb,
                                                          uint32_t *hist /* [256][nb] */, uint32_t *dtot /* [256] digit totals, zero on entry */)
{
    __shared__ uint32_t s_h[256];
    const uint32_t tid = threadIdx.x;
    if (tid < 256) s_h[tid] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kRxTile;
    // two keys per 16-byte load (the tile base is a multiple of kRxTile keys, the buffers are hipMalloc-aligned)
    static_assert(kRxTile == kRxHistThreads * 4, "two 16-byte loads per thread");
    ulonglong2 k2[2];
    uint32_t have[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const uint32_t i = base + (it * kRxHistThreads + tid) * 2u;
        have[it] = i + 1 < n ? 2u : (i < n ? 1u : 0u);
        k2[it] = have[it] == 2u ? *reinterpret_cast<const ulonglong2 *>(keys + i) : make_ulonglong2(have[it] ? keys[i] : 0ull, 0ull);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        if (have[it] >= 1u) atomicAdd(&s_h[(uint32_t)(k2[it].x >> shift) & dmask], 1u);
        if (have[it] == 2u) atomicAdd(&s_h[(uint32_t)(k2[it].y >> shift) & dmask], 1u);
    }
    __syncthreads();
    if (tid < 256) {
        const uint32_t c = s_h[tid];
        hist[(size_t)tid * nb + blockIdx.x] = c;
        if (dtot && c) atomicAdd(&dtot[(blockIdx.x % kDtotCopies) * 256u + tid], c);
    }
}

// hist[d][b] <- keys with a smaller digit + keys with digit d in the tiles before b: one workgroup per digit scans its
// row, the digits below it come from the totals rx_hist accumulated (one launch where a generic scan takes two or three)
__global__ __launch_bounds__(256) void rx_offsets(uint32_t *hist, uint32_t nb, const uint32_t *dtot)
{
    __shared__ uint32_t s_w[4], s_b[4];
    const uint32_t d = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t b = 0;
    if (tid < d) {
#pragma unroll
        for (int c = 0; c < kDtotCopies; ++c) b += dtot[c * 256 + tid];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b += __shfl_xor(b, o, 64);
    if (lane == 0) s_b[wave] = b;
    __syncthreads();
    uint32_t carry = s_b[0] + s_b[1] + s_b[2] + s_b[3];
    uint32_t *row = hist + (size_t)d * nb;
    for (uint32_t i0 = 0; i0 < nb; i0 += 256u * 8u) {
        uint32_t v[8], acc = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t i = i0 + tid * 8u + j;
            v[j] = i < nb ? row[i] : 0u;
            acc += v[j];
        }
        uint32_t x = acc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if ((int)lane >= o) x += y;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        uint32_t run = carry + x - acc;
        for (uint32_t w = 0; w < wave; ++w) run += s_w[w];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t i = i0 + tid * 8u + j;
            if (i < nb) row[i] = run;
            run += v[j];
        }
        carry += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
}

// generic block-tiled scan: op 0 = exclusive sum, op 1 = inclusive max
template <int OP>
__device__ __forceinline__ uint32_t scan_op(uint32_t a, uint32_t b) { return OP == 0 ? a + b : (a > b ? a : b); }

struct LoadPlain {
    const uint32_t *in;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return in[i]; }
};
struct StorePlain {
    uint32_t *out;
    __device__ __forceinline__ void operator()(uint32_t i, uint32_t v, uint32_t) const { out[i] = v; }
};
// n_dyn (may be null): the real element count when the host only knows the bound n -- tiles past it hold nothing (the neutral
// element, for both operations) and are not read
template <int OP, class Load>
__global__ __launch_bounds__(kScanThreads) void scan_reduce(const Load in, uint32_t n, uint32_t *part, uint32_t *zero14, const uint32_t *n_dyn)
{
    __shared__ uint32_t s_w[kScanThreads / 64];
    const uint32_t tid = threadIdx.x;
    if (zero14 && blockIdx.x == 0 && tid < 14) zero14[tid] = 0;     // (scans without a spine launch: see scan_spine)
    if (n_dyn) {
        const uint32_t nd = *n_dyn;
        n = nd < n ? nd : n;
        if (blockIdx.x * kScanTile >= n) {
            if (tid == 0) part[blockIdx.x] = 0;
            return;
        }
    }
    const uint32_t base = blockIdx.x * kScanTile + tid * kScanItems;
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j)
        if (base + j < n) acc = scan_op<OP>(acc, in(base + j));
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

// single block: part[i] <- combination of part[0..i) (exclusive); *total <- combination of everything.
// Thread t owns a contiguous run of entries and takes them through registers sixteen at a time (their loads leave
// together: one round trip per sixteen), the runs' totals are scanned across the block, a second sweep writes the
// entries back -- no loop of block-wide rounds.
template <int OP>
__global__ __launch_bounds__(1024) void scan_spine(uint32_t *part, uint32_t n, uint32_t *total, uint32_t *zero14)
{
    if (zero14 && threadIdx.x < 14) zero14[threadIdx.x] = 0;     // the work-list counters of the kernels that follow
    __shared__ uint32_t s_w[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t per = (n + 1023u) / 1024u, lo = min(n, tid * per), hi = min(n, lo + per);
    constexpr uint32_t kHold = 16;
    uint32_t acc = 0;
    for (uint32_t t0 = lo; t0 < hi; t0 += kHold) {
        uint32_t mine[kHold];
#pragma unroll
        for (uint32_t j = 0; j < kHold; ++j) mine[j] = part[min(t0 + j, hi - 1u)];
#pragma unroll
        for (uint32_t j = 0; j < kHold; ++j)
            if (t0 + j < hi) acc = scan_op<OP>(acc, mine[j]);
    }
    uint32_t x = acc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if ((int)lane >= d) x = scan_op<OP>(x, y);
    }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    uint32_t run = 0;
    for (uint32_t w = 0; w < wave; ++w) run = scan_op<OP>(run, s_w[w]);
    const uint32_t prev = __shfl_up(x, 1, 64);
    if (lane > 0) run = scan_op<OP>(run, prev);
    if (tid == 1023 && total) *total = scan_op<OP>(run, acc);
    for (uint32_t t0 = lo; t0 < hi; t0 += kHold) {
        uint32_t mine[kHold];
#pragma unroll
        for (uint32_t j = 0; j < kHold; ++j) mine[j] = part[min(t0 + j, hi - 1u)];
#pragma unroll
        for (uint32_t j = 0; j < kHold; ++j) {
            if (t0 + j < hi) {
                part[t0 + j] = run;
                run = scan_op<OP>(run, mine[j]);
            }
        }
    }
}

// out[i] = exclusive sum (OP 0) / inclusive max (OP 1) of in[0..i] given the per-tile carries in part[].
// SELF: there was no spine launch -- part[] still holds the tiles' own totals and every block combines the ones before
// it by itself (up to kSelfSpine of them: a few loads per thread, cheaper than one more dependent launch); the last
// block also delivers *total.
constexpr uint32_t kSelfSpine = 2048;
template <int OP, class Load, class Store, bool SELF>
__global__ __launch_bounds__(kScanThreads) void scan_apply(const Load in, uint32_t n, const uint32_t *part, const Store out,
                                                           uint32_t *total, const uint32_t *n_dyn)
{
    __shared__ uint32_t s_w[kScanThreads / 64];
    __shared__ uint32_t s_c[kScanThreads / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (n_dyn) {
        const uint32_t nd = *n_dyn;
        n = nd < n ? nd : n;
        // a tile past the real count has nothing to store; only the last block of a spine-less scan still has the total to deliver
        if (blockIdx.x * kScanTile >= n && !(SELF && total && blockIdx.x == gridDim.x - 1)) return;
    }
    const uint32_t base = blockIdx.x * kScanTile + tid * kScanItems;
    uint32_t v[kScanItems];
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        v[j] = base + j < n ? in(base + j) : 0u;
        acc = scan_op<OP>(acc, v[j]);
    }
    uint32_t x = acc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if ((int)lane >= d) x = scan_op<OP>(x, y);
    }
    if (lane == 63) s_w[wave] = x;
    uint32_t carry = 0;
    if (SELF) {
        for (uint32_t i = tid; i < blockIdx.x; i += kScanThreads) carry = scan_op<OP>(carry, part[i]);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) carry = scan_op<OP>(carry, __shfl_xor(carry, d, 64));
        if (lane == 0) s_c[wave] = carry;
    }
    __syncthreads();
    if (SELF) {
        carry = 0;
#pragma unroll
        for (int w = 0; w < kScanThreads / 64; ++w) carry = scan_op<OP>(carry, s_c[w]);
        if (total && tid == 0 && blockIdx.x == gridDim.x - 1) {
            uint32_t t = carry;
#pragma unroll
            for (int w = 0; w < kScanThreads / 64; ++w) t = scan_op<OP>(t, s_w[w]);
            *total = t;
        }
    } else {
        carry = part[blockIdx.x];
    }
    uint32_t run = carry;
    for (uint32_t w = 0; w < wave; ++w) run = scan_op<OP>(run, s_w[w]);
    const uint32_t prev = __shfl_up(x, 1, 64);
    if (lane > 0) run = scan_op<OP>(run, prev);
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        if (OP == 0) {
            if (base + j < n) out(base + j, run, v[j]);
            run += v[j];
        } else {
            run = scan_op<OP>(run, v[j]);
            if (base + j < n) out(base + j, run, v[j]);
        }
    }
}

// stable scatter of one digit of up to 8 bits (dmask: the last pass of a sort may take fewer, so that whatever sits above the
// sorted bits -- a payload packed into the key -- stays out of it); hist holds the scanned (digit-major) offsets.
// VALS = false: keys only.
// Each wave of a block owns a contiguous quarter of the block's 4096 keys and ranks it on its own (ballot match
// inside the wave, a running per-digit count in the wave's LDS row: no workgroup barrier inside the loop; block
// order = wave, round, lane = input order, so the sort stays stable).  The tile is then laid out digit-sorted in
// LDS and written from there: consecutive lanes hold consecutive keys of one digit run, so each run leaves as whole
// cache lines in one go.  (Writing straight from the ranking loop touched every run one 8-byte key at a time; with
// thousands of blocks in flight the partly written lines fell out of L2 and the scatter ran at half this speed.)
template <bool VALS>
__global__ __launch_bounds__(kRxThreads) void rx_scatter(const uint64_t *keys_in, const uint32_t *vals_in, uint32_t n,
                                                         uint32_t shift, uint32_t dmask, uint32_t nb, const uint32_t *hist,
                                                         uint64_t *keys_out, uint32_t *vals_out, uint32_t *dtot)
{
    constexpr int kWaves = kRxThreads / 64, kPerWave = kRxTile / kWaves;
    __shared__ uint64_t s_key[kRxTile];
    __shared__ uint32_t s_val[VALS ? kRxTile : 1];
    __shared__ uint32_t s_gbase[256];                      // global position of the block's first key of each digit
    __shared__ uint32_t s_start[256];                      // where each digit starts inside the tile
    __shared__ uint32_t s_wloc[kWaves][256];               // per wave: keys of each digit so far; then the wave's offset
    __shared__ uint32_t s_wsum[kWaves];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    s_gbase[tid] = hist[(size_t)tid * nb + blockIdx.x];
#pragma unroll
    for (int w = 0; w < kWaves; ++w) s_wloc[w][tid] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kRxTile, wbase = base + wave * kPerWave;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    uint64_t key[kRxItems];
    uint32_t val[VALS ? kRxItems : 1], lrank[kRxItems];
#pragma unroll
    for (int it = 0; it < kRxItems; ++it) {
        const uint32_t i = wbase + it * 64 + lane;
        key[it] = i < n ? keys_in[i] : ~0ull;
        if (VALS) val[it] = i < n ? vals_in[i] : 0u;
    }
#pragma unroll
    for (int it = 0; it < kRxItems; ++it) {
        const bool valid = wbase + it * 64 + lane < n;
        const uint32_t d = (uint32_t)(key[it] >> shift) & dmask;
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bm = __ballot(bit);
            same &= bit ? bm : ~bm;
        }
        const uint32_t rank = (uint32_t)__popcll(same & lt);
        const uint32_t seen = s_wloc[wave][d];
        lrank[it] = seen + rank;
        __builtin_amdgcn_wave_barrier();                   // every lane has read the count before its leader bumps it
        if (valid && rank == 0) s_wloc[wave][d] = seen + (uint32_t)__popcll(same);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // digit starts inside the tile (exclusive scan of the digit totals over the 256 threads) and, per wave, the
    // keys of the same digit in earlier waves
    {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const uint32_t c = s_wloc[w][tid];
            s_wloc[w][tid] = t;
            t += c;
        }
        uint32_t x = t;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t y = __shfl_up(x, dd, 64);
            if ((int)lane >= dd) x += y;
        }
        if (lane == 63) s_wsum[wave] = x;
        __syncthreads();
        uint32_t carry = 0;
        for (uint32_t w = 0; w < wave; ++w) carry += s_wsum[w];
        s_start[tid] = carry + x - t;
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < kRxItems; ++it) {
        if (wbase + it * 64 + lane < n) {
            const uint32_t d = (uint32_t)(key[it] >> shift) & dmask;
            const uint32_t at = s_start[d] + s_wloc[wave][d] + lrank[it];
            s_key[at] = key[it];
            if (VALS) s_val[at] = val[it];
        }
    }
    __syncthreads();
    const uint32_t count = min((uint32_t)kRxTile, n - base);
#pragma unroll
    for (int it = 0; it < kRxItems; ++it) {
        const uint32_t q = it * kRxThreads + tid;
        if (q < count) {
            const uint64_t k = s_key[q];
            const uint32_t d = (uint32_t)(k >> shift) & dmask;
            const uint32_t at = s_gbase[d] + (q - s_start[d]);
            keys_out[at] = k;
            if (VALS) vals_out[at] = s_val[q];
        }
    }
    if (dtot && blockIdx.x < kDtotCopies) dtot[blockIdx.x * 256u + tid] = 0;   // rx_offsets is done with the totals: zero again for the next pass
}

uint32_t bits_for(uint64_t max_value)
{
    uint32_t b = 0;
    while (b < 64 && (max_value >> b)) ++b;
    return b ? b : 1;
}

template <int OP, class Load, class Store>
void launch_scan(const Load in, uint32_t n, uint32_t *part, const Store out, uint32_t *total, hipStream_t st,
                 uint32_t *zero14 = nullptr, bool force_spine = false /* tests: the path of more than kSelfSpine tiles */,
                 const uint32_t *n_dyn = nullptr /* device: the real element count, n being its bound */)
{
    const uint32_t nb = (n + kScanTile - 1) / kScanTile;
    // (a device-side element count, usually a fraction of the bound -- stage A0's partitions are 7 % of its marks --: the tiles
    // past it neither load nor combine anything, so the spine-less path also pays where the BOUND has more tiles than kSelfSpine;
    // up to kSelfSpineDyn of them, so that an input whose count does reach the bound costs tens of microseconds, not more)
    constexpr uint32_t kSelfSpineDyn = 16384;
    if (nb >= 1 && (nb <= kSelfSpine || (n_dyn && nb <= kSelfSpineDyn)) && !force_spine) {
        hipLaunchKernelGGL((scan_reduce<OP, Load>), dim3(nb), dim3(kScanThreads), 0, st, in, n, part, zero14, n_dyn);
        hipLaunchKernelGGL((scan_apply<OP, Load, Store, true>), dim3(nb), dim3(kScanThreads), 0, st, in, n, (const uint32_t *)part, out,
                           total, n_dyn);
        return;
    }
    hipLaunchKernelGGL((scan_reduce<OP, Load>), dim3(nb), dim3(kScanThreads), 0, st, in, n, part, (uint32_t *)nullptr, n_dyn);
    hipLaunchKernelGGL(scan_spine<OP>, dim3(1), dim3(1024), 0, st, part, nb, total, zero14);
    hipLaunchKernelGGL((scan_apply<OP, Load, Store, false>), dim3(nb), dim3(kScanThreads), 0, st, in, n, (const uint32_t *)part, out,
                       (uint32_t *)nullptr, n_dyn);
}

// stable LSD radix sort of n (key, value) pairs -- or, with null value buffers, of the keys alone -- on the low key_bits of
// the keys (bits above them are carried along untouched); buffers A hold the input, the result ends in whichever buffers
// *keys_out / *vals_out point to afterwards.  hist: 256 * ceil(n / kRxTile) words,
// spart: ceil(256 * ceil(n / kRxTile) / kScanTile) + 1 words.
inline void radix_sort_pairs(uint64_t *keysA, uint64_t *keysB, uint32_t *valsA, uint32_t *valsB, uint32_t n, uint32_t key_bits,
                             uint32_t *hist, uint32_t *spart, uint32_t *dtot /* [kDtotCopies][256], zero between sorts */, hipStream_t st,
                             uint64_t **keys_out, uint32_t **vals_out, uint64_t **keys_spare, bool force_scan = false,
                             bool first_hist_done = false /* the producer of the keys already filled hist (and dtot) for the first digit */,
                             uint32_t first_shift = 0 /* the passes cover bits [first_shift, key_bits) only */)
{
    const uint32_t nb_rx = (n + kRxTile - 1) / kRxTile, nh = 256 * nb_rx;
    // up to 1024 tiles (4 M keys) the tile offsets take ONE launch: rx_hist also accumulates the digit totals (atomics, a
    // few hundred per address) and rx_offsets scans one digit row per workgroup; beyond, the atomics would cost more
    // than the launch they save (2e7 keys: +40 us per pass) and the generic scan does it
    if (nb_rx > kRxTotalsTiles || force_scan) dtot = nullptr;
    uint64_t *kin = keysA, *kout = keysB;
    uint32_t *vin = valsA, *vout = valsB;
    for (uint32_t shift = first_shift; shift < key_bits; shift += 8) {
        const uint32_t dmask = key_bits - shift >= 8 ? 255u : (1u << (key_bits - shift)) - 1u;
        if (shift != first_shift || !first_hist_done)
            hipLaunchKernelGGL(rx_hist, dim3(nb_rx), dim3(kRxHistThreads), 0, st, (const uint64_t *)kin, n, shift, dmask, nb_rx, hist, dtot);
        if (dtot) hipLaunchKernelGGL(rx_offsets, dim3(256), dim3(256), 0, st, hist, nb_rx, (const uint32_t *)dtot);
        else launch_scan<0>(LoadPlain{hist}, nh, spart, StorePlain{hist}, nullptr, st, nullptr, force_scan);   // in place: scan_apply reads a tile before writing it
        if (valsA)
            hipLaunchKernelGGL(rx_scatter<true>, dim3(nb_rx), dim3(kRxThreads), 0, st, (const uint64_t *)kin, (const uint32_t *)vin, n,
                               shift, dmask, nb_rx, (const uint32_t *)hist, kout, vout, dtot);
        else
            hipLaunchKernelGGL(rx_scatter<false>, dim3(nb_rx), dim3(kRxThreads), 0, st, (const uint64_t *)kin, (const uint32_t *)nullptr,
                               n, shift, dmask, nb_rx, (const uint32_t *)hist, kout, (uint32_t *)nullptr, dtot);
        uint64_t *tk = kin; kin = kout; kout = tk;
        uint32_t *tv = vin; vin = vout; vout = tv;
    }
    *keys_out = kin;
    if (vals_out) *vals_out = vin;
    if (keys_spare) *keys_spare = kout;
}


// ---------------------------------------------------------------------------------------------
// the low bits, locally: keys that global passes have ordered by their bits [lo, key_bits) (stably: ties in input order)
// ---------------------------------------------------------------------------------------------
//
// A GROUP is a run of keys that agree in the bits [lo, key_bits).  Small inputs on a genome have small groups (a group is the
// marks of one type within 2^lo centres): each is put in order by its low bits with a rank count in LDS -- no more passes over
// the whole array.  rx_local: one workgroup per tile of kLocTile positions takes the groups that START in its tile and hold at
// most `cap` keys (cap <= kLocHalo: such a group ends inside the tile's window); larger groups go on a list for rx_big.
// The result is stable (ties keep their order), so the whole sort is.
constexpr int kLocTile = 2048, kLocHalo = 256;
// E: the sort element -- an 8-byte key (the mark index in its spare bits) or, where the record travels with the key, the
// 16-byte mark record (duet_recsort.hip.h); keyof(e) is its sort key
template <class E> __device__ __forceinline__ E rx_zero();
template <> __device__ __forceinline__ uint64_t rx_zero<uint64_t>() { return 0ull; }
template <> __device__ __forceinline__ uint4 rx_zero<uint4>() { return make_uint4(0u, 0u, 0u, 0u); }
template <int THREADS, class E, class KeyOf>               // 1024 threads for small inputs (a tile's latency counts), 256 for large ones (more tiles in flight)
__global__ __launch_bounds__(THREADS) void rx_local(const E *in, E *out, uint32_t n, uint32_t lo, const KeyOf keyof, uint32_t cap,
                                                    uint32_t *big_list, uint32_t *big_count)
{
    constexpr int W = kLocTile + kLocHalo, kNone = 0x7FFF;
    constexpr int kIter = (W + 1 + THREADS - 1) / THREADS;  // keys per thread (position tid + it * THREADS: it keeps them in registers)
    constexpr int kPer = kIter;                            // positions per thread in the scans (tid * kPer + j)
    __shared__ uint32_t s_low[W + 4];                      // the keys' low bits (lo < 32 for every input this path takes), see below
    __shared__ uint8_t s_head[W + 1];                      // the position starts a group (or lies behind the last key)
    __shared__ int16_t s_gs[W + 1];                        // start of the position's group inside the window (-1: it starts before the window)
    __shared__ int16_t s_ge[W + 1];                        // end of the position's group (first position behind it; kNone: beyond the window)
    __shared__ int s_carry[2][THREADS / 64];
    __shared__ uint64_t s_edge[kIter][THREADS / 64];       // the group number of every wave's last key, per round of loads
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, t0 = blockIdx.x * kLocTile;
    const uint32_t lmask = (1u << lo) - 1u;                 // (lo < 32: the callers see to it)
    // (up to 20 low bits: a key's low bits and its place in the window make ONE 32-bit number, distinct for every key -- the
    // rank is a count of smaller numbers, one compare per key; beyond, low bits and place are compared separately)
    const bool packed = lo <= 20u;
    E key[kIter];
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        const uint32_t i = tid + it * THREADS;
        key[it] = (i <= (uint32_t)W && t0 + i < n) ? in[t0 + i] : rx_zero<E>();
    }
    // (the key in front of the window leaves with the others: a lane 0 that fetched its predecessor when it got to it made
    // every round of the loop below wait for a round trip of its own -- ten in a row per tile)
    const E before = (tid == 0 && t0 > 0u) ? in[t0 - 1u] : rx_zero<E>();
#pragma unroll
    for (int it = 0; it < kIter; ++it)
        if (lane == 63u) s_edge[it][wave] = keyof(key[it]) >> lo;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        const uint32_t i = tid + it * THREADS;
        const bool live = i <= (uint32_t)W && t0 + i < n;
        const uint64_t kk = keyof(key[it]);
        const uint64_t g = kk >> lo;
        // the group number of the position in front: the lane in front holds it; lane 0 has it from the wave in front
        uint64_t pg = ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(g >> 32), 1, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)g, 1, 64);
        if (lane == 0) pg = wave > 0 ? s_edge[it][wave - 1u] : (it > 0 ? s_edge[it > 0 ? it - 1 : 0][THREADS / 64 - 1] : keyof(before) >> lo);
        if (i <= (uint32_t)W) {
            s_head[i] = (!live || t0 + i == 0u || g != pg) ? 1 : 0;      // (the end of the keys closes the last group)
            const uint32_t low = (uint32_t)kk & lmask;
            s_low[i] = packed ? (low << 12) | i : low;
        }
    }
    __syncthreads();
    // group starts: a running maximum of (head ? position : -1) over the window; ends: a running minimum from the right of
    // (head ? position : kNone) over the positions behind each one.  kPer consecutive positions per thread
    bool head[kPer];
    int mine[kPer];
    int run = -1;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const uint32_t i = tid * kPer + j;
        head[j] = i <= (uint32_t)W && s_head[i] != 0;
        run = head[j] ? (int)i : run;
        mine[j] = run;
    }
    int x = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if ((int)lane >= d) x = max(x, y);
    }
    if (lane == 63) s_carry[0][wave] = x;
    int back[kPer];
    int rrun = kNone;
#pragma unroll
    for (int j = kPer - 1; j >= 0; --j) {
        const uint32_t i = tid * kPer + j;
        back[j] = rrun;                                    // (the nearest head strictly behind position i, within this thread's positions)
        rrun = head[j] ? (int)i : rrun;
    }
    int z = rrun;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_down(z, d, 64);
        if ((int)lane + d < 64) z = min(z, y);
    }
    if (lane == 0) s_carry[1][wave] = z;
    __syncthreads();
    int carry = -1, rcarry = kNone;
    for (uint32_t w = 0; w < wave; ++w) carry = max(carry, s_carry[0][w]);
    for (uint32_t w = wave + 1; w < (uint32_t)(THREADS / 64); ++w) rcarry = min(rcarry, s_carry[1][w]);
    const int px = __shfl_up(x, 1, 64), nz = __shfl_down(z, 1, 64);
    if (lane > 0) carry = max(carry, px);
    if (lane < 63) rcarry = min(rcarry, nz);
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const uint32_t i = tid * kPer + j;
        if (i <= (uint32_t)W) {
            s_gs[i] = (int16_t)max(mine[j], carry);
            s_ge[i] = (int16_t)min(back[j], rcarry);
        }
    }
    __syncthreads();
    // the groups that start in the tile and hold more than cap keys go on the list (their first keys report them)
    for (uint32_t i = tid; i < (uint32_t)kLocTile; i += THREADS)
        if (t0 + i < n && s_gs[i] == (int)i && (uint32_t)(s_ge[i] - (int)i) > cap) big_list[atomicAdd(big_count, 1u)] = t0 + i;
    // every key of a group this tile owns: its rank by the low bits, ties in input order
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        const uint32_t i = tid + it * THREADS;
        if (i >= (uint32_t)W || t0 + i >= n) continue;
        const int gs = s_gs[i];
        if (gs < 0 || gs >= kLocTile) continue;
        const uint32_t len = (uint32_t)(s_ge[gs] - gs);
        if (len > cap) continue;
        const uint32_t me = s_low[i], e = (uint32_t)gs + len;
        uint32_t rank = 0, j = (uint32_t)gs;
        if (packed) {
            for (const uint32_t e4 = (uint32_t)gs + (len & ~3u); j < e4; j += 4)
                rank += (uint32_t)(s_low[j] < me) + (uint32_t)(s_low[j + 1] < me) + (uint32_t)(s_low[j + 2] < me) + (uint32_t)(s_low[j + 3] < me);
            for (; j < e; ++j) rank += (uint32_t)(s_low[j] < me);
        } else {
            for (; j < e; ++j) {
                const uint32_t o = s_low[j];
                rank += (uint32_t)((o < me) | ((o == me) & (j < i)));    // (bitwise on purpose: no branch per key)
            }
        }
        out[t0 + (uint32_t)gs + rank] = key[it];
    }
}

// The groups of more than cap keys: one workgroup per group.  Up to kBigLds keys: loaded into LDS and ranked there like rx_local
// does; beyond that stable LSD passes over the low bits within the group's own range of the two buffers (256 keys at a time:
// ballot match inside a wave, the waves' counts side by side).  Rare on a genome; slow but sure when a locus piles up (2 us per
// 256 keys and pass).  a: the keys as the global passes left them (the groups' ranges of b are nobody else's); the result
// ends in b.
constexpr int kBigLds = 1024;
template <class E, class KeyOf>
__global__ __launch_bounds__(256) void rx_big(E *a, E *b, uint32_t n, uint32_t lo, const KeyOf keyof, const uint32_t *big_list,
                                              const uint32_t *big_count)
{
    __shared__ E s_k[kBigLds];
    __shared__ uint32_t s_low[kBigLds + 4];
    __shared__ uint32_t s_base[256];                       // where the next key of each digit goes
    __shared__ uint32_t s_wcnt[4][256];
    __shared__ uint32_t s_scan[4];
    __shared__ uint32_t s_end;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t lmask = (1u << lo) - 1u;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const uint32_t count = *big_count;
    for (uint32_t e = blockIdx.x; e < count; e += gridDim.x) {
        const uint32_t gs = big_list[e];
        __syncthreads();
        if (tid == 0) s_end = 0xFFFFFFFFu;
        __syncthreads();
        // the group's first kBigLds keys (and the one behind them) in one round of loads: most groups end among them
        uint64_t g = 0;
        {
            g = keyof(a[gs]) >> lo;
            for (uint32_t i = tid; i <= (uint32_t)kBigLds; i += 256u) {
                const uint32_t pos = gs + i;
                const E k = pos < n ? a[pos] : rx_zero<E>();
                const uint64_t kk = keyof(k);
                const bool in = pos < n && (kk >> lo) == g;
                if (i < (uint32_t)kBigLds) { s_k[i] = k; s_low[i] = (uint32_t)kk & lmask; }
                if (!in) atomicMin(&s_end, i);              // (group numbers do not decrease along a: the first "no" ends the group)
            }
        }
        __syncthreads();
        const uint32_t len_lds = s_end;
        if (len_lds <= (uint32_t)kBigLds) {
            for (uint32_t i = tid; i < len_lds; i += 256u) {
                const uint32_t me = s_low[i];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < len_lds; j += 4) {
                    const uint32_t o0 = s_low[j], o1 = s_low[j + 1], o2 = s_low[j + 2], o3 = s_low[j + 3];
                    rank += (uint32_t)((o0 < me) | ((o0 == me) & (j < i)));
                    rank += (uint32_t)((j + 1 < len_lds) & ((o1 < me) | ((o1 == me) & (j + 1 < i))));
                    rank += (uint32_t)((j + 2 < len_lds) & ((o2 < me) | ((o2 == me) & (j + 2 < i))));
                    rank += (uint32_t)((j + 3 < len_lds) & ((o3 < me) | ((o3 == me) & (j + 3 < i))));
                }
                b[gs + rank] = s_k[i];
            }
            continue;
        }
        uint32_t lo_i = gs + (uint32_t)kBigLds, hi_i = n;  // first position behind the group, by bisection
        while (hi_i - lo_i > 1u) {
            const uint32_t mid = lo_i + ((hi_i - lo_i) >> 1);
            if ((keyof(a[mid]) >> lo) == g) lo_i = mid; else hi_i = mid;
        }
        const uint32_t ge = hi_i;
        E *src = a, *dst = b;
        for (uint32_t shift = 0; shift < lo; shift += 8) {
            const uint32_t dmask = lo - shift >= 8u ? 255u : (1u << (lo - shift)) - 1u;
            __syncthreads();
            s_base[tid] = 0;
            __syncthreads();
            for (uint32_t i = gs + tid; i < ge; i += 256u) atomicAdd(&s_base[(uint32_t)(keyof(src[i]) >> shift) & dmask], 1u);
            __syncthreads();
            {   // exclusive scan of the 256 digit counts
                const uint32_t c = s_base[tid];
                uint32_t x = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t y = __shfl_up(x, d, 64);
                    if ((int)lane >= d) x += y;
                }
                if (lane == 63) s_scan[wave] = x;
                __syncthreads();
                uint32_t carry = 0;
                for (uint32_t w = 0; w < wave; ++w) carry += s_scan[w];
                s_base[tid] = gs + carry + x - c;
            }
            for (uint32_t c0 = gs; c0 < ge; c0 += 256u) {
                __syncthreads();
#pragma unroll
                for (int w = 0; w < 4; ++w) s_wcnt[w][tid] = 0;
                __syncthreads();
                const uint32_t i = c0 + tid;
                const bool valid = i < ge;
                const E k = valid ? src[i] : rx_zero<E>();
                const uint32_t d = (uint32_t)(keyof(k) >> shift) & dmask;
                unsigned long long same = __ballot(valid);
#pragma unroll
                for (int bit = 0; bit < 8; ++bit) {
                    const bool on = (d >> bit) & 1u;
                    const unsigned long long bm = __ballot(on);
                    same &= on ? bm : ~bm;
                }
                const uint32_t rank = (uint32_t)__popcll(same & lt);
                if (valid && rank == 0) s_wcnt[wave][d] = (uint32_t)__popcll(same);
                __syncthreads();
                if (valid) {
                    uint32_t at = s_base[d] + rank;
                    for (uint32_t w = 0; w < wave; ++w) at += s_wcnt[w][d];
                    dst[at] = k;
                }
                __syncthreads();
                s_base[tid] += s_wcnt[0][tid] + s_wcnt[1][tid] + s_wcnt[2][tid] + s_wcnt[3][tid];
            }
            __syncthreads();
            E *t = src; src = dst; dst = t;
        }
        if (src == a) {                                    // an even number of passes left the result in a
            __threadfence_block();
            for (uint32_t i = gs + tid; i < ge; i += 256u) b[i] = a[i];
        }
    }
}

#endif
