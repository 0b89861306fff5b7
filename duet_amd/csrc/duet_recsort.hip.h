// duet_recsort.hip.h -- stage A0's sort when the mark RECORD travels with the key (round 4).  Included inside
// duet_cluster.hip's anonymous namespace, after duet_prims.hip.h.
//
// The key-only sort of rounds 1-3 moved 8-byte keys (contig | type | centre, the mark index in the spare bits) and left the
// marks' records where the caller's order put them: the box test then fetched every 16-byte record through the sort
// permutation, a 128-byte line apiece on a shuffled input (2e7 marks: 517 us, 140 B of traffic per mark).  Here the record
// IS the sort element:
//     x = pos, y = span, z = read index (fused pipeline; else 0), w = (contig << type_bits | type) << idx_bits | mark index
// -- 16 bytes, and the key is a function of it (contig | type in w's high bits, centre = x + y / 2), so no key array exists
// at all.  Taken when contig, type and mark index fit w's 32 bits (2e7 marks over a genome: 5 + 1 + 25); everything else
// keeps the key-only path.
//
//   rs_hist<RAW>      per 4096-mark tile, the digit counts of one pass; table laid out TILE-major (a tile's row is one
//                     coalesced store here and one coalesced load in rs_scatter -- with 1024-way digits a digit-major table
//                     costs a 4-byte access per digit and tile on both sides, as many sectors as the records themselves)
//   rs_offsets_small  (up to 1024 tiles) one launch: digit totals from rs_hist's atomics, a workgroup per 64 digits runs down
//                     its columns in 16 segments
//   rs_col_*          (beyond) column sums per chunk of tiles -> one spine block -> exclusive offsets written back
//   rs_scatter<WB>    stable scatter of one digit of up to WB bits: ballot-ranked per wave, the tile laid out digit-sorted in
//                     LDS and written from there (runs leave as whole lines).  The first pass reads the caller's arrays and
//                     builds the records (RAW).  Tiles are dealt to workgroups so that the workgroups of one XCD hold a
//                     CONTIGUOUS range of tiles: the runs a digit receives from neighbouring tiles are neighbours in memory,
//                     and with 1024-way digits a run is 4 records = half a line -- the halves meet in that XCD's L2.
//   the low bits      rx_local / rx_big (duet_prims.hip.h) instantiated on records: groups of marks that agree in the globally
//                     sorted top bits are ordered by a rank count in LDS, the record stored to its place
//
// Passes: LSD over the key's top bits [lo, key_bits), lo chosen so that a group holds a few dozen marks (16-17 bits of a
// 1 M-mark input's key in two 8-bit passes, 20 bits of a 2e7-mark genome's in two 10-bit passes), each pass stable, so equal
// keys keep their input order (rule 1 of oracle/cluster_oracle.c).
#ifndef DUET_RECSORT_HIP_H
#define DUET_RECSORT_HIP_H

constexpr int kRsTile = 4096;                          // marks per tile (rs_hist and rs_scatter)
constexpr int kRsThreads = 256, kRsItems = kRsTile / kRsThreads;
constexpr int kRsHistThreads = 1024;
constexpr int kRsMaxW = 10;                            // bits of one global digit, at most
constexpr int kRsDtotCopies = 8;
constexpr uint32_t kRsSmallTiles = 1024;               // up to this many tiles the offsets take one launch (digit totals by atomics)
constexpr int kRsChunk = 16;                           // tiles per column chunk beyond

// the caller's arrays, and how a record is made of mark i
struct RsSrc {
    const uint16_t *contig;
    const uint8_t *type;
    const uint32_t *pos, *span, *read;                 // read: null outside the fused pipeline
    uint32_t type_bits, idx_bits, centre_bits;
};
__device__ __forceinline__ uint4 rs_make(const RsSrc &s, uint32_t i)
{
    const uint32_t hi = ((uint32_t)s.contig[i] << s.type_bits) | (uint32_t)s.type[i];
    return make_uint4(s.pos[i], s.span[i], s.read ? s.read[i] : 0u, (hi << s.idx_bits) | i);
}
__device__ __forceinline__ uint64_t rs_key(const uint4 &r, uint32_t centre_bits, uint32_t idx_bits)
{
    return ((uint64_t)(r.w >> idx_bits) << centre_bits) | ((uint64_t)r.x + (r.y >> 1));
}
// sort-key functors: what the partition scan, rx_local and rx_big see of an element
struct KeyOfU64 {
    uint64_t km;                                       // the key proper (a packed mark index sits above it)
    __device__ __forceinline__ uint64_t operator()(const uint64_t &e) const { return e & km; }
};
struct KeyOfRec {
    uint32_t centre_bits, idx_bits;
    __device__ __forceinline__ uint64_t operator()(const uint4 &r) const { return rs_key(r, centre_bits, idx_bits); }
};

// workgroup -> tile, so that the workgroups that share an XCD (equal blockIdx % 8 under the dispatcher's round robin; a speed
// matter only) take a contiguous range of tiles.  Bijective for any tile count.
__device__ __forceinline__ uint32_t rs_tile_of(uint32_t bid, uint32_t nb)
{
    const uint32_t q = nb >> 3, r = nb & 7u, x = bid & 7u;
    return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + (bid >> 3);
}

template <bool RAW>
__global__ __launch_bounds__(kRsHistThreads) void rs_hist(const RsSrc src, const uint4 *in, uint32_t n, uint32_t shift, uint32_t wbits, uint32_t *hist /* [tiles][bins] */,
                                                          uint32_t *dtot /* [kRsDtotCopies][bins], zero on entry; or null */, uint32_t *zero /* a counter of a later launch, or null */)
{
    __shared__ uint32_t s_h[1 << kRsMaxW];
    const uint32_t tid = threadIdx.x, bins = 1u << wbits, tile = blockIdx.x;
    if (zero && tile == 0 && tid == 0) *zero = 0;
    for (uint32_t d = tid; d < bins; d += kRsHistThreads) s_h[d] = 0;
    __syncthreads();
    uint64_t key[kRsTile / kRsHistThreads];
    bool live[kRsTile / kRsHistThreads];
#pragma unroll
    for (int it = 0; it < kRsTile / kRsHistThreads; ++it) {
        const uint32_t i = tile * kRsTile + it * kRsHistThreads + tid;
        live[it] = i < n;
        key[it] = 0;
        if (live[it]) {
            if (RAW) {
                const uint64_t hi = ((uint64_t)src.contig[i] << src.type_bits) | (uint64_t)src.type[i];
                key[it] = (hi << src.centre_bits) | ((uint64_t)src.pos[i] + (src.span[i] >> 1));
            } else {
                key[it] = rs_key(in[i], src.centre_bits, src.idx_bits);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < kRsTile / kRsHistThreads; ++it)
        if (live[it]) atomicAdd(&s_h[(uint32_t)(key[it] >> shift) & (bins - 1u)], 1u);
    __syncthreads();
    for (uint32_t d = tid; d < bins; d += kRsHistThreads) {
        const uint32_t c = s_h[d];
        hist[(size_t)tile * bins + d] = c;
        if (dtot && c) atomicAdd(&dtot[(tile % kRsDtotCopies) * bins + d], c);
    }
}

// exclusive scan over a workgroup of 1024 threads (x: the thread's value); s_w: [16]
__device__ __forceinline__ uint32_t rs_block_exscan_1024(uint32_t x, uint32_t *s_w)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t inc = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(inc, o, 64);
        if ((int)lane >= o) inc += y;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t carry = 0;
    for (uint32_t w = 0; w < wave; ++w) carry += s_w[w];
    return carry + inc - x;
}

// hist[t][d] <- marks with a smaller digit + marks with digit d in the tiles before t.  One launch: the digit totals come from
// rs_hist's atomics; workgroup b takes the digits 64 b .. 64 b + 63, its 1024 threads = 16 segments of tiles x 64 digits
// (a wave reads 64 consecutive digits of one tile: one 256-byte row piece)
__global__ __launch_bounds__(1024) void rs_offsets_small(uint32_t *hist, uint32_t nb, uint32_t wbits, const uint32_t *dtot)
{
    __shared__ uint32_t s_w[16], s_base[1 << kRsMaxW], s_seg[16][64];
    const uint32_t tid = threadIdx.x, bins = 1u << wbits;
    uint32_t tot = 0;
    if (tid < bins) {
#pragma unroll
        for (int c = 0; c < kRsDtotCopies; ++c) tot += dtot[c * bins + tid];
    }
    const uint32_t ex = rs_block_exscan_1024(tot, s_w);
    if (tid < bins) s_base[tid] = ex;
    const uint32_t seg = tid >> 6, dl = tid & 63u, d = blockIdx.x * 64u + dl;
    const uint32_t per = (nb + 15u) / 16u, t_lo = min(nb, seg * per), t_hi = min(nb, t_lo + per);
    const bool on = d < bins;
    uint32_t sum = 0;
    if (on)
        for (uint32_t t = t_lo; t < t_hi; ++t) sum += hist[(size_t)t * bins + d];
    s_seg[seg][dl] = sum;
    __syncthreads();
    if (!on) return;
    uint32_t run = s_base[d];
    for (uint32_t s = 0; s < seg; ++s) run += s_seg[s][dl];
    for (uint32_t t = t_lo; t < t_hi; ++t) {
        const uint32_t v = hist[(size_t)t * bins + d];
        hist[(size_t)t * bins + d] = run;
        run += v;
    }
}

// more tiles: column sums per chunk of kRsChunk tiles, one block for the chunks' running sums and the digits' bases, the
// offsets written back per chunk
__global__ __launch_bounds__(256) void rs_col_reduce(const uint32_t *hist, uint32_t nb, uint32_t wbits, uint32_t *partial /* [chunks][bins] */)
{
    const uint32_t bins = 1u << wbits, c = blockIdx.x, t0 = c * kRsChunk, t1 = min(nb, t0 + kRsChunk);
    for (uint32_t d = threadIdx.x; d < bins; d += 256u) {
        uint32_t s = 0;
        for (uint32_t t = t0; t < t1; ++t) s += hist[(size_t)t * bins + d];
        partial[(size_t)c * bins + d] = s;
    }
}
__global__ __launch_bounds__(1024) void rs_col_spine(uint32_t *partial, uint32_t nchunk, uint32_t wbits, uint32_t *base /* [bins] */)
{
    __shared__ uint32_t s_w[16];
    const uint32_t tid = threadIdx.x, bins = 1u << wbits;
    uint32_t run = 0;
    if (tid < bins) {
        constexpr uint32_t kHold = 8;
        for (uint32_t c0 = 0; c0 < nchunk; c0 += kHold) {
            uint32_t v[kHold];
#pragma unroll
            for (uint32_t j = 0; j < kHold; ++j) v[j] = c0 + j < nchunk ? partial[(size_t)(c0 + j) * bins + tid] : 0u;
#pragma unroll
            for (uint32_t j = 0; j < kHold; ++j) {
                if (c0 + j < nchunk) partial[(size_t)(c0 + j) * bins + tid] = run;
                run += v[j];
            }
        }
    }
    const uint32_t ex = rs_block_exscan_1024(run, s_w);
    if (tid < bins) base[tid] = ex;
}
__global__ __launch_bounds__(256) void rs_col_apply(uint32_t *hist, uint32_t nb, uint32_t wbits, const uint32_t *partial, const uint32_t *base)
{
    const uint32_t bins = 1u << wbits, c = blockIdx.x, t0 = c * kRsChunk, t1 = min(nb, t0 + kRsChunk);
    for (uint32_t d = threadIdx.x; d < bins; d += 256u) {
        uint32_t run = partial[(size_t)c * bins + d] + base[d];
        uint32_t v[kRsChunk];
#pragma unroll
        for (int j = 0; j < kRsChunk; ++j) v[j] = t0 + j < t1 ? hist[(size_t)(t0 + j) * bins + d] : 0u;
#pragma unroll
        for (int j = 0; j < kRsChunk; ++j) {
            if (t0 + j < t1) hist[(size_t)(t0 + j) * bins + d] = run;
            run += v[j];
        }
    }
}

// stable scatter of one digit (bits [shift, shift + wbits) of the key, wbits <= WB); hist holds the offsets
template <int WB, bool RAW>
__global__ __launch_bounds__(kRsThreads) void rs_scatter(const RsSrc src, const uint4 *in, uint32_t n, uint32_t shift, uint32_t wbits, uint32_t nb,
                                                         const uint32_t *hist, uint4 *out, uint32_t *dtot)
{
    constexpr int BINS = 1 << WB, kWaves = kRsThreads / 64, kPerWave = kRsTile / kWaves, DPT = BINS / kRsThreads > 0 ? BINS / kRsThreads : 1;
    static_assert(BINS >= kRsThreads, "a thread per digit at least");
    __shared__ uint4 s_rec[kRsTile];
    __shared__ uint32_t s_gbase[BINS];                     // global position of the tile's first mark of each digit
    __shared__ uint16_t s_start[BINS];                     // where each digit starts inside the tile
    __shared__ uint16_t s_wloc[kWaves][BINS];              // per wave: marks of each digit so far; then the wave's offset
    __shared__ uint32_t s_wsum[kWaves];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, bins = 1u << wbits, dmask = bins - 1u;
    const uint32_t tile = rs_tile_of(blockIdx.x, nb);
    for (uint32_t d = tid; d < (uint32_t)BINS; d += kRsThreads) {
        s_gbase[d] = d < bins ? hist[(size_t)tile * bins + d] : 0u;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) s_wloc[w][d] = 0;
    }
    __syncthreads();
    const uint32_t base = tile * kRsTile, wbase = base + wave * kPerWave;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    uint4 rec[kRsItems];
    uint16_t lrank[kRsItems];
#pragma unroll
    for (int it = 0; it < kRsItems; ++it) {
        const uint32_t i = wbase + it * 64 + lane;
        rec[it] = i < n ? (RAW ? rs_make(src, i) : in[i]) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int it = 0; it < kRsItems; ++it) {
        const bool valid = wbase + it * 64 + lane < n;
        const uint32_t d = (uint32_t)(rs_key(rec[it], src.centre_bits, src.idx_bits) >> shift) & dmask;
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < WB; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bm = __ballot(bit);
            same &= bit ? bm : ~bm;
        }
        const uint32_t rank = (uint32_t)__popcll(same & lt);
        const uint32_t seen = s_wloc[wave][d];
        lrank[it] = (uint16_t)(seen + rank);
        __builtin_amdgcn_wave_barrier();                   // every lane has read the count before its leader bumps it
        if (valid && rank == 0) s_wloc[wave][d] = (uint16_t)(seen + (uint32_t)__popcll(same));
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // digit starts inside the tile (exclusive scan of the digit totals, DPT consecutive digits per thread) and, per wave, the
    // marks of the same digit in earlier waves
    {
        uint32_t tot[DPT], sum = 0;
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            const uint32_t d = tid * DPT + j;
            uint32_t t = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) {
                const uint32_t c = s_wloc[w][d];
                s_wloc[w][d] = (uint16_t)t;
                t += c;
            }
            tot[j] = t;
            sum += t;
        }
        uint32_t x = sum;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t y = __shfl_up(x, dd, 64);
            if ((int)lane >= dd) x += y;
        }
        if (lane == 63) s_wsum[wave] = x;
        __syncthreads();
        uint32_t run = x - sum;
        for (uint32_t w = 0; w < wave; ++w) run += s_wsum[w];
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            s_start[tid * DPT + j] = (uint16_t)run;
            run += tot[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < kRsItems; ++it) {
        if (wbase + it * 64 + lane < n) {
            const uint32_t d = (uint32_t)(rs_key(rec[it], src.centre_bits, src.idx_bits) >> shift) & dmask;
            s_rec[(uint32_t)s_start[d] + (uint32_t)s_wloc[wave][d] + (uint32_t)lrank[it]] = rec[it];
        }
    }
    __syncthreads();
    const uint32_t count = min((uint32_t)kRsTile, n - base);
#pragma unroll
    for (int it = 0; it < kRsItems; ++it) {
        const uint32_t q = it * kRsThreads + tid;
        if (q < count) {
            const uint4 r = s_rec[q];
            const uint32_t d = (uint32_t)(rs_key(r, src.centre_bits, src.idx_bits) >> shift) & dmask;
            out[s_gbase[d] + (q - (uint32_t)s_start[d])] = r;
        }
    }
    // rs_offsets_small is done with the totals: zero again for the next pass
    if (dtot && blockIdx.x < (uint32_t)kRsDtotCopies)
        for (uint32_t d = tid; d < bins; d += kRsThreads) dtot[blockIdx.x * bins + d] = 0;
}

#endif
