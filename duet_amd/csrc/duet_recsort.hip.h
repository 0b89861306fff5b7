// duet_recsort.hip.h -- stage A0's sort when the mark RECORD travels with the key (round 4).  Included inside
// duet_cluster.hip's anonymous namespace, after duet_prims.hip.h.
//
// The key-only sort of rounds 1-3 moved 8-byte keys (contig | type | centre, the mark index in the spare bits) and left the
// marks' records where the caller's order put them: the box test then fetched every 16-byte record through the sort
// permutation, a 128-byte line apiece on a shuffled input (2e7 marks: 517 us, 140 B of traffic per mark).  Here the record
// IS the sort element:
//     x = pos, y = span, z = read index (fused pipeline; else 0), w = (contig << type_bits | type) << idx_bits | mark index
// -- 16 bytes, and the key is a function of it (contig | type in w's high bits, centre = x + y / 2), so no key array exists
// at all.  Taken from 1.25 M marks on (below, every pass is launch-latency bound and 8-byte keys are the cheaper thing to move)
// when contig, type and mark index fit w's 32 bits (2e7 marks over a genome: 5 + 1 + 25); everything else keeps the key-only
// path.
//
//   rs_hist<RAW>      per 4096-mark tile, the digit counts of one pass; table laid out TILE-major (a tile's row is one
//                     coalesced store here and one coalesced load in rs_scatter -- with 1024-way digits a digit-major table
//                     costs a 4-byte access per digit and tile on both sides, as many sectors as the records themselves)
//   rs_offsets_small  (up to 1024 tiles) one launch: digit totals from rs_hist's atomics, a workgroup per 64 digits runs down
//                     its columns in 16 segments
//   rs_col_reduce     (beyond) column sums per chunk of 16 tiles -> rs_offsets_small over the chunks; rs_scatter adds the counts of the
//                     tiles in front of its own in the chunk itself
//   rs_scatter<WB>    stable scatter of one digit of up to WB bits: ballot-ranked per wave, the tile laid out digit-sorted in
//                     LDS and written from there (runs leave as whole lines); it also leaves the next pass's digit of every
//                     record beside it (rs_hist_dig reads 2 bytes per mark where rs_hist would read 16).  The first pass reads the caller's arrays and
//                     builds the records (RAW).  Tiles are dealt to workgroups so that the workgroups of one XCD hold a
//                     CONTIGUOUS range of tiles: the runs a digit receives from neighbouring tiles are neighbours in memory,
//                     and with 1024-way digits a run is 4 records = half a line -- the halves meet in that XCD's L2.
//   the low bits      rx_local / rx_big (duet_prims.hip.h) instantiated on records: groups of marks that agree in the globally
//                     sorted top bits are ordered by a rank count in LDS, the record stored to its place
//
// Passes: LSD over the key's top bits [lo, key_bits), lo chosen so that a group holds a few dozen marks (21 bits of a 2e7-mark
// genome's 34-bit key in an 11-bit and a 10-bit pass; 16 bits in two 8-bit passes where a 1 M-mark input is sent here), each
// pass stable, so equal keys keep their input order (rule 1 of oracle/cluster_oracle.c).
#ifndef DUET_RECSORT_HIP_H
#define DUET_RECSORT_HIP_H

constexpr int kRsTile = 4096;                          // marks per tile (rs_hist and rs_scatter)
constexpr int kRsThreads = 512, kRsItems = kRsTile / kRsThreads;
constexpr int kRsHistThreads = 1024;
constexpr int kRsMaxW = 11;                            // bits of one global digit, at most
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

// ... from the digits the pass before left beside the records.  256 threads, sixteen digits each (two 16-byte loads): with
// 1024 threads of four digits the launch was 78 k waves of a few dozen instructions -- 20 us for 40 MB at 2e7 marks
constexpr int kRsDigThreads = 256;
__global__ __launch_bounds__(kRsDigThreads) void rs_hist_dig(const uint16_t *dig, uint32_t n, uint32_t wbits, uint32_t *hist, uint32_t *dtot)
{
    __shared__ uint32_t s_h[1 << kRsMaxW];
    const uint32_t tid = threadIdx.x, bins = 1u << wbits, tile = blockIdx.x;
    for (uint32_t d = tid; d < bins; d += kRsDigThreads) s_h[d] = 0;
    __syncthreads();
    static_assert(kRsTile == kRsDigThreads * 16, "sixteen digits (two 16-byte loads) per thread");
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // (thread tid takes digits [8 tid, 8 tid + 8) of the tile's first and of its second half: a wave's load is one run of memory)
        const uint32_t i = tile * kRsTile + h * (kRsTile / 2) + tid * 8u;
        if (i + 7u < n) {
            const uint4 v = *reinterpret_cast<const uint4 *>(dig + i);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                atomicAdd(&s_h[w[k] & 0xFFFFu], 1u);
                atomicAdd(&s_h[w[k] >> 16], 1u);
            }
        } else {
            for (uint32_t j = i; j < n && j < i + 8u; ++j) atomicAdd(&s_h[dig[j]], 1u);
        }
    }
    __syncthreads();
    for (uint32_t d = tid; d < bins; d += kRsDigThreads) {
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
    // the digits' bases: exclusive scan of the digit totals, two consecutive digits per thread when there are 2048
    const uint32_t dp = bins > 1024u ? 2u : 1u;
    uint32_t tot[2] = {0, 0};
    for (uint32_t j = 0; j < dp; ++j) {
        const uint32_t d = tid * dp + j;
        if (d < bins) {
#pragma unroll
            for (int c = 0; c < kRsDtotCopies; ++c) tot[j] += dtot[c * bins + d];
        }
    }
    const uint32_t ex = rs_block_exscan_1024(tot[0] + tot[1], s_w);
    if (tid * dp < bins) s_base[tid * dp] = ex;
    if (dp == 2u && tid * 2u + 1u < bins) s_base[tid * 2u + 1u] = ex + tot[0];
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

// more tiles: column sums per chunk of kRsChunk tiles, one block for the chunks' running sums and the digits' bases; rs_scatter
// finds a tile's offsets from its chunk's (below)
// (the chunks' sums then go through rs_offsets_small as if they were tiles -- a few hundred of them: the digit totals by one
// atomic per chunk and digit)
__global__ __launch_bounds__(256) void rs_col_reduce(const uint32_t *hist, uint32_t nb, uint32_t wbits, uint32_t *partial /* [chunks][bins] */,
                                                     uint32_t *dtot /* [kRsDtotCopies][bins], zero on entry */)
{
    const uint32_t bins = 1u << wbits, c = blockIdx.x, t0 = c * kRsChunk, t1 = min(nb, t0 + kRsChunk);
    for (uint32_t d = threadIdx.x; d < bins; d += 256u) {
        uint32_t v[kRsChunk], s = 0;
#pragma unroll
        for (int j = 0; j < kRsChunk; ++j) v[j] = t0 + j < t1 ? hist[(size_t)(t0 + j) * bins + d] : 0u;
#pragma unroll
        for (int j = 0; j < kRsChunk; ++j) s += v[j];
        partial[(size_t)c * bins + d] = s;
        if (s) atomicAdd(&dtot[(c % kRsDtotCopies) * bins + d], s);
    }
}
// stable scatter of one digit (bits [shift, shift + wbits) of the key, wbits <= WB); hist holds the offsets.
// 512 threads = 8 waves, 8 marks per thread.  Each wave ranks its 512 marks on its own: ballot match inside the wave, then ONE
// returning LDS add per round by the digit's first lane into the wave's counter row (two 16-bit counters per word) -- the adds
// of a wave reach the LDS in program order, so the sixteen-deep read-bump-write chain of the key sort's scatter is gone and
// the rounds' LDS trips overlap; the other lanes of the digit fetch the leader's answer with a bpermute.  The tile is then
// laid out digit-sorted in LDS, 2048 records at a time IN THE COUNTERS' SPACE (they are dead by then: 36 KB per workgroup at
// 10-bit digits, four workgroups per CU), and written from there: consecutive lanes hold consecutive records of one digit run.
// dig_out (or null): the NEXT pass's digit of every record, at the record's new position -- that pass's histogram then reads
// two bytes per mark instead of sixteen.
template <int WB, bool RAW>
__global__ __launch_bounds__(kRsThreads) void rs_scatter(const RsSrc src, const uint4 *in, uint32_t n, uint32_t shift, uint32_t wbits, uint32_t nb,
                                                         const uint32_t *hist, uint4 *out, uint32_t *dtot, uint16_t *dig_out,
                                                         uint32_t next_shift, uint32_t next_mask,
                                                         const uint32_t *chunk_off /* [chunks][bins], or null: hist holds the tiles' offsets */)
{
    constexpr int BINS = 1 << WB, WORDS = BINS / 2, kWaves = kRsThreads / 64, kPerWave = kRsTile / kWaves, DPT = BINS / kRsThreads;
    constexpr int kStage = 2048, kRounds = kRsTile / kStage;
    static_assert(DPT >= 2 && DPT % 2 == 0, "a thread takes whole counter words");
    constexpr size_t kCntBytes = (size_t)kWaves * WORDS * 4 + (size_t)BINS * 2, kStageBytes = (size_t)kStage * 16;
    __shared__ __align__(16) unsigned char s_raw[kCntBytes > kStageBytes ? kCntBytes : kStageBytes];
    __shared__ uint32_t s_gadj[BINS];                      // global position of the tile's first mark of each digit - the digit's start inside the tile
    __shared__ uint32_t s_wsum[kWaves];
    uint32_t(*s_wloc)[WORDS] = reinterpret_cast<uint32_t(*)[WORDS]>(s_raw);              // per wave: marks of each digit so far; then the wave's offset
    uint16_t *s_start = reinterpret_cast<uint16_t *>(s_raw + (size_t)kWaves * WORDS * 4);  // where each digit starts inside the tile
    uint4 *s_rec = reinterpret_cast<uint4 *>(s_raw);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, bins = 1u << wbits, dmask = bins - 1u;
    const uint32_t tile = rs_tile_of(blockIdx.x, nb);
    if (chunk_off) {
        // more than kRsSmallTiles tiles: hist holds the tiles' COUNTS and chunk_off the offsets of every chunk of kRsChunk tiles;
        // the tile adds up the counts of the tiles in front of it in its chunk itself (rows its XCD's other workgroups read as well:
        // they come out of L2) -- a launch that wrote every tile's offsets back read and wrote the 40 MB table once more
        const uint32_t c = tile / kRsChunk, t0 = c * kRsChunk;
        for (uint32_t d = tid; d < (uint32_t)BINS; d += kRsThreads) {
            uint32_t v = 0;
            if (d < bins) {
                v = chunk_off[(size_t)c * bins + d];
#pragma unroll
                for (int j = 0; j < kRsChunk - 1; ++j) v += t0 + j < tile ? hist[(size_t)(t0 + j) * bins + d] : 0u;
            }
            s_gadj[d] = v;
        }
    } else {
        for (uint32_t d = tid; d < (uint32_t)BINS; d += kRsThreads) s_gadj[d] = d < bins ? hist[(size_t)tile * bins + d] : 0u;
    }
    for (uint32_t w = tid; w < (uint32_t)(kWaves * WORDS); w += kRsThreads) (&s_wloc[0][0])[w] = 0;
    const uint32_t base = tile * kRsTile, wbase = base + wave * kPerWave;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    uint4 rec[kRsItems];
    uint32_t q[kRsItems];
#pragma unroll
    for (int it = 0; it < kRsItems; ++it) {
        const uint32_t i = wbase + it * 64 + lane;
        rec[it] = i < n ? (RAW ? rs_make(src, i) : in[i]) : make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
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
        uint32_t seen = 0;
        if (valid && rank == 0) seen = (atomicAdd(&s_wloc[wave][d >> 1], (uint32_t)__popcll(same) << ((d & 1u) * 16u)) >> ((d & 1u) * 16u)) & 0xFFFFu;
        seen = (uint32_t)__shfl((int)seen, valid ? (int)__ffsll((long long)same) - 1 : (int)lane, 64);
        q[it] = seen + rank;
    }
    __syncthreads();
    // digit starts inside the tile (exclusive scan of the digit totals, DPT consecutive digits per thread) and, per wave, the
    // marks of the same digit in earlier waves
    {
        uint32_t tot[DPT], sum = 0;
#pragma unroll
        for (int j = 0; j < DPT; j += 2) {
            const uint32_t wd = (tid * DPT + j) >> 1;
            uint32_t t0 = 0, t1 = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) {
                const uint32_t c = s_wloc[w][wd];
                s_wloc[w][wd] = t0 | (t1 << 16);
                t0 += c & 0xFFFFu;
                t1 += c >> 16;
            }
            tot[j] = t0;
            tot[j + 1] = t1;
            sum += t0 + t1;
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
            const uint32_t d = tid * DPT + j;
            s_start[d] = (uint16_t)run;
            s_gadj[d] -= run;                              // (mod 2^32: the sum below comes out right)
            run += tot[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < kRsItems; ++it) {
        const uint32_t d = (uint32_t)(rs_key(rec[it], src.centre_bits, src.idx_bits) >> shift) & dmask;
        q[it] += (uint32_t)s_start[d] + ((s_wloc[wave][d >> 1] >> ((d & 1u) * 16u)) & 0xFFFFu);
    }
    __syncthreads();                                       // the counters are dead: their space takes the records
    const uint32_t count = min((uint32_t)kRsTile, n - base);
#pragma unroll
    for (int h = 0; h < kRounds; ++h) {
#pragma unroll
        for (int it = 0; it < kRsItems; ++it)
            if (wbase + it * 64 + lane < n && (q[it] / kStage) == (uint32_t)h) s_rec[q[it] % kStage] = rec[it];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kStage / kRsThreads; ++k) {
            const uint32_t pq = k * kRsThreads + tid, gq = h * kStage + pq;
            if (gq < count) {
                const uint4 r = s_rec[pq];
                const uint64_t key = rs_key(r, src.centre_bits, src.idx_bits);
                const uint32_t at = s_gadj[(uint32_t)(key >> shift) & dmask] + gq;
                out[at] = r;
                if (dig_out) dig_out[at] = (uint16_t)((uint32_t)(key >> next_shift) & next_mask);
            }
        }
        __syncthreads();
    }
    // rs_offsets_small is done with the totals: zero again for the next pass
    if (dtot && blockIdx.x < (uint32_t)kRsDtotCopies)
        for (uint32_t d = tid; d < bins; d += kRsThreads) dtot[blockIdx.x * bins + d] = 0;
}

#endif
