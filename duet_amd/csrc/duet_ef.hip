// duet_ef.hip -- gfx950 (MI355X) kernels and C ABI for Duet's step E/F.
//
// Pipeline (three launches on one stream, no host round trip):
//
//   ef_classify   one 256-thread workgroup per 256 consecutive candidates.  The workgroup streams
//                 its slice of mark_read with 16-byte loads, gathers the 8-byte read tag of every
//                 mark (the join of sv_phasing_fn.py:46-48) into LDS, then each thread walks ITS
//                 candidate's marks in list order out of LDS: filter (:189-190), PS-class (:191-194),
//                 seed PS (:199-203), class-0/1 vote (:74-84) and decision (:142-183); multi-PS
//                 candidates leave a 14-word summary of their first two voter groups.  Seeds leave the
//                 tile as compacted (candidate, PS) entries in candidate order -- repeats of the previous
//                 seed-bearing candidate dropped with ballots -- in the tile's own slots: no atomics, no
//                 hash set.
//   ef_seed_sort  one workgroup per contig: gathers the contig's entries in candidate order (an
//                 exclusive scan of per-thread counts), finishes local disorder with a few odd-even
//                 rounds, merges a few long ascending runs by rank or falls back to a bitonic network
//                 (LDS up to 15,360 seeds, HBM beyond), drops duplicates -> ascending `oneps` (:107).
//   ef_finalize   per candidate: contig drop (:209-210), nearest-PS (:106-111), and the class-2
//                 grouped vote (:85-105) + decision (:148-155) for the few multi-PS candidates.
//
// Everything order-dependent upstream (first qualifying mark, first-seen PS wins ties, last voter's
// PS) is reproduced by walking marks in list order; no result depends on atomic arrival order (the
// only LDS atomics hand out summary slots, whose numbering is not observable).
//
// Floating point: IEEE binary64, operations exactly as upstream, built with -ffp-contract=off.

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "duet_ef.h"
#include "duet_internal.h"

#pragma clang fp contract(off)

namespace {

#ifdef DUET_STAMPS
#define STAMP(k, i) do { if (threadIdx.x == 0 && p.stamps) p.stamps[((size_t)(k) * 65536 + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define STAMP(k, i) do { } while (0)
#endif

constexpr uint32_t kEmpty = 0xFFFFFFFFu;          // "no seed" / "no PS yet"
constexpr uint64_t kUntagged = ~0ull;             // LDS tag word of an absent mark
constexpr uint32_t kPcMax = DUET_PC_MAX;

// provisional code written by ef_classify into out_pred; values < 4 are already final
constexpr uint8_t kNeedNearest = 4;               // ps must become nearest(oneps, pos)
constexpr uint8_t kClass2 = 8;                    // multi-PS candidate with a group summary (out_ps = slot)
constexpr uint8_t kClass2Slow = 16;               // multi-PS candidate without one: ef_finalize re-gathers
constexpr uint8_t kDivZero = 32;                  // kept candidate with svread + refread == 0

constexpr int kCandPerBlock = 256;
constexpr int kChunk = 3072;                      // marks staged in LDS per pass: 24 KiB of tags, six workgroups per CU.  A 256-candidate tile of
                                                  // config 2 (2560 marks on average) still fits one pass; 4096 (four per CU): ef_classify 74 us instead
                                                  // of 61 us at 2e7 marks; 2048 (two passes per tile): -7 % at 2e7 marks but +38 % at 1e6
constexpr int kStageIt = kChunk / (kCandPerBlock * 4);   // 16-byte index loads per thread and pass
constexpr int kSmallK = 64;
constexpr int kSortThreads = 1024;
constexpr uint32_t kSortLds = 15360;              // seeds sorted in LDS (60 KiB)
constexpr uint32_t kLongCnt = 16;                   // ... when they hold more entries than this
constexpr uint32_t kLongTiles = 256;                // ef_seed_sort: tiles whose entries are copied by the whole workgroup
constexpr uint32_t kMaxRuns = 32;                  // ef_seed_sort merges up to this many ascending runs by rank
constexpr uint32_t kSeedTab = 4096;                // ef_seed_sort: slots of the hash set an unsorted seed list goes through first
constexpr uint32_t kFewRuns = 4;                   // ... right away when it finds no more descents than this
constexpr uint32_t kOneLds = 4096;                // seed array staged in LDS by ef_finalize (16 KiB)
constexpr uint32_t kC2Quota = 64;                 // group-summary slots of a classify tile's own; a tile with more multi-PS candidates reserves the rest from
                                                  // a shared pool (c2_slots).  Without a slot a candidate is left to ef_finalize's walk over its marks, ~10x
                                                  // slower: 2e6 marks over the 24 hg19 contigs -- one read per 8 kb, 37 % multi-PS candidates -- made
                                                  // ef_finalize 65 us with 32 slots per tile and no pool
constexpr int kC2Groups = 2;                      // voter groups kept per summary (first two seen)
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
    // hapread_ratio = allhap / deg (:112) is only ever compared with 0.75, which binary64 represents exactly: for integers
    // below 2^32 the rounded quotient is <= 0.75 exactly when 4 * allhap <= 3 * deg (a quotient above 0.75 exceeds it by at
    // least 1 / (4 * deg) > 2^-34, far more than the spacing of doubles there) -- one binary64 division less per candidate.
    // deg == 0 makes the quotient NaN, which fails both comparisons.
    const bool hp_le = deg != 0 && 4ull * v.allhap <= 3ull * deg, hp_gt = deg != 0 && !hp_le;
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
        const bool gate = (hp_le && diff <= 2400.0) || hp_gt;
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

// The same decision for classes 0 and 1 WITHOUT the four binary64 divisions (round 5: they were 80 of the 151 vector instructions of
// the decision, a fifth of what a wave of ef_classify issues) -- in integers, wherever integers provably give what the binary64
// expressions of :112-183 give; `need_fp` says where they do not, and the caller then takes decide() for that candidate.
//   * sv_ratio = fl(s / (s + r)) against 0.24, 0.9, 0.3, 0.45, 0.75 (and == 1): s + r < 2^26 here (the argument holds up to 2^33).  For a decimal c = p / q (q <= 25) and
//     its nearest double c_d:  s / (s + r) != c  =>  |s / (s + r) - c| >= 1 / (q (s + r)) > 4e-12, five orders of magnitude beyond
//     |c_d - c| + the quotient's rounding (< 3e-16): the same side.  s / (s + r) == c  =>  the correctly rounded quotient IS c_d (the
//     literal's value is RN(c) by definition): "<=" holds, as s q <= p (s + r) says.  Exact, always.  == 1.0 <=> r == 0.
//   * hapX_avgsc = fl(tX / hapX), diff = |a2 - a1| <= 2400.  One haplotype without votes: diff is ONE rounded quotient, rounding is
//     monotone and 2400 a double: fl(t / h) <= 2400 <=> t <= 2400 h.  Exact.  Both with votes: N = t2 h1 - t1 h2, B = 2400 h1 h2;
//     |N| != B  =>  ||a2 - a1| - 2400| >= 1 / (h1 h2) >= 2^-16 for h < 2^8, against < 2e-12 of accumulated rounding (quotients
//     <= 8100): the same side.  |N| == B with both sums positive, or numbers outside the ranges below: need_fp.
//   * totsc_ratio = fl(hi / lo) <= 9.72 = 243 / 25, lo > 0: 25 hi != 243 lo => |hi / lo - 9.72| >= 1 / (25 lo) > 1e-8 for lo < 2^21,
//     against ~1e-15: the same side; 25 hi == 243 lo => the quotient rounds to RN(9.72): "<=" holds.
//   * hap1_avgsc > 0 <=> hap1 > 0 and t1 > 0.
// tests: the 38 known answers of SURVEY 8c and the 20,000 boundary-biased random vectors (tests/golden/kat_*), which sit on exactly
// these thresholds, through the device (tests/test_gpu_parity.py::test_decision_known_answers_on_the_device) -- and every golden / fuzz case.
// (32-bit arithmetic throughout, products by 24-bit multiplies and shift-adds -- full-rate instructions; a first form with 64-bit
// products took 23 v_mad_u64_u32, quarter rate, and cost what the four divisions did.  Hence the narrower ranges: read counts below
// 2^26, PC sums below 2^21, haplotype votes below 2^8, degree below 2^28 -- every candidate a lane walks is inside them; outside: need_fp.)
__device__ __forceinline__ int decide01_int(int cls, const Vote &v, uint32_t deg, uint32_t svread, uint32_t refread, bool &need_fp)
{
    const uint64_t sr64 = (uint64_t)svread + (uint64_t)refread;                       // > 0: the caller has dealt with :123
    const uint32_t sr = (uint32_t)sr64, s = svread;
    const uint32_t t1 = (uint32_t)v.t1, t2 = (uint32_t)v.t2, h1 = v.hap1, h2 = v.hap2;
    need_fp = (sr64 >> 26) != 0 || ((v.t1 | v.t2) >> 21) != 0 || ((h1 | h2) >> 8) != 0 || (deg >> 28) != 0;
    if (cls == 0) return (refread == 0 && svread >= 4) ? 3 : 0;                       // :145-147 (exact for any counts)
    // sv_ratio <= p / q  <=>  s q <= p (s + r): the multiples by shift-adds (all below 2^31)
    const uint32_t s4 = s << 2, s10 = (s << 3) + (s << 1), s20 = s10 << 1, s25 = (s << 4) + (s << 3) + s;
    const uint32_t sr3 = (sr << 1) + sr, sr6 = sr3 << 1, sr9 = (sr << 3) + sr;
    const bool le024 = s25 <= sr6, le09 = s10 <= sr9, le03 = s10 <= sr3, le045 = s20 <= sr9, le075 = s4 <= sr3;
    const bool hp_le = deg != 0 && (v.allhap << 2) <= (deg << 1) + deg, hp_gt = deg != 0 && !hp_le;
    const uint32_t lo = t1 < t2 ? t1 : t2, hi = t1 < t2 ? t2 : t1;
    bool diff_le = true;                                                              // |a2 - a1| <= 2400 (:132, :161-166)
    if (h1 != 0 && h2 != 0) {
        const int32_t N = (int32_t)__umul24(t2, h1) - (int32_t)__umul24(t1, h2);      // (t < 2^21, h < 2^8: below 2^29)
        const uint32_t A = (uint32_t)(N < 0 ? -N : N), B = __umul24(__umul24(h1, h2), 2400u);
        need_fp = need_fp || (A == B && t1 != 0 && t2 != 0);
        diff_le = A <= B;
    } else if ((h1 | h2) != 0) {
        diff_le = (h1 != 0 ? t1 : t2) <= __umul24(h1 | h2, 2400u);
    }
    const bool gate = (hp_le && diff_le) || hp_gt;
    int pred = 0;
    if (lo == 0 && hi != 0) {                                                         // onehap_totsc != 0 (:159-167)
        if (le024) pred = 0;
        else if (le09) { if (gate) pred = (h1 != 0 && t1 != 0) ? 1 : 2; }
        else { if (gate) pred = 3; }
    } else {                                                                          // :168-182
        const bool ratio_le = lo == 0 || __umul24(hi, 25u) <= __umul24(lo, 243u);     // totsc_ratio <= 9.72 (0 when a sum is 0)
        if (le03) pred = 0;
        else if (le045) pred = refread > 10 ? 0 : (t1 > t2 ? 1 : 2);
        else if (le075) pred = ratio_le ? 3 : (t1 > t2 ? 1 : 2);
        else pred = 3;
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
    uint32_t n_reads;                 // entries of read_tag (an index at or beyond it: the mark has no tag)
    const uint64_t *untagged;         // the context's all-ones word (what such a mark gathers instead)
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
    uint64_t *blk_rec;                // [B][8] per classify block, one 64-byte line: {entry count, entries 0 .. 6} (entries 3 .. also in seed_ent:
                                      //        ef_seed_sort reads the first half, ef_finalize_own the line -- no dependent trip for a tile of up to seven)
    uint64_t *seed_ent;               // [B*256] entries 3.. of a block: (candidate << 32 | seed PS)
    uint32_t *onebuf;                 // [C+K] per contig at ctg_off[k]+k: {n, ascending distinct seed PS...}
    uint32_t one_cap;                 // C + K
    uint32_t *tmpbuf;                 // [C]   scratch of the out-of-LDS seed sort
    uint32_t *n_one;                  // [K]   length of contig k's seed array
    uint32_t *c2rec;                  // [c2_cap][kC2Words] group summaries of multi-PS candidates: kC2Quota slots per tile, then a shared pool
    uint32_t c2_fixed, c2_cap;        // B * kC2Quota; that + the pool's slots
    uint32_t *status;                 // [0] = div-zero flag, [1] = pool slots handed out in this run (ef_finalize zeroes it again)
    uint8_t *out_pred;
    uint32_t *out_ps;
    uint32_t n_small;                 // K when K <= kSmallK: the contig offsets then also ride in the kernel arguments
    uint32_t ctg_small[kSmallK + 1];  //   (scalar loads instead of a dependent HBM round trip)
    const uint32_t *dyn_c;            // device-planned runs (DYN kernels): the candidate count lives on the device, C is an upper bound
    const uint16_t *cand_contig;      // ... and, when the plan came from the candidates' contig column, that column (else null)
    uint32_t dbg;                     // diagnostic ablation bits (0 in production)
    uint32_t heavy_t;                 // ef_classify: candidates with more marks than this leave the lanes' serial walk for the wave-cooperative one
    unsigned long long *stamps;       // diagnostic build only: [kernel][block][8] wall-clock stamps
};

// ---------------------------------------------------------------------------------------------
// kernel 1: join + filter + PS-class + seeds + one-PS vote/decision + multi-PS group summaries
// ---------------------------------------------------------------------------------------------

// Per-candidate running state over its marks in list order.
struct CandState {
    // distinct PS among ALL tagged marks (no PC filter, Q7): first one seen, and whether another was seen (:191-194)
    uint32_t first_ps = kEmpty;
    bool multi = false;
    // voters (tagged, pc <= 8100) grouped by PS: first two groups in first-seen order (A, B); a third group sets
    // `more`.  Group A's PS is the PS of the FIRST voter = the seed (:199-203); with a single PS (class 1) group A
    // is the whole vote (:74-84) and its PS is also the last voter's PS (:77); with several PS it feeds :85-105.
    uint32_t ps_a = kEmpty, ps_b = kEmpty, n_a = 0, n_b = 0, nv = 0;
    uint32_t a1 = 0, a2 = 0, b1 = 0, b2 = 0;  // hap-1 / hap-2 counts per group
    uint32_t ta1 = 0, ta2 = 0, tb1 = 0, tb2 = 0;   // PC sums; A's are per-chunk partials folded into TA1/TA2
    uint64_t TA1 = 0, TA2 = 0;
    bool more = false;

    __device__ __forceinline__ uint32_t n_ps() const { return first_ps == kEmpty ? 0u : (multi ? 2u : 1u); }
    __device__ __forceinline__ uint32_t seed() const { return ps_a; }        // kEmpty when nobody voted
};

// thread walks marks [lo, hi) of its candidate; tags of mark m sit at s_tag[m - cs].
// An absent mark is the all-ones word: its "ps" is kEmpty and its "pc" 2^30-1, so it is neither a new PS nor a voter
// without a separate test.
__device__ __forceinline__ void consume_range_r4(CandState &st, const uint64_t *s_tag, uint32_t lo, uint32_t hi, uint32_t cs)
{
    for (uint32_t m = lo; m < hi; ++m) {
        const uint64_t tag = s_tag[m - cs];
        const uint32_t ps = (uint32_t)tag, w = (uint32_t)(tag >> 32);
        const uint32_t pc = w & 0x3FFFFFFFu, hap = w >> 30;
        const bool voter = pc <= kPcMax;
        st.first_ps = st.first_ps == kEmpty ? ps : st.first_ps;
        st.multi = st.multi || (ps != st.first_ps && ps != kEmpty);
        st.nv += voter;
        st.ps_a = (voter && st.ps_a == kEmpty) ? ps : st.ps_a;
        const bool in_a = voter && ps == st.ps_a;
        const bool rest = voter && !in_a;
        st.ps_b = (rest && st.ps_b == kEmpty) ? ps : st.ps_b;
        const bool in_b = rest && ps == st.ps_b;
        st.more = st.more || (rest && !in_b);
        const bool is1 = hap == 1, is2 = hap == 2;
        st.n_a += in_a; st.n_b += in_b;
        st.a1 += in_a && is1; st.a2 += in_a && is2;
        st.b1 += in_b && is1; st.b2 += in_b && is2;
        st.ta1 += (in_a && is1) ? pc : 0u; st.ta2 += (in_a && is2) ? pc : 0u;
        st.tb1 += (in_b && is1) ? pc : 0u; st.tb2 += (in_b && is2) ? pc : 0u;
    }
    st.TA1 += st.ta1; st.TA2 += st.ta2; st.ta1 = 0; st.ta2 = 0;
}

// The same walk with a shorter body (round 5; the counters put the vector units at 80 % busy and the walk at 18 x 35 of a wave's 1,012
// vector instructions, and every restructuring with control flow in it lost -- so: the same straight line, fewer instructions).
//   * haplotype 1 / 2 as ONE signed compare each on the tag's upper word (hap | pc): hap 1 <=> (int)w >= 0x40000000,
//     hap 2 <=> (int)w < -0x40000000 -- no shift;
//   * "one phase set or several" from a running minimum and maximum of the tagged marks' PS (an absent mark's PS is all ones: the
//     minimum ignores it, and the maximum runs over PS + 1, where it is zero) instead of first-PS bookkeeping: three instructions for four;
//   * every hap count rides in the top byte of its PC sum: a lane walks at most 255 marks per call (longer candidates are the
//     wavefront's), a PC that counts is at most 8100, so count < 2^8 and sum < 2^21 share a word and ONE select + ONE add replace an
//     add-with-carry, a select and an add -- four times;
//   * the marks are walked by LDS address, not by index + address.
// Exact: the same integers come out (tests force this walk, round 4's, and the wavefront's on every candidate).
// (the three vector instructions the walk is made of besides compares, min / max and adds -- spelled out, with the predicate as the
// lane mask it is: written in C++ the compiler issues a second compare for every negated predicate and a select + add for every
// "counter += predicate")
__device__ __forceinline__ uint32_t sel0(unsigned long long mask, uint32_t v)                  // mask ? v : 0
{
    uint32_t d;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(d) : "v"(v), "s"(mask));
    return d;
}
__device__ __forceinline__ uint32_t selv(unsigned long long mask, uint32_t a, uint32_t b)      // mask ? b : a
{
    uint32_t d;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(mask));
    return d;
}
__device__ __forceinline__ uint32_t inc(uint32_t x, unsigned long long mask)                   // x + (this lane's bit of mask)
{
    uint32_t d;
    asm("v_addc_co_u32_e64 %0, vcc, 0, %1, %2" : "=v"(d) : "v"(x), "s"(mask) : "vcc");
    return d;
}

__device__ __forceinline__ void consume_range(CandState &st, const uint64_t *s_tag, uint32_t lo, uint32_t hi, uint32_t cs)
{
    uint32_t mn = st.first_ps;                                     // min over the tagged marks' PS (kEmpty: none yet)
    uint32_t mx = st.first_ps == kEmpty ? 0u : st.first_ps + 1u;   // max over PS + 1
    uint32_t pa1 = 0, pa2 = 0, pb1 = 0, pb2 = 0;                   // count << 24 | PC sum
    // per-lane flags as the lane masks they are (scalar registers): group A / B has its PS; a voter outside both was seen
    unsigned long long has_a = __ballot(st.ps_a != kEmpty), has_b = __ballot(st.ps_b != kEmpty), more = 0;
    uint32_t nv = 0, nb = 0, ps_a = st.ps_a, ps_b = st.ps_b;
    // (a candidate that lies in front of or behind this pass has lo >= hi: an empty range, not a wrapped one)
    const uint64_t *q = s_tag + (lo < hi ? lo - cs : 0u), *qe = q + (lo < hi ? hi - lo : 0u);
    for (; q < qe; ++q) {
        const uint64_t tag = *q;
        const uint32_t ps = (uint32_t)tag, w = (uint32_t)(tag >> 32);
        const uint32_t pc = w & 0x3FFFFFFFu;
        const unsigned long long vm = __ballot(pc <= kPcMax);                                  // voters
        const unsigned long long m1 = __ballot((int32_t)w >= 0x40000000), m2 = __ballot((int32_t)w < -0x40000000);
        mn = min(mn, ps);
        mx = max(mx, ps + 1u);
        nv = inc(nv, vm);
        ps_a = selv(vm & ~has_a, ps_a, ps);
        has_a |= vm;
        const unsigned long long ina = vm & __ballot(ps == ps_a), rest = vm ^ ina;             // (ina is part of vm)
        ps_b = selv(rest & ~has_b, ps_b, ps);
        has_b |= rest;
        const unsigned long long inb = rest & __ballot(ps == ps_b);
        more |= rest ^ inb;                                                                    // (inb is part of rest)
        nb = inc(nb, inb);
        const uint32_t pcx = pc | (1u << 24);
        pa1 += sel0(ina & m1, pcx); pa2 += sel0(ina & m2, pcx);
        pb1 += sel0(inb & m1, pcx); pb2 += sel0(inb & m2, pcx);
    }
    const uint32_t lane = threadIdx.x & 63u;
    st.ps_a = ps_a; st.ps_b = ps_b;
    st.more = st.more || ((more >> lane) & 1ull) != 0ull;
    st.multi = st.multi || (mn != kEmpty && mn + 1u != mx);
    st.first_ps = mn;                                              // (any tagged mark's PS serves while there is one phase set)
    // group A's voters are not counted per mark: they are the voters that are not B's -- unless a third group exists, and then
    // nobody reads n_a (decide_store: such a candidate gets no summary; classes 0 and 1 never look at it)
    st.nv += nv; st.n_b += nb;
    st.n_a = st.more ? (st.ps_a != kEmpty ? 1u : 0u) : st.nv - st.n_b;
    st.a1 += pa1 >> 24; st.a2 += pa2 >> 24; st.b1 += pb1 >> 24; st.b2 += pb2 >> 24;
    st.TA1 += pa1 & 0xFFFFFFu; st.TA2 += pa2 & 0xFFFFFFu;
    st.tb1 += pb1 & 0xFFFFFFu; st.tb2 += pb2 & 0xFFFFFFu;
}

// Sum of one value per lane over the wavefront (row rotations, then the two row broadcasts of gfx9: six adds with a DPP operand,
// nothing through LDS); the total comes back on the scalar unit.
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);     // quad_perm:[1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);     // quad_perm:[2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false);    // row_ror:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);    // row_ror:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);    // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);    // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// the value of lane (lane ^ J): inside a row of 16 lanes as DPP operands (J = 4: the two directions by bank mask), across rows
// through the LDS crossbar
template <uint32_t J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v)
{
    if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);       // quad_perm:[1,0,3,2]
    else if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);  // quad_perm:[2,3,0,1]
    else if constexpr (J == 4) {
        const int r = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xF, 0x5, false);                        // row_shl:4 into banks 0, 2
        return (uint32_t)__builtin_amdgcn_update_dpp(r, (int)v, 0x114, 0xF, 0xA, false);                          // row_shr:4 into banks 1, 3
    } else if constexpr (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false); // row_ror:8
    else return (uint32_t)__shfl_xor((int)v, (int)J, 64);
}

// 64 values, one per lane, into ascending order over the lanes (bitonic network: 21 exchanges, 18 of them inside rows)
template <uint32_t KK, uint32_t J>
__device__ __forceinline__ uint32_t bitonic_step(uint32_t v, uint32_t lane)
{
    const uint32_t o = lane_xor<J>(v);
    const bool up = KK == 64u || (lane & KK) == 0u, low = (lane & J) == 0u;
    v = (low == up) ? min(v, o) : max(v, o);
    if constexpr (J > 1) return bitonic_step<KK, J / 2>(v, lane);
    else return v;
}
__device__ __forceinline__ uint32_t wave_sort64(uint32_t v, uint32_t lane)
{
    v = bitonic_step<2, 1>(v, lane);
    v = bitonic_step<4, 2>(v, lane);
    v = bitonic_step<8, 4>(v, lane);
    v = bitonic_step<16, 8>(v, lane);
    v = bitonic_step<32, 16>(v, lane);
    return bitonic_step<64, 32>(v, lane);
}

// The wave-cooperative walk (north_star's "wavefront-reduced per-cluster haplotype vote"): the WHOLE wavefront takes marks
// [lo, hi) of the candidate that lane `h` owns, 64 marks per step, and lane h's state advances exactly as consume_range would
// have advanced it.  Every order rule of the serial walk is a "first lane with ..." here: the first tagged mark's PS (:191-194),
// the first voter's PS = group A = the seed (:199-203), the first voter outside A = group B (:85-105 keeps first-seen order),
// anything else a third group; counts are population counts of lane masks (a compare IS the ballot on this machine) and only
// the four PC sums go through a reduction -- exact integer work throughout.  lo, hi, h are wave-uniform.
// A serial walk costs a wave as many dependent LDS round trips as its LARGEST candidate has marks; candidates with a tail of
// sizes (stage A0's own output, any real caller's) leave 63 lanes idle behind one.  This path bounds that chain.
__device__ __forceinline__ void consume_heavy(CandState &st, const uint64_t *s_tag, uint32_t lo, uint32_t hi, uint32_t cs, uint32_t lane,
                                              uint32_t h)
{
    uint32_t first_ps = (uint32_t)__builtin_amdgcn_readlane((int)st.first_ps, h);
    uint32_t ps_a = (uint32_t)__builtin_amdgcn_readlane((int)st.ps_a, h);
    uint32_t ps_b = (uint32_t)__builtin_amdgcn_readlane((int)st.ps_b, h);
    bool multi = false, more = false;
    uint32_t nv = 0, n_a = 0, n_b = 0, a1 = 0, a2 = 0, b1 = 0, b2 = 0;          // wave-uniform
    uint32_t ta1 = 0, ta2 = 0, tb1 = 0, tb2 = 0;                                // per lane
    for (uint32_t m0 = lo; m0 < hi; m0 += 64u) {
        const uint32_t m = m0 + lane;
        const uint64_t tag = m < hi ? s_tag[m - cs] : kUntagged;
        const uint32_t ps = (uint32_t)tag, w = (uint32_t)(tag >> 32);
        const uint32_t pc = w & 0x3FFFFFFFu, hap = w >> 30;
        const bool voter = pc <= kPcMax;                                        // (an absent mark: pc = 2^30 - 1, ps = kEmpty)
        const unsigned long long tmask = __ballot(ps != kEmpty);
        if (first_ps == kEmpty && tmask) first_ps = (uint32_t)__builtin_amdgcn_readlane((int)ps, (uint32_t)__ffsll((long long)tmask) - 1u);
        multi = multi || __ballot(ps != first_ps && ps != kEmpty) != 0ull;
        const unsigned long long vmask = __ballot(voter);
        if (ps_a == kEmpty && vmask) ps_a = (uint32_t)__builtin_amdgcn_readlane((int)ps, (uint32_t)__ffsll((long long)vmask) - 1u);
        const bool in_a = voter && ps == ps_a;
        const unsigned long long amask = __ballot(in_a);
        const unsigned long long rmask = vmask & ~amask;
        if (ps_b == kEmpty && rmask) ps_b = (uint32_t)__builtin_amdgcn_readlane((int)ps, (uint32_t)__ffsll((long long)rmask) - 1u);
        const bool in_b = voter && !in_a && ps == ps_b;
        const unsigned long long bmask = __ballot(in_b);
        more = more || (rmask & ~bmask) != 0ull;
        const bool is1 = hap == 1, is2 = hap == 2;
        const unsigned long long m1 = __ballot(is1), m2 = __ballot(is2);
        nv += (uint32_t)__popcll(vmask);
        n_a += (uint32_t)__popcll(amask); n_b += (uint32_t)__popcll(bmask);
        a1 += (uint32_t)__popcll(amask & m1); a2 += (uint32_t)__popcll(amask & m2);
        b1 += (uint32_t)__popcll(bmask & m1); b2 += (uint32_t)__popcll(bmask & m2);
        ta1 += (in_a && is1) ? pc : 0u; ta2 += (in_a && is2) ? pc : 0u;
        tb1 += (in_b && is1) ? pc : 0u; tb2 += (in_b && is2) ? pc : 0u;
    }
    // (a chunk holds 3072 marks of at most 8100 each: the sums fit 32 bits, as the serial walk's per-chunk partials do)
    const uint32_t s_a1 = a1 ? wave_sum(ta1) : 0u, s_a2 = a2 ? wave_sum(ta2) : 0u;
    const uint32_t s_b1 = b1 ? wave_sum(tb1) : 0u, s_b2 = b2 ? wave_sum(tb2) : 0u;
    if (lane == h) {
        st.first_ps = first_ps; st.multi = st.multi || multi;
        st.ps_a = ps_a; st.ps_b = ps_b; st.more = st.more || more;
        st.nv += nv; st.n_a += n_a; st.n_b += n_b;
        st.a1 += a1; st.a2 += a2; st.b1 += b1; st.b2 += b2;
        st.TA1 += s_a1; st.TA2 += s_a2; st.tb1 += s_b1; st.tb2 += s_b2;
    }
}

// a tile's 64-byte record: [b] = its first half {count, entries 0 .. 2}, second(b) = entries 3 .. 6
struct TileRecs {
    const ulonglong4 *p;
    __device__ __forceinline__ ulonglong4 operator[](uint32_t b) const { return p[2 * (size_t)b]; }
    __device__ __forceinline__ ulonglong4 second(uint32_t b) const { return p[2 * (size_t)b + 1]; }
};

struct TileShared {
    uint32_t seed[kCandPerBlock];
    uint32_t wcnt[kCandPerBlock / 64], wlast[kCandPerBlock / 64], wclean[kCandPerBlock / 64];
    uint32_t c2n;                             // summary slots handed out in this tile (reset by the caller)
    uint32_t c2base;                          // first pool slot of a tile that needs more than its own kC2Quota
};

// a multi-PS candidate's summary: the first two voter groups
__device__ __forceinline__ void store_summary(const Params &p, uint32_t slot, const CandState &st)
{
    uint32_t *rec = p.c2rec + (size_t)slot * kC2Words;
    const uint32_t ng = (st.n_a != 0) + (st.n_b != 0);
    rec[0] = st.nv; rec[1] = ng;
    // ps_a / ps_b are only meaningful when n_a / n_b > 0 (ng says how many groups exist)
    rec[2] = st.ps_a; rec[3] = st.n_a; rec[4] = st.a1; rec[5] = st.a2;
    rec[6] = (uint32_t)st.TA1; rec[7] = (uint32_t)st.TA2;
    rec[8] = st.ps_b; rec[9] = st.n_b; rec[10] = st.b1; rec[11] = st.b2; rec[12] = st.tb1; rec[13] = st.tb2;
}

// decision + outputs of ONE candidate (the one this thread walked); returns its seed entry or kEmpty.  c2_rank: the
// candidate's number among the tile's multi-PS candidates with a summary (kEmpty: it is none) -- the first kC2Quota of them
// have their slot here, the caller finds the others one in the pool.
__device__ __forceinline__ uint32_t decide_store(const Params &p, TileShared &sh, uint32_t tile, bool live, uint32_t c,
                                                 bool active, bool divzero, const CandState &st, uint32_t deg,
                                                 uint32_t svread, uint32_t refread, uint32_t &c2_rank)
{
    uint8_t code = 0;
    uint32_t ps_out = 0;
    bool want_seed = false;
    // multi-PS candidates get a summary slot (first two voter groups) unless a third group exists or
    // the PC sums could leave 32 bits; those few are re-gathered by ef_finalize
    const uint32_t n_ps = st.n_ps();
    const bool c2 = active && n_ps == 2;
    const bool c2_fast = c2 && !st.more && deg < 500000u;
    const uint32_t rank = c2_fast ? atomicAdd(&sh.c2n, 1u) : kEmpty;
    c2_rank = rank;
    if (live) {
        // the seed sets are built from every kept class-1 candidate BEFORE any decision is taken (:195-203), so a candidate
        // whose decision would divide by zero (:123) still contributes its seed -- and raises only if its contig then has one
        want_seed = (active || divzero) && (n_ps == 1) && st.seed() != kEmpty;                     // :198-203
        if (divzero) {
            code = kDivZero;
        } else if (active) {
            if (c2) {
                if (rank < kC2Quota) {
                    const uint32_t slot = tile * kC2Quota + rank;
                    store_summary(p, slot, st);
                    code = kClass2;
                    ps_out = slot;
                } else {
                    code = kClass2Slow;                                   // (until the caller finds it a pool slot)
                }
            } else {
                Vote v;
                v.hap1 = st.a1; v.hap2 = st.a2; v.hap0 = 0;
                v.allhap = st.a1 + st.a2;
                v.t1 = st.TA1; v.t2 = st.TA2;
                // (in integers where integers provably agree with :112-183's binary64; the few candidates for which they might
                // not -- an exact tie of two rounded quotients, sums beyond 2^31 -- take decide(): a wave without such a
                // candidate never issues its four divisions)
                bool need_fp = false;
                int pred = (p.dbg & DUET_DBG_EF_FP_DECIDE) ? 0 : decide01_int((int)n_ps, v, deg, svread, refread, need_fp);
                need_fp = need_fp || (p.dbg & DUET_DBG_EF_FP_DECIDE) != 0;
                if (need_fp) pred = decide((int)n_ps, v, deg, svread, refread);
                code = (p.dbg & 2) ? 0 : (uint8_t)pred;
                ps_out = st.ps_a == kEmpty ? 0u : st.ps_a;                // class 1: the single PS = last voter's PS (:77)
                if (n_ps == 0 || (st.a1 == 0 && st.a2 == 0)) code |= kNeedNearest;                 // :106
            }
        }
        p.out_pred[c] = code;
        p.out_ps[c] = ps_out;
    }
    if (p.dbg & 1) want_seed = false;
    return want_seed ? st.seed() : kEmpty;
}

// Seed entries of one tile, in candidate order (thread t speaks for candidate c0 + t; sh.seed[] must hold every
// candidate's seed or kEmpty and be visible).  A seed equal to the seed of the previous seed-bearing candidate is
// dropped, unless a contig starts in between (seed sets are per contig).  Only the set matters downstream, so
// dropping duplicates early just shortens ef_seed_sort's input.  All 256 threads call it (it synchronises).
__device__ __forceinline__ void seeds_tile(const Params &p, TileShared &sh, uint32_t tile, uint32_t tid, uint32_t c,
                                           uint32_t is_start)
{
    const uint32_t seed = sh.seed[tid];
    const bool want_seed = seed != kEmpty;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const unsigned long long wmask = __ballot(want_seed);
    const unsigned long long smask = __ballot(is_start != 0);
    const unsigned long long upto = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);    // lanes 0..lane
    if (lane == 0) {
        if (wmask) {
            const uint32_t li = 63u - (uint32_t)__clzll(wmask);
            sh.wlast[wave] = sh.seed[wave * 64 + li];
            sh.wclean[wave] = (li == 63 || (smask >> (li + 1)) == 0) ? 1u : 0u;   // no contig start after it
        } else {
            sh.wlast[wave] = kEmpty;
            sh.wclean[wave] = 0;
        }
    }
    __syncthreads();
    bool keep = want_seed;
    if (want_seed) {
        const unsigned long long lower = wmask & (upto >> 1);                   // seed lanes below this one
        if (lower) {
            const uint32_t pl = 63u - (uint32_t)__clzll(lower);
            const unsigned long long between = smask & upto & ~((1ull << (pl + 1)) - 1ull);   // starts in (pl, lane]
            keep = !(between == 0 && sh.seed[wave * 64 + pl] == seed);
        } else if (wave > 0) {
            keep = !((smask & upto) == 0 && sh.wclean[wave - 1] && sh.wlast[wave - 1] == seed);
        }
    }
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) sh.wcnt[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kCandPerBlock / 64; ++w) {
        before += w < wave ? sh.wcnt[w] : 0u;
        total += sh.wcnt[w];
    }
    if (keep) {
        // the first three entries sit beside the count (one 32-byte record per tile: ef_seed_sort reads count and
        // entries in a single round trip); the rest go to the tile's overflow slots
        const uint32_t at = before + (uint32_t)__popcll(mask & (upto >> 1));
        const uint64_t e = ((uint64_t)c << 32) | seed;
        if (at < 7) p.blk_rec[(size_t)tile * 8 + 1 + at] = e;
        if (at >= 3) p.seed_ent[(size_t)tile * kCandPerBlock + at] = e;
    }
    if (tid == 0) p.blk_rec[(size_t)tile * 8] = total;
}

// ---- staging pieces: 4 * kStageIt marks per thread and pass, kStageIt x (16-byte index load -> 4 tag gathers) ----
// Index loads are branch-free and their values are not touched here: a conditional load, or any use of the
// loaded registers, makes the compiler wait on the spot, which would serialise the four loads and break the
// software pipeline.  The mark array is 16-byte aligned on this path, so an aligned 16-byte load that starts
// inside it stays inside its last 16-byte granule even when M is not a multiple of 4; lanes past m_end read
// a harmless in-range address.  stage_gather masks both cases when it finally consumes the indices.
__device__ __forceinline__ void stage_load_marks(const Params &p, uint4 (&r)[kStageIt], uint32_t cs, uint32_t m_end, uint32_t tid)
{
#pragma unroll
    for (int it = 0; it < kStageIt; ++it) {
        const uint32_t m = cs + 4u * (tid + it * kCandPerBlock);
        r[it] = *reinterpret_cast<const uint4 *>(p.mark_read + (m < m_end ? m : cs));
    }
}

// Tag gathers, branch-free for the same reason -- and with nothing left to do once the tags arrive: a mark without a tag (an absent
// name: index all ones) gathers the context's own "untagged" word instead of a table entry, by ADDRESS (one compare of the index
// with the table's size, the address, two selects), so that what comes back goes to LDS as it is.  (Round 4 gathered entry 0 for such
// marks, kept twelve validity bits packed in a word and unpacked them again for a select per tag half: 8.5 vector instructions per
// mark where this takes 4 -- the counters put this kernel at 80 % vector issue.)  The compare also covers the indices a 16-byte load
// picks up behind the last mark of the array (anything below the table's size is a harmless gather into an LDS slot nobody reads).
__device__ __forceinline__ void stage_gather(const Params &p, uint64_t (&t)[4 * kStageIt], const uint4 (&r)[kStageIt])
{
#pragma unroll
    for (int it = 0; it < kStageIt; ++it) {
        const uint32_t idx[4] = {r[it].x, r[it].y, r[it].z, r[it].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t *a = idx[j] < p.n_reads ? p.read_tag + idx[j] : p.untagged;
            t[4 * it + j] = *a;
        }
    }
}

__device__ __forceinline__ void stage_store(uint64_t *s_tag, const uint64_t (&t)[4 * kStageIt], uint32_t cs, uint32_t m_end, uint32_t tid)
{
#pragma unroll
    for (int it = 0; it < kStageIt; ++it) {
        const uint32_t i = 4u * (tid + it * kCandPerBlock);
        if (cs + i < m_end) {
#pragma unroll
            for (int j = 0; j < 4; ++j) s_tag[i + j] = t[4 * it + j];
        }
    }
}

// One tile (256 candidates) per workgroup.  (A persistent variant that software-pipelined the two staging
// round trips of tile i+1 / i+2 under the LDS walk of tile i was measured at 2e7 marks and was NOT faster --
// 80 us vs 72 us: with four workgroups per CU in different phases the staging latency is already hidden and
// the kernel is bound by the sum of VALU issue (consume + fp64 decision) and memory time; see DESIGN.md.)
// (six waves per SIMD: what the 26.7 KB of LDS allow -- the device-planned variant came out at 83 registers, five waves, without the bound)
template <bool VEC, bool DYN = false>
__global__ __launch_bounds__(kCandPerBlock, 6) void ef_classify(const Params p)
{
    __shared__ uint64_t s_tag[kChunk];
    __shared__ uint32_t s_off[kCandPerBlock + 1];
    __shared__ TileShared sh;

    const uint32_t tid = threadIdx.x;
    STAMP(0, 0);
    const uint32_t n_cands = DYN ? *p.dyn_c : p.C;
    // device-planned runs know only an upper bound of the candidate count on the host: their grid is a fraction of that
    // bound and strides over the real tiles (a grid of the full bound is ~10x too many workgroups on SV-like data, and
    // starting the empty ones cost the fused pipeline 9 of 18 us at 1 M marks)
    for (uint32_t tile = blockIdx.x; !DYN || tile * (uint32_t)kCandPerBlock < n_cands; tile += gridDim.x) {
        const uint32_t c0 = tile * kCandPerBlock;
        const uint32_t nc = min((uint32_t)kCandPerBlock, n_cands - c0);
        for (uint32_t i = tid; i <= nc; i += kCandPerBlock) s_off[i] = p.cand_off[c0 + i];
        if (tid == 0) sh.c2n = 0;

        // candidate scalars, coalesced (candidate c0 + tid)
        uint32_t svlen = 0, svread = 0, refread = 0, gt_ok = 0, is_start = 0;
        if (tid < nc) {
            svlen = p.cand_svlen[c0 + tid];
            svread = p.cand_svread[c0 + tid];
            refread = p.cand_refread[c0 + tid];
            gt_ok = p.cand_gt_ok[c0 + tid];
            if (DYN) {
                // device-planned runs have no start flags.  With the candidates' contig column at hand a start is where the column
                // changes (two coalesced loads beside the other scalars, no dependent chain); else the contig of this candidate is
                // walked from the contig of the tile's first one
                const uint32_t c = c0 + tid;
                if (p.cand_contig) {
                    is_start = (c == 0 || p.cand_contig[c] != p.cand_contig[c - 1]) ? 1u : 0u;
                } else {
                    uint32_t k = p.blk_ctg[tile];
                    while (c >= p.ctg_off[k + 1]) ++k;
                    is_start = c == p.ctg_off[k] ? 1u : 0u;
                }
            } else {
                is_start = p.ctg_start[c0 + tid];
            }
        }
        const bool kept = tid < nc && svlen >= p.svlen_thres && svread >= p.suppread_thres && gt_ok != 0;     // :189-190
        const bool divzero = kept && ((uint64_t)svread + (uint64_t)refread == 0);
        __syncthreads();
        STAMP(0, 1);
        const uint32_t m_begin = s_off[0], m_end = s_off[nc];

        // thread t walks candidate c0 + t.  (Dealing a tile's candidates to the lanes in descending order of their mark count -- so
        // that the four waves loop 18/14/10/6 times instead of 4 x 18 -- was measured in rounds 1 and 4, and in round 5 with the wave
        // that takes the long quarter rotating from tile to tile (so that no SIMD of a CU gets all the long walks): NOT faster,
        // profiles/history/r05_ef_classify_dealt_by_size_with_rotation_REJECTED.txt.  The kernel is bound by the latency of a tile --
        // three dependent round trips before its walk can start -- at five tiles per CU, not by the walk's issue slots.)
        const uint32_t j = tid;
        const bool live = j < nc;
        const bool active = kept && !divzero;
        const uint32_t my_b = live ? s_off[j] : 0, my_e = live ? s_off[j + 1] : 0;
        CandState st;

        const uint32_t base = VEC ? (m_begin & ~3u) : m_begin;
        for (uint32_t cs = base; cs < m_end; cs += kChunk) {
            // ---- stage: LDS[i] = tag of mark cs+i -----------------------------------------------
            if (VEC) {
                uint4 r[kStageIt];
                uint64_t t[4 * kStageIt];
                stage_load_marks(p, r, cs, m_end, tid);
                stage_gather(p, t, r);
                stage_store(s_tag, t, cs, m_end, tid);
            } else {
                for (uint32_t i = tid; i < kChunk && cs + i < m_end; i += kCandPerBlock) {
                    const uint32_t r = p.mark_read[cs + i];
                    s_tag[i] = r < p.n_reads ? p.read_tag[r] : kUntagged;        // (as stage_gather: an index beyond the table has no tag)
                }
            }
            __syncthreads();
            STAMP(0, 2);
            // ---- consume: each thread walks its candidate's part of this chunk ----------------------
            uint32_t lo = max(my_b, cs);
            const uint32_t hi = min(my_e, cs + (uint32_t)kChunk);
            if (!kept || (p.dbg & 4)) lo = hi;
            // candidates with more marks than heavy_t: the wavefront walks them together, 64 marks per step, after the lanes
            // have walked the others side by side
            const bool heavy = lo < hi && my_e - my_b > min(p.heavy_t, 255u);      // (the lane walk packs counts into bytes)
            unsigned long long hm = __ballot(heavy);
            if (p.dbg & DUET_DBG_EF_WALK_R4) consume_range_r4(st, s_tag, heavy ? hi : lo, hi, cs);
            else consume_range(st, s_tag, heavy ? hi : lo, hi, cs);
            while (hm) {
                const uint32_t h = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)__ffsll((long long)hm) - 1u));
                hm &= hm - 1ull;
                const uint32_t lo_h = (uint32_t)__builtin_amdgcn_readlane((int)lo, h), hi_h = (uint32_t)__builtin_amdgcn_readlane((int)hi, h);
                consume_heavy(st, s_tag, lo_h, hi_h, cs, tid & 63u, h);
            }
            STAMP(0, 3);
            __syncthreads();
        }
        STAMP(0, 4);
        uint32_t c2_rank;
        const uint32_t seed = decide_store(p, sh, tile, live, c0 + j, active, divzero, st, my_e - my_b, svread, refread, c2_rank);
        sh.seed[j] = seed;
        __syncthreads();
        if (sh.c2n > kC2Quota) {
            // more multi-PS candidates than the tile has slots (fragmented phasing: short phase sets against long reads): the
            // rest get theirs from the pool, one reservation per tile.  Only such tiles pay the atomic's round trip.
            if (tid == 0) sh.c2base = atomicAdd(&p.status[1], sh.c2n - kC2Quota);
            __syncthreads();
            if (live && c2_rank != kEmpty && c2_rank >= kC2Quota) {
                const uint32_t at = sh.c2base + (c2_rank - kC2Quota);
                if (at < p.c2_cap - p.c2_fixed) {
                    store_summary(p, p.c2_fixed + at, st);
                    p.out_pred[c0 + j] = kClass2;
                    p.out_ps[c0 + j] = p.c2_fixed + at;
                }
            }
        }
        seeds_tile(p, sh, tile, tid, c0 + tid, is_start);
        STAMP(0, 6);
        if (!DYN) break;
        __syncthreads();                                           // the tile's LDS is reused
    }
}

// ---------------------------------------------------------------------------------------------
// kernel 2: per contig, distinct seeds -> ascending array (np.sort(list(oneps_set)), :107)
// ---------------------------------------------------------------------------------------------

// All-ascending bitonic network ("flip" form): every compare-exchange puts the smaller key at the
// lower index, so indices >= n behave as +inf padding without being stored.
__device__ void bitonic_sort(uint32_t *a, uint32_t n, uint32_t tid, uint32_t nthreads)
{
    uint32_t lN = 0;
    while ((1u << lN) < n) ++lN;
    const uint32_t pairs = (1u << lN) >> 1;
    for (uint32_t lk = 1; lk <= lN; ++lk) {
        const uint32_t k = 1u << lk, half = k >> 1;
        for (uint32_t i = tid; i < pairs; i += nthreads) {
            const uint32_t base = (i >> (lk - 1)) << lk, off = i & (half - 1);
            const uint32_t x = base + off, y = base + (k - 1 - off);
            if (y < n) {
                const uint32_t ax = a[x], ay = a[y];
                if (ax > ay) { a[x] = ay; a[y] = ax; }
            }
        }
        __syncthreads();
        for (int lj = (int)lk - 2; lj >= 0; --lj) {
            const uint32_t j = 1u << lj;
            for (uint32_t i = tid; i < pairs; i += nthreads) {
                const uint32_t x = ((i >> lj) << (lj + 1)) + (i & (j - 1)), y = x + j;
                if (y < n) {
                    const uint32_t ax = a[x], ay = a[y];
                    if (ax > ay) { a[x] = ay; a[y] = ax; }
                }
            }
            __syncthreads();
        }
    }
}

// A few rounds of odd-even transposition: finishes lists that are sorted up to local disorder (seed
// lists of position-sorted VCFs).  Returns true when the list is sorted afterwards.
__device__ bool oddeven_rounds(uint32_t *a, uint32_t n, uint32_t tid, uint32_t nthreads, uint32_t *s_flag, int max_rounds)
{
    for (int r = 0; r < max_rounds; ++r) {
        if (tid == 0) *s_flag = 0;
        __syncthreads();
        for (int par = 0; par < 2; ++par) {
            for (uint32_t i = 2 * tid + par; i + 1 < n; i += 2 * nthreads) {
                const uint32_t x = a[i], y = a[i + 1];
                if (x > y) { a[i] = y; a[i + 1] = x; *s_flag = 1; }
            }
            __syncthreads();
        }
        if (*s_flag == 0) return true;
        __syncthreads();
    }
    return false;
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

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *a, uint32_t n, uint32_t key)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(kSortThreads) void ef_seed_sort(const Params p)
{
    __shared__ __align__(16) uint32_t s_key[kSortLds];
    __shared__ uint32_t s_part[kSortThreads / 64 + 1];
    __shared__ uint32_t s_unsorted;
    __shared__ uint32_t s_nruns, s_run[kMaxRuns + 1];
    __shared__ uint32_t s_nlong, s_long[kLongTiles][2];
    __shared__ uint32_t s_tab[kSeedTab];                  // the distinct seeds of an unsorted list (open addressing)
    __shared__ uint32_t s_ndist;
    const uint32_t k = blockIdx.x, tid = threadIdx.x;
    STAMP(1, 0);
    const uint32_t c_lo = p.n_small ? p.ctg_small[k] : p.ctg_off[k];
    const uint32_t c_hi = p.n_small ? p.ctg_small[k + 1] : p.ctg_off[k + 1];
    uint32_t *region = p.onebuf + c_lo + k;                  // {n, seeds...}; capacity 1 + (c_hi - c_lo)
    if (c_lo == c_hi) {
        if (tid == 0) { p.n_one[k] = 0; region[0] = 0; }
        return;
    }
    if (tid == 0) { s_unsorted = 0; s_nruns = 0; s_nlong = 0; }
    // Seed entries of the classify tiles that overlap this contig, in candidate order: thread t owns
    // a contiguous run of tiles, counts the entries it will keep, and an exclusive scan gives its
    // output position -- so a position-sorted VCF yields an (almost always) already ascending list.
    const uint32_t b_lo = c_lo / kCandPerBlock, b_hi = (c_hi - 1) / kCandPerBlock;
    const uint32_t nb = b_hi - b_lo + 1;
    const uint32_t per = (nb + kSortThreads - 1) / kSortThreads;
    const uint32_t my_lo = min(b_lo + tid * per, b_hi + 1), my_hi = min(my_lo + per, b_hi + 1);
    uint32_t *glist = region + 1;
    // entry j of tile b, given the tile's record
    auto entry = [&](uint32_t b, uint32_t j, const ulonglong4 &rec) -> uint64_t {
        return j == 0 ? rec.y : (j == 1 ? rec.z : (j == 2 ? rec.w : p.seed_ent[(size_t)b * kCandPerBlock + j]));
    };
    const TileRecs recs{reinterpret_cast<const ulonglong4 *>(p.blk_rec)};
    // records of my first tile and of the tile before it (whose last entry, if it belongs to this contig,
    // absorbs an equal first entry): both loads are independent -> one round trip
    ulonglong4 rec0 = {0, 0, 0, 0}, recp = {0, 0, 0, 0};
    if (my_lo < my_hi) {
        rec0 = recs[my_lo];
        if (my_lo > b_lo) recp = recs[my_lo - 1];
    }
    // (the edge tiles' counts ride in the same round trip)
    const uint32_t cnt_lo = (uint32_t)recs[b_lo].x, cnt_hi = (uint32_t)recs[b_hi].x;
    STAMP(1, 4);
    uint32_t prev0 = kEmpty;
    if (my_lo < my_hi && my_lo > b_lo && (uint32_t)recp.x) {
        const uint64_t pe = entry(my_lo - 1, (uint32_t)recp.x - 1, recp);
        if ((uint32_t)(pe >> 32) >= c_lo) prev0 = (uint32_t)pe;
    }
    // A tile's entries are already free of repeats among themselves (seeds_tile), so for a tile that lies wholly inside
    // the contig only its first entry can fall to the neighbour rule: its count needs no loop, and its entries beyond
    // the three inline ones are copied by the whole workgroup afterwards (a tile of sparse candidates can hold 256 of
    // them: one thread walking those through HBM twice cost 47 us).  Tiles on the contig's edges take the loop.
    const uint32_t n_total = p.dyn_c ? *p.dyn_c : p.C;
    auto interior = [&](uint32_t b) {
        return b * kCandPerBlock >= c_lo && ((b + 1) * kCandPerBlock <= c_hi || c_hi == n_total);
    };
    // A tile on the contig's edge holds entries of two contigs: its owner walks it -- unless it is long, then the
    // workgroup filters it together (one scan), in front of / behind the tiles in between.
    const bool lo_coop = !interior(b_lo) && cnt_lo > kLongCnt;
    const bool hi_coop = b_hi != b_lo && !interior(b_hi) && cnt_hi > kLongCnt;
    auto edge_tile = [&](uint32_t b, uint32_t cnt, uint32_t base) -> uint32_t {
        const ulonglong4 rec = recs[b];
        bool keep = false;
        uint32_t ps = 0;
        if (tid < cnt) {
            const uint64_t e = entry(b, tid, rec);
            const uint32_t c = (uint32_t)(e >> 32);
            ps = (uint32_t)e;
            keep = c >= c_lo && c < c_hi;
        }
        uint32_t total;
        const uint32_t at = base + block_exscan(keep ? 1u : 0u, tid, s_part, kSortThreads, &total);
        if (keep) {
            if (at < kSortLds) s_key[at] = ps;
            glist[at] = ps;
        }
        return total;
    };
    const uint32_t n_lo = lo_coop ? edge_tile(b_lo, cnt_lo, 0) : 0u;
    uint32_t mid_total = 0;
    uint32_t mine = 0;
    for (int pass = 0; pass < 2; ++pass) {
        uint32_t at = 0;
        if (pass == 1) {
            uint32_t total;
            at = n_lo + block_exscan(mine, tid, s_part, kSortThreads, &total);
            mid_total = total;
            STAMP(1, 5);
        }
        uint32_t prev_ps = prev0;
        for (uint32_t b = my_lo; b < my_hi; ++b) {
            const ulonglong4 rec = b == my_lo ? rec0 : recs[b];
            const uint32_t cnt = (uint32_t)rec.x;
            if (cnt == 0 || (b == b_lo && lo_coop) || (b == b_hi && hi_coop)) continue;
            if (interior(b)) {
                const uint32_t skip = (uint32_t)rec.y == prev_ps ? 1u : 0u;
                if (pass == 0) {
                    mine += cnt - skip;
                } else {
                    for (uint32_t j = skip; j < cnt && j < 3; ++j) {
                        const uint32_t ps = (uint32_t)entry(b, j, rec);
                        if (at + j - skip < kSortLds) s_key[at + j - skip] = ps;
                        glist[at + j - skip] = ps;
                    }
                    if (cnt > 3) {
                        const uint32_t w = cnt > kLongCnt ? atomicAdd(&s_nlong, 1u) : kLongTiles;
                        if (w < kLongTiles) {
                            s_long[w][0] = b;
                            s_long[w][1] = at + 3 - skip;
                        } else {                               // a few more entries (or a full work list): copy them here
                            for (uint32_t j = 3; j < cnt; ++j) {
                                const uint32_t ps = (uint32_t)p.seed_ent[(size_t)b * kCandPerBlock + j];
                                if (at + j - skip < kSortLds) s_key[at + j - skip] = ps;
                                glist[at + j - skip] = ps;
                            }
                        }
                    }
                    at += cnt - skip;
                }
                if (b + 1 < my_hi) prev_ps = (uint32_t)entry(b, cnt - 1, rec);
                continue;
            }
            for (uint32_t j = 0; j < cnt; ++j) {
                const uint64_t e = entry(b, j, rec);
                const uint32_t c = (uint32_t)(e >> 32), ps = (uint32_t)e;
                if (c < c_lo || c >= c_hi) continue;
                if (ps == prev_ps) continue;                  // same contig (both candidates in [c_lo, c_hi))
                if (pass == 0) {
                    ++mine;
                } else {
                    if (at < kSortLds) s_key[at] = ps;
                    glist[at] = ps;                           // also kept in HBM for the large path
                    ++at;
                }
                prev_ps = ps;
            }
        }
    }
    __syncthreads();
    STAMP(1, 6);
    {
        const uint32_t nl = min(s_nlong, kLongTiles);
        // (one wave per long tile: the tiles' round trips overlap instead of queueing behind each other)
        for (uint32_t w = tid >> 6; w < nl; w += kSortThreads / 64) {
            const uint32_t b = s_long[w][0], at = s_long[w][1];
            const uint32_t cnt = (uint32_t)recs[b].x;
            for (uint32_t j = 3 + (tid & 63u); j < cnt; j += 64u) {
                const uint32_t ps = (uint32_t)p.seed_ent[(size_t)b * kCandPerBlock + j];
                if (at + j - 3 < kSortLds) s_key[at + j - 3] = ps;
                glist[at + j - 3] = ps;
            }
        }
    }
    __syncthreads();
    const uint32_t n_hi = hi_coop ? edge_tile(b_hi, cnt_hi, n_lo + mid_total) : 0u;
    __syncthreads();
    STAMP(1, 1);
    const uint32_t n = n_lo + mid_total + n_hi;
    {
        // descents = starts of new ascending runs; none: the list is sorted
        const uint32_t *lst = n <= kSortLds ? s_key : glist;
        for (uint32_t i = tid + 1; i < n; i += kSortThreads)
            if (lst[i] < lst[i - 1]) {
                const uint32_t r = atomicAdd(&s_nruns, 1u);
                if (r < kMaxRuns) s_run[r] = i;
            }
    }
    __syncthreads();
    STAMP(1, 7);
    const bool unsorted = s_nruns != 0;
    uint32_t n_one;
    if (n == 0) {
        n_one = 0;
    } else if (n <= kSortLds) {
        // Many descents are usually the local disorder of position-ordered candidates (neighbours that straddle a phase-set
        // boundary): one or two odd-even rounds remove it (the rounds stop as soon as nothing moves).  What is unsorted after
        // that -- long runs: candidates ordered by type, then position (stage A0's order) give one run per type; or real
        // disorder -- goes through a hash set when its distinct values are few (below), else: a few long runs are merged by
        // rank, after four more rounds the rest takes the bitonic network.  (Six rounds before looking at the runs again cost
        // the fused pipeline 12 of its 31 us here -- the long runs never go away --, and merging 20 short runs by rank costs a
        // position-sorted VCF 10 us where two rounds cost 1.)
        bool todo = unsorted;
        auto count_runs = [&]() {
            __syncthreads();
            if (tid == 0) s_nruns = 0;
            __syncthreads();
            for (uint32_t i = tid + 1; i < n; i += kSortThreads)
                if (s_key[i] < s_key[i - 1]) {
                    const uint32_t r = atomicAdd(&s_nruns, 1u);
                    if (r < kMaxRuns) s_run[r] = i;
                }
            __syncthreads();
        };
        // (the device-planned launch is the fused pipeline's: its candidates come by type, the rounds would be wasted -- 1.5 us)
        const bool type_major = p.dyn_c != nullptr && !(p.dbg & 0x40u);
        if (todo && s_nruns > kFewRuns && !type_major) {
            todo = !oddeven_rounds(s_key, n, tid, kSortThreads, &s_unsorted, 2);
            if (todo) count_runs();
        }
        // (a handful of long ascending runs -- stage A0's candidates: one run per type once the rounds above have dealt with the
        // local disorder -- are merged by rank below, two rounds of binary searches; the hash set is for real disorder)
        if (todo && n >= 256u && s_nruns > kFewRuns && !(p.dbg & 0x40u)) {
            // What two odd-even rounds did not fix is unsorted because of the ORDER of the candidates (stage A0 emits them by type,
            // then position; a caller's VCF may not be position-sorted), not because there are many different seeds: a contig has a few
            // hundred phase sets.  The distinct values through a hash set, then those few are ranked -- instead of ordering
            // all n entries.  More than kSeedTab / 2 distinct seeds, or a probe sequence of more than 32: the paths below.
            for (uint32_t i = tid; i < kSeedTab; i += kSortThreads) s_tab[i] = kEmpty;
            if (tid == 0) { s_ndist = 0; s_unsorted = 0; }
            __syncthreads();
            for (uint32_t i = tid; i < n; i += kSortThreads) {
                const uint32_t key = s_key[i];                 // (a PS is at most 2^32 - 2: kEmpty is free)
                uint32_t h = (key * 2654435761u) >> 20, probes = 0;
                for (;;) {
                    const uint32_t old = atomicCAS(&s_tab[h], kEmpty, key);
                    if (old == kEmpty) { atomicAdd(&s_ndist, 1u); break; }
                    if (old == key) break;
                    h = (h + 1u) & (kSeedTab - 1u);
                    if (++probes > 32u) { s_unsorted = 1; break; }
                }
            }
            __syncthreads();
            const uint32_t u = s_ndist;
            if (s_unsorted == 0 && u <= kSeedTab / 2u) {
                // compact the set (four slots per thread), order the values (blocks of 64 in registers, places by lower bounds), store them
                uint32_t mine[kSeedTab / kSortThreads], cnt = 0;
#pragma unroll
                for (uint32_t j = 0; j < kSeedTab / kSortThreads; ++j) {
                    mine[j] = s_tab[tid * (kSeedTab / kSortThreads) + j];
                    cnt += mine[j] != kEmpty ? 1u : 0u;
                }
                uint32_t total;
                uint32_t at = block_exscan(cnt, tid, s_part, kSortThreads, &total);
                __syncthreads();                               // (everybody has read s_key and the table)
#pragma unroll
                for (uint32_t j = 0; j < kSeedTab / kSortThreads; ++j)
                    if (mine[j] != kEmpty) s_key[at++] = mine[j];
                // The u distinct values in order.  Ranking every value against every other one is u x u compares on ONE compute
                // unit (4.6 of this kernel's 13.9 us for 499 values, whether the columns come as LDS broadcasts -- the CU's LDS
                // port -- or rotate through the lanes by DPP -- 2 vector instructions per compare); instead: every wavefront
                // orders blocks of 64 values in its registers (wave_sort64: no barrier), and a value's place
                // is the sum of its lower bounds in all blocks (its own included: the values are distinct) -- seven reads per
                // block, four blocks side by side, the blocks of a value shared by 1024 / U threads.
                const uint32_t U = (u + 63u) & ~63u, nblk = U >> 6, lane = tid & 63u;
                if (tid < U - u) s_key[u + tid] = kEmpty;                     // (pads the last block: sorts behind every seed)
                for (uint32_t i = tid; i < U; i += kSortThreads) s_tab[i] = 0u;
                __syncthreads();
                for (uint32_t b = tid >> 6; b < nblk; b += kSortThreads / 64) {
                    s_key[b * 64u + lane] = wave_sort64(s_key[b * 64u + lane], lane);
                }
                __syncthreads();
                {
                    const uint32_t P = max(1u, (uint32_t)kSortThreads / U), per_part = (nblk + P - 1u) / P;
                    for (uint32_t w = tid; w < U * P; w += kSortThreads) {
                        const uint32_t part = w / U, i = w - part * U;
                        const uint32_t b0 = min(part * per_part, nblk), b1 = min(b0 + per_part, nblk);
                        const uint32_t me = s_key[i];
                        uint32_t rank = 0;
                        for (uint32_t b = b0; b < b1; b += 4u) {
                            const uint32_t *blk[4];
                            uint32_t pos[4];
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                blk[t] = s_key + min(b + (uint32_t)t, b1 - 1u) * 64u;
                                pos[t] = 0;
                            }
#pragma unroll
                            for (uint32_t st = 32; st > 0; st >>= 1)
#pragma unroll
                                for (int t = 0; t < 4; ++t) pos[t] += blk[t][pos[t] + st - 1u] < me ? st : 0u;
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                pos[t] += blk[t][pos[t]] < me ? 1u : 0u;
                                rank += b + (uint32_t)t < b1 ? pos[t] : 0u;
                            }
                        }
                        if (P == 1u) s_tab[i] = rank;
                        else if (rank) atomicAdd(&s_tab[i], rank);
                    }
                }
                __syncthreads();
                for (uint32_t i = tid; i < U; i += kSortThreads) {
                    const uint32_t v = s_key[i];
                    if (v != kEmpty) glist[s_tab[i]] = v;
                }
                if (tid == 0) { p.n_one[k] = u; region[0] = u; }
                STAMP(1, 3);
                return;
            }
            __syncthreads();
        }
        if (todo && s_nruns >= kMaxRuns) {
            todo = !oddeven_rounds(s_key, n, tid, kSortThreads, &s_unsorted, 4);
            if (todo) count_runs();
        }
        if (todo && s_nruns < kMaxRuns) {
            // Up to 31 ascending runs: merge neighbouring runs pairwise, log2(runs) rounds.  An element's place in the merged
            // run is its index in its own run plus the number of elements of the sibling run that go before it (for the left
            // run those smaller than it, for the right run those smaller or equal: distinct places, equal seeds stay apart until
            // unique_copy drops them).  One binary search per element and round -- ranking every element against every other
            // run at once took 19 searches per element for 20 runs.
            uint32_t R = s_nruns + 1;
            if (tid == 0) {
                for (uint32_t a = 1; a < R - 1; ++a) {          // the descents were appended in any order
                    const uint32_t v = s_run[a];
                    uint32_t b = a;
                    while (b > 0 && s_run[b - 1] > v) { s_run[b] = s_run[b - 1]; --b; }
                    s_run[b] = v;
                }
                for (uint32_t a = R - 1; a > 0; --a) s_run[a] = s_run[a - 1];
                s_run[0] = 0;
                s_run[R] = n;
            }
            __syncthreads();
            constexpr uint32_t kPer = (kSortLds + kSortThreads - 1) / kSortThreads;
            while (R > 1) {
                uint32_t val[kPer], dst[kPer];
#pragma unroll
                for (uint32_t t = 0; t < kPer; ++t) {
                    const uint32_t i = tid + t * kSortThreads;
                    dst[t] = kEmpty;
                    if (i < n) {
                        uint32_t lo_r = 0, hi_r = R;                 // the run of position i: last a with s_run[a] <= i
                        while (hi_r - lo_r > 1) {
                            const uint32_t mid = (lo_r + hi_r) >> 1;
                            if (s_run[mid] <= i) lo_r = mid; else hi_r = mid;
                        }
                        const uint32_t a = lo_r, b = a ^ 1u;
                        if (b < R) {
                            const uint32_t x = s_key[i], lo = s_run[a], blo = s_run[b], blen = s_run[b + 1] - blo;
                            // PS <= 2^32 - 2: x + 1 cannot wrap
                            const uint32_t before = lower_bound_u32(s_key + blo, blen, a < b ? x : x + 1u);
                            val[t] = x;
                            dst[t] = (a < b ? lo : blo) + (i - lo) + before;
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (uint32_t t = 0; t < kPer; ++t)
                    if (dst[t] != kEmpty) s_key[dst[t]] = val[t];
                __syncthreads();
                const uint32_t newR = (R + 1u) >> 1;
                if (tid == 0) {
                    for (uint32_t j = 1; j < newR; ++j) s_run[j] = s_run[2 * j];
                    s_run[newR] = n;
                }
                __syncthreads();
                R = newR;
            }
        } else if (todo) {
            bitonic_sort(s_key, n, tid, kSortThreads);
        }
        STAMP(1, 2);
        n_one = unique_copy(s_key, n, glist, tid, kSortThreads, s_part);
    } else {
        // more seeds than LDS holds (unsorted input with very many phase sets): work in HBM
        __threadfence_block();
        if (unsorted) bitonic_sort(glist, n, tid, kSortThreads);
        uint32_t *tmp = p.tmpbuf + c_lo + k + 1;
        n_one = unique_copy(glist, n, tmp, tid, kSortThreads, s_part);
        __syncthreads();
        for (uint32_t i = tid; i < n_one; i += kSortThreads) glist[i] = tmp[i];
    }
    if (tid == 0) { p.n_one[k] = n_one; region[0] = n_one; }
    STAMP(1, 3);
}

// ---------------------------------------------------------------------------------------------
// kernel 3: contig drop, nearest PS, multi-PS vote + decision
// ---------------------------------------------------------------------------------------------

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

// f(tag) for every mark of [b, e) in list order, the marks' two dependent loads (read index, then tag) eight marks at a time:
// one pair of round trips per eight marks instead of one per mark
template <class F>
__device__ __forceinline__ void for_each_tag(const Params &p, uint32_t b, uint32_t e, F f)
{
    for (uint32_t m0 = b; m0 < e; m0 += 8) {
        uint32_t r[8];
        uint64_t t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = p.mark_read[m0 + j < e ? m0 + j : b];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = p.read_tag[r[j] == kEmpty ? 0u : r[j]];       // (read_tag always has >= 1 readable word)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (m0 + j < e) f(r[j] == kEmpty ? kUntagged : t[j]);
    }
}

// multi-PS vote straight from the marks (:85-105): only for candidates without a group summary.  mem(ps): is this PS a seed of
// the contig (:91)
template <class Mem>
__device__ __forceinline__ void class2_from_marks_m(const Params &p, uint32_t c, Mem mem, Vote &v, uint32_t &ps)
{
    const uint32_t b = p.cand_off[c], e = p.cand_off[c + 1];
    uint32_t best = 0;
    for_each_tag(p, b, e, [&](uint64_t t) {
        if (t != kUntagged && tag_pc(t) <= kPcMax) ++v.allhap;
    });
    uint32_t done_ps = kEmpty;                                 // last group evaluated (cheap duplicate skip)
    for_each_tag(p, b, e, [&](uint64_t t) {
        if (t == kUntagged || tag_pc(t) > kPcMax) return;
        const uint32_t g = tag_ps(t);
        if (g == done_ps || (g == ps && best)) return;
        if (!mem(g)) return;                                   // :91
        // size and sums of g's group over the whole list; a later occurrence of an already
        // evaluated group reproduces the same n and cannot beat it (strict '>', :101)
        uint32_t n = 0, n1 = 0, n2 = 0;
        uint64_t s1 = 0, s2 = 0;
        for_each_tag(p, b, e, [&](uint64_t u) {
            if (u == kUntagged || tag_pc(u) > kPcMax || tag_ps(u) != g) return;
            ++n;
            const uint32_t hap = tag_hap(u);
            if (hap == 1) { ++n1; s1 += tag_pc(u); }
            else if (hap == 2) { ++n2; s2 += tag_pc(u); }
        });
        done_ps = g;
        if (n > best) {
            best = n; ps = g;
            v.hap1 = n1; v.hap2 = n2; v.t1 = s1; v.t2 = s2;
            v.hap0 = v.allhap - n1 - n2;                       // only with a winner (:105)
        }
    });
}
__device__ void class2_from_marks(const Params &p, uint32_t c, const uint32_t *one, uint32_t n_one, Vote &v, uint32_t &ps)
{
    class2_from_marks_m(p, c, [&](uint32_t g) { return is_member(one, n_one, g); }, v, ps);
}

// one candidate's part of ef_finalize (everything after the tile's seed array is staged)
__device__ __forceinline__ void finalize_candidate(const Params &p, uint32_t c, uint8_t code, uint32_t ps_in, uint32_t k0, bool lds_mode,
                                                   const uint32_t *s_one, uint32_t n_lds, uint32_t any_empty)
{
    if (code < 4 && !any_empty) return;                        // already final
    // (what the candidate will need of its own columns leaves now, beside the group summary and the seeds -- not one
    // round trip behind them)
    const bool two = (code & (kClass2 | kClass2Slow)) != 0 && !(code & kDivZero);
    const bool work = code >= 4 && !(code & kDivZero);
    uint32_t c_pos = 0, c_o0 = 0, c_o1 = 0, c_svread = 0, c_refread = 0;
    if (work) c_pos = p.cand_pos[c];
    if (two) { c_o0 = p.cand_off[c]; c_o1 = p.cand_off[c + 1]; c_svread = p.cand_svread[c]; c_refread = p.cand_refread[c]; }
    const uint32_t *one;
    uint32_t n_one;
    if (lds_mode) {
        one = s_one;
        n_one = n_lds;
    } else {
        uint32_t k = k0;
        while (c >= p.ctg_off[k + 1]) ++k;
        n_one = p.n_one[k];
        one = p.onebuf + p.ctg_off[k] + k + 1;
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
        if (v.hap1 == 0 && v.hap2 == 0) ps = nearest_ps(one, n_one, c_pos);              // :106
        const uint32_t deg = c_o1 - c_o0;
        p.out_pred[c] = (uint8_t)decide(2, v, deg, c_svread, c_refread);
        p.out_ps[c] = ps;
        return;
    }
    // kNeedNearest
    p.out_pred[c] = code & 3;
    p.out_ps[c] = nearest_ps(one, n_one, c_pos);
}

// TPB: tiles of 256 candidates per workgroup (host-planned runs with the contig offsets in the kernel arguments only)
template <bool DYN = false, int TPB = 1>
__global__ __launch_bounds__(256) void ef_finalize(const Params p)
{
    __shared__ uint32_t s_one[kOneLds];
    __shared__ uint32_t s_meta[4];                             // k0, n_one[k0] or ~0 (not LDS mode), any_empty, seed base of k0
    const uint32_t tid = threadIdx.x;
    STAMP(2, 0);
    if (blockIdx.x == 0 && tid == 0) p.status[1] = 0;          // the summary pool's counter, for the next run's ef_classify
    const uint32_t n_cands = DYN ? *p.dyn_c : p.C;
    if (!DYN && p.n_small) {
        // The contig offsets ride in the kernel arguments (K <= kSmallK): the tile's contigs are a binary search on the
        // scalar unit, no memory; the seed count and the first 1024 seeds of the tile's contig leave TOGETHER with the
        // candidates' codes -- one round trip where thread 0 used to walk four dependent ones (tile's contig -> contig
        // offsets -> seed count -> seeds) in front of everybody.
        const uint32_t c0 = blockIdx.x * (256u * TPB);
        uint8_t code[TPB];
        uint32_t ps_in[TPB];
#pragma unroll
        for (int r = 0; r < TPB; ++r) {
            const uint32_t c = c0 + 256u * r + tid;
            code[r] = c < n_cands ? p.out_pred[c] : 0;
            ps_in[r] = c < n_cands ? p.out_ps[c] : 0;
        }
        const uint32_t last = min(c0 + 256u * TPB - 1u, n_cands - 1);
        uint32_t lo = 0, hi = p.n_small;                       // the last k with ctg_small[k] <= c0
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (p.ctg_small[mid] <= c0) lo = mid; else hi = mid;
        }
        const uint32_t k0 = lo;
        uint32_t k1 = k0;
        while (last >= p.ctg_small[k1 + 1]) ++k1;
        const uint32_t base = p.ctg_small[k0] + k0 + 1;
        uint32_t ahead[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ahead[i] = p.onebuf[min(base + tid + 256u * i, p.one_cap - 1u)];
        uint32_t any = 0;
        for (uint32_t k = k0; k <= k1; ++k) any |= (p.n_one[k] == 0) ? 1u : 0u;
        const uint32_t n0 = p.n_one[k0];
        const bool lds_mode = k0 == k1 && n0 > 0 && n0 <= kOneLds;
        if (lds_mode) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (tid + 256u * i < n0) s_one[tid + 256u * i] = ahead[i];
            for (uint32_t i = tid + 1024u; i < n0; i += 256u) s_one[i] = p.onebuf[base + i];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < TPB; ++r) {
            const uint32_t c = c0 + 256u * r + tid;
            if (c < n_cands) finalize_candidate(p, c, code[r], ps_in[r], k0, lds_mode, s_one, n0, any);
        }
        return;
    }
    // (device-planned runs stride over the real tiles like ef_classify; TPB > 1 only with the candidates' contig column)
    for (uint32_t tile = blockIdx.x; !DYN || tile * (256u * (p.cand_contig ? TPB : 1)) < n_cands; tile += gridDim.x) {
        if (DYN && p.cand_contig) {
            // device-planned run with the candidates' contig column: the first and last contig of the workgroup's TPB tiles from the
            // column (they leave with the codes), then the contig's offset and seed count, then the seeds -- every thread the same
            // (scalar) loads, not thread 0 walking them in front of a barrier; with TPB > 1 (large inputs) those round trips and the
            // seeds' way into LDS are paid once per 256 x TPB candidates, as in the host-planned variant above
            const uint32_t g0 = tile * (256u * TPB);
            if (g0 >= n_cands) break;
            uint8_t codes[TPB];
            uint32_t pss[TPB];
#pragma unroll
            for (int r = 0; r < TPB; ++r) {
                const uint32_t cc = g0 + 256u * r + tid;
                codes[r] = cc < n_cands ? p.out_pred[cc] : 0;
                pss[r] = cc < n_cands ? p.out_ps[cc] : 0;
            }
            const uint32_t last = min(g0 + 256u * TPB - 1u, n_cands - 1);
            const uint32_t k0 = p.cand_contig[g0], k1 = p.cand_contig[last];
            uint32_t any = 0;
            for (uint32_t k = k0; k <= k1; ++k) any |= (p.n_one[k] == 0) ? 1u : 0u;
            const uint32_t n0 = p.n_one[k0], base = p.ctg_off[k0] + k0 + 1;
            const bool lds_mode = k0 == k1 && n0 > 0 && n0 <= kOneLds;
            if (lds_mode)
                for (uint32_t i = tid; i < n0; i += 256u) s_one[i] = p.onebuf[base + i];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < TPB; ++r) {
                const uint32_t cc = g0 + 256u * r + tid;
                if (cc < n_cands) finalize_candidate(p, cc, codes[r], pss[r], k0, lds_mode, s_one, n0, any);
            }
            __syncthreads();                                   // s_one is reused
            continue;
        }
        const uint32_t c0 = tile * 256u;
        const uint32_t c = c0 + tid;
        const bool live = c < n_cands;
        const uint8_t code = live ? p.out_pred[c] : 0;
        const uint32_t ps_in = live ? p.out_ps[c] : 0;
        if (tid == 0) {
            const uint32_t last = min(c0 + 255u, n_cands - 1);
            // (a device-planned run with the candidates' contig column reads the tile's contig there: no table of the tiles' contigs)
            const uint32_t k0 = (DYN && p.cand_contig) ? (uint32_t)p.cand_contig[c0] : p.blk_ctg[tile];
            uint32_t k1 = k0;
            while (last >= p.ctg_off[k1 + 1]) ++k1;
            uint32_t any = 0;
            for (uint32_t k = k0; k <= k1; ++k) any |= (p.n_one[k] == 0);
            const uint32_t n0 = p.n_one[k0];
            s_meta[0] = k0;
            s_meta[1] = (k0 == k1 && n0 > 0 && n0 <= kOneLds) ? n0 : kEmpty;
            s_meta[2] = any;
            s_meta[3] = p.ctg_off[k0] + k0 + 1;
        }
        __syncthreads();
        STAMP(2, 1);
        const uint32_t k0 = s_meta[0], n_lds = s_meta[1], any_empty = s_meta[2];
        const bool lds_mode = n_lds != kEmpty;
        if (lds_mode) {
            const uint32_t *src = p.onebuf + s_meta[3];
            for (uint32_t i = tid; i < n_lds; i += 256u) s_one[i] = src[i];
            __syncthreads();
        }
        STAMP(2, 2);
        if (live) finalize_candidate(p, c, code, ps_in, k0, lds_mode, s_one, n_lds, any_empty);
        if (!DYN) break;
        __syncthreads();                                       // s_meta / s_one are reused
    }
}

// ---------------------------------------------------------------------------------------------
// kernels 2 + 3 in ONE launch for problems of the size real genomes have (round 6): every finalize tile builds the seed
// set of its contig ITSELF
// ---------------------------------------------------------------------------------------------
// at BASELINE configs[1] (and at SURVEY 8d's estimate of real ONT genomes, 1e5 .. 1e6 marks) the step
// was three dependent launches, ~15 us of launch and first-round-trip floors for ~2 us of traffic.  ef_seed_sort -- ONE
// workgroup per contig between two grid-wide launches -- was 6.9 us of config 2's 23.9.  Here it is gone: a finalize tile reads the
// seed entries of ALL tiles of its contig (32 bytes per tile beside a handful of overflow entries: 12.5 KB at config 2, out of
// L2 / the Infinity Cache, in the same round trip that fetches the tile's own codes), takes their distinct values through a hash set
// in LDS, orders the few hundred that are left (blocks of 64 in registers, places by lower bounds: ef_seed_sort's own scheme) and goes
// on with :106-111 / :85-105 / :209-210 on an array in LDS.  The work is redundant -- T tiles each read T records -- which is why
// it is only taken up to kOwnMaxTiles tiles (T x T x 32 B = 33 MB of cache reads at the limit); no workgroup waits for another one,
// the launch boundary behind ef_classify is the only synchronisation (a single cooperative launch was priced and measured
// instead of built: DESIGN.md section 3, profiles/history/r06_single_launch_*).  Only the SET of seeds matters downstream
// (np.sort(list(oneps_set)), :107): candidate order, repeats and the neighbour rule of ef_seed_sort's gather are immaterial.
// More than kOwnKeys distinct seeds in a contig (an unsorted VCF over thousands of phase sets): the tile answers its few
// questions -- is this PS a seed, which seed is nearest -- by walking the contig's entries where they lie.  The ascending array
// the ABI can hand out (duet_ef_get_seed_ps, duet_ef_stats.n_seed_ps) is made on demand by ef_seed_sort from the same entries.
constexpr uint32_t kOwnMaxTiles = 2048;            // (measured, tools/own_sweep.py: ahead up to ~1,600 tiles on the 24-contig genome, level at 3,100, behind beyond)
constexpr uint32_t kOwnTab = 2048;                 // hash-set slots (open addressing; the hash takes the product's top 11 bits)
constexpr uint32_t kOwnKeys = 1024;                // distinct seeds kept in LDS

// what the array-free walk needs of the kernel's arguments, by value: the walk is a rare path behind a real call, and a call that
// took the 600-byte argument block by reference would make EVERY launch copy that block to scratch memory first
struct OwnArgs {
    TileRecs recs;
    const uint64_t *seed_ent, *read_tag;
    const uint32_t *cand_pos, *cand_off, *cand_svread, *cand_refread, *mark_read, *c2rec;
    uint32_t *out_ps, *status;
    uint8_t *out_pred;
    uint32_t c_lo, c_hi;
};

// f(ps) for every seed entry of contig [c_lo, c_hi), tiles dealt out with a stride
template <class F>
__device__ __forceinline__ void own_each_entry(TileRecs recs, const uint64_t *seed_ent, uint32_t c_lo, uint32_t c_hi, uint32_t first,
                                               uint32_t stride, F f)
{
    const uint32_t b_lo = c_lo / kCandPerBlock, b_hi = (c_hi - 1) / kCandPerBlock;
    for (uint32_t b = b_lo + first; b <= b_hi; b += stride) {
        const ulonglong4 rec = recs[b];
        const uint32_t cnt = (uint32_t)rec.x;
        for (uint32_t j = 0; j < cnt; ++j) {
            const uint64_t e = j == 0 ? rec.y : (j == 1 ? rec.z : (j == 2 ? rec.w : seed_ent[(size_t)b * kCandPerBlock + j]));
            const uint32_t c = (uint32_t)(e >> 32);
            if (c >= c_lo && c < c_hi) f((uint32_t)e);
        }
    }
}

// the two questions without an array: the thread walks all entries of the contig (rare: see above)
__device__ __forceinline__ uint32_t own_nearest_walk(const OwnArgs &a, uint32_t pos)
{
    uint32_t best = 0;
    int64_t bd = -1;
    own_each_entry(a.recs, a.seed_ent, a.c_lo, a.c_hi, 0u, 1u, [&](uint32_t ps) {
        const int64_t d = llabs((int64_t)pos - (int64_t)ps);
        // nearest_ps: the left neighbour only when strictly nearer -- among equally near seeds the larger one
        if (bd < 0 || d < bd || (d == bd && ps > best)) { bd = d; best = ps; }
    });
    return best;
}
__device__ __forceinline__ bool own_member_walk(const OwnArgs &a, uint32_t key)
{
    bool hit = false;
    own_each_entry(a.recs, a.seed_ent, a.c_lo, a.c_hi, 0u, 1u, [&](uint32_t ps) { hit = hit || ps == key; });
    return hit;
}

template <class F>
__device__ __forceinline__ void own_each_tag(const OwnArgs &a, uint32_t b, uint32_t e, F f)
{
    for (uint32_t m = b; m < e; ++m) {
        const uint32_t r = a.mark_read[m];
        f(r == kEmpty ? kUntagged : a.read_tag[r]);
    }
}

// one candidate of a contig whose seeds did not fit the LDS set (finalize_candidate's logic, array-free; the contig has seeds)
__device__ __noinline__ void finalize_candidate_walk(OwnArgs a, uint32_t c, uint32_t code, uint32_t ps_in)
{
    if (code & kDivZero) {
        atomicOr(&a.status[0], 1u);
        a.out_pred[c] = 0;
        a.out_ps[c] = 0;
        return;
    }
    const uint32_t c_pos = a.cand_pos[c];
    if (code & (kClass2 | kClass2Slow)) {
        Vote v = {0, 0, 0, 0, 0, 0};
        uint32_t ps = 0;
        const uint32_t mb = a.cand_off[c], me = a.cand_off[c + 1];
        if (code & kClass2) {
            const uint32_t *rec = a.c2rec + (size_t)ps_in * kC2Words;
            v.allhap = rec[0];
            const uint32_t ng = rec[1];
            uint32_t best = 0;
            for (uint32_t k = 0; k < (uint32_t)kC2Groups; ++k) {
                const uint32_t *g = rec + 2 + 6 * k;
                if (k < ng && g[1] > best && own_member_walk(a, g[0])) {
                    best = g[1];
                    ps = g[0];
                    v.hap1 = g[2]; v.hap2 = g[3];
                    v.t1 = g[4]; v.t2 = g[5];
                    v.hap0 = v.allhap - v.hap1 - v.hap2;       // only with a winner (:105)
                }
            }
        } else {                                               // (class2_from_marks, membership by walking)
            uint32_t best = 0;
            own_each_tag(a, mb, me, [&](uint64_t t) {
                if (t != kUntagged && tag_pc(t) <= kPcMax) ++v.allhap;
            });
            uint32_t done_ps = kEmpty;
            own_each_tag(a, mb, me, [&](uint64_t t) {
                if (t == kUntagged || tag_pc(t) > kPcMax) return;
                const uint32_t g = tag_ps(t);
                if (g == done_ps || (g == ps && best)) return;
                if (!own_member_walk(a, g)) return;            // :91
                uint32_t n = 0, n1 = 0, n2 = 0;
                uint64_t s1 = 0, s2 = 0;
                own_each_tag(a, mb, me, [&](uint64_t u) {
                    if (u == kUntagged || tag_pc(u) > kPcMax || tag_ps(u) != g) return;
                    ++n;
                    const uint32_t hap = tag_hap(u);
                    if (hap == 1) { ++n1; s1 += tag_pc(u); }
                    else if (hap == 2) { ++n2; s2 += tag_pc(u); }
                });
                done_ps = g;
                if (n > best) {                                // strict: the first-seen group wins ties (:101)
                    best = n; ps = g;
                    v.hap1 = n1; v.hap2 = n2; v.t1 = s1; v.t2 = s2;
                    v.hap0 = v.allhap - n1 - n2;
                }
            });
        }
        if (v.hap1 == 0 && v.hap2 == 0) ps = own_nearest_walk(a, c_pos);                 // :106
        a.out_pred[c] = (uint8_t)decide(2, v, me - mb, a.cand_svread[c], a.cand_refread[c]);
        a.out_ps[c] = ps;
        return;
    }
    a.out_pred[c] = (uint8_t)(code & 3u);
    a.out_ps[c] = own_nearest_walk(a, c_pos);
}

// A workgroup barrier that waits for the wavefront's LDS traffic only.  __syncthreads() is a fence as well: it drains the global
// loads in flight (s_waitcnt vmcnt(0)) in front of s_barrier -- here that would put the summaries' round trip, issued early so that it
// runs beside the seed set's phases, in front of the first of them.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// maximum of one value per lane over the wavefront (row rotations, then the two row broadcasts: wave_sum's pattern)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// (rare path of ef_finalize_own: candidates out of position order) the hash set's values, ascending, into s_one[0 .. u): compacted,
// every wavefront orders blocks of 64 in its registers, a value's place is the sum of its lower bounds in all blocks (its own
// included: the values are distinct) -- ef_seed_sort's scheme.  Every thread of the workgroup calls it.
__device__ __noinline__ void own_sort_set(const uint32_t *s_tab, uint32_t *s_key, uint32_t *s_one, uint32_t *s_part, uint32_t u)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t mv[kOwnTab / 256u], cnt = 0;
#pragma unroll
    for (uint32_t j = 0; j < kOwnTab / 256u; ++j) {
        mv[j] = s_tab[tid + 256u * j];
        cnt += mv[j] != kEmpty ? 1u : 0u;
    }
    uint32_t x = cnt;                                               // inclusive running count inside the wavefront
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if ((int)lane >= d) x += y;
    }
    if (lane == 63) s_part[wave] = x;
    __syncthreads();
    uint32_t at = x - cnt;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) at += w < wave ? s_part[w] : 0u;
#pragma unroll
    for (uint32_t j = 0; j < kOwnTab / 256u; ++j)
        if (mv[j] != kEmpty) s_key[at++] = mv[j];
    const uint32_t U = (u + 63u) & ~63u, nblk = U >> 6;
    if (tid < U - u) s_key[u + tid] = kEmpty;                       // (pads the last block: sorts behind every seed)
    __syncthreads();
    for (uint32_t b = wave; b < nblk; b += 4u) s_key[b * 64u + lane] = wave_sort64(s_key[b * 64u + lane], lane);
    __syncthreads();
    for (uint32_t i = tid; i < U; i += 256u) {
        const uint32_t me = s_key[i];
        uint32_t rank = 0;
        // (four blocks side by side: their seven dependent reads each overlap)
        for (uint32_t b = 0; b < nblk; b += 4u) {
            const uint32_t *b0 = s_key + b * 64u, *b1 = s_key + min(b + 1u, nblk - 1u) * 64u;
            const uint32_t *b2 = s_key + min(b + 2u, nblk - 1u) * 64u, *b3 = s_key + min(b + 3u, nblk - 1u) * 64u;
            uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
#pragma unroll
            for (uint32_t st = 32; st > 0; st >>= 1) {
                p0 += b0[p0 + st - 1u] < me ? st : 0u;
                p1 += b1[p1 + st - 1u] < me ? st : 0u;
                p2 += b2[p2 + st - 1u] < me ? st : 0u;
                p3 += b3[p3 + st - 1u] < me ? st : 0u;
            }
            p0 += b0[p0] < me ? 1u : 0u;
            p1 += b1[p1] < me ? 1u : 0u;
            p2 += b2[p2] < me ? 1u : 0u;
            p3 += b3[p3] < me ? 1u : 0u;
            rank += p0 + (b + 1u < nblk ? p1 : 0u) + (b + 2u < nblk ? p2 : 0u) + (b + 3u < nblk ? p3 : 0u);
        }
        if (me != kEmpty) s_one[rank] = me;
    }
    __syncthreads();
}

// (rare path of ef_finalize_own) the multi-PS vote of a candidate without a group summary, straight from its marks (:85-105,
// class2_from_marks); "is this PS a seed" (:91) from the hash set in LDS
__device__ __noinline__ void own_vote_from_marks(const uint32_t *mark_read, const uint64_t *read_tag, uint32_t mb, uint32_t me_,
                                                 const uint32_t *s_tab, Vote *vo, uint32_t *pso)
{
    auto each = [&](auto f) {                                      // (eight marks' two dependent loads at a time, as for_each_tag)
        for (uint32_t m0 = mb; m0 < me_; m0 += 8) {
            uint32_t r[8];
            uint64_t t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = mark_read[m0 + j < me_ ? m0 + j : mb];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = read_tag[r[j] == kEmpty ? 0u : r[j]];       // (read_tag always has >= 1 readable word)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (m0 + j < me_) f(r[j] == kEmpty ? kUntagged : t[j]);
        }
    };
    auto member = [&](uint32_t key) -> bool {
        uint32_t h = (key * 2654435761u) >> (32u - 11u);
        for (;;) {
            const uint32_t v = s_tab[h];
            if (v == key) return true;
            if (v == kEmpty) return false;
            h = (h + 1u) & (kOwnTab - 1u);
        }
    };
    Vote v = {0, 0, 0, 0, 0, 0};
    uint32_t ps = 0, best = 0, done_ps = kEmpty;
    each([&](uint64_t t) {
        if (t != kUntagged && tag_pc(t) <= kPcMax) ++v.allhap;
    });
    each([&](uint64_t t) {
        if (t == kUntagged || tag_pc(t) > kPcMax) return;
        const uint32_t g = tag_ps(t);
        if (g == done_ps || (g == ps && best)) return;
        if (!member(g)) return;                                    // :91
        uint32_t n = 0, n1 = 0, n2 = 0;
        uint64_t s1 = 0, s2 = 0;
        each([&](uint64_t u) {
            if (u == kUntagged || tag_pc(u) > kPcMax || tag_ps(u) != g) return;
            ++n;
            const uint32_t hap = tag_hap(u);
            if (hap == 1) { ++n1; s1 += tag_pc(u); }
            else if (hap == 2) { ++n2; s2 += tag_pc(u); }
        });
        done_ps = g;
        if (n > best) {                                            // strict: the first-seen group wins ties (:101)
            best = n; ps = g;
            v.hap1 = n1; v.hap2 = n2; v.t1 = s1; v.t2 = s2;
            v.hap0 = v.allhap - n1 - n2;                           // only with a winner (:105)
        }
    });
    *vo = v;
    *pso = ps;
}

// ef_finalize_own's hash set, the probe sequence of `key` from slot h on: 1 when the key is new to the set, 0 when it is there already
// (or the table is full).  ONE copy of this loop behind a call: inlined at the fourteen places that may need it, it was 2,000 of the
// hot phase's 2,100 instructions.
__device__ __noinline__ uint32_t own_insert_slow(uint32_t *s_tab, uint32_t key, uint32_t h)
{
    for (uint32_t probes = 0; probes < kOwnTab; ++probes) {
        const uint32_t old = atomicCAS(&s_tab[h], kEmpty, key);
        if (old == kEmpty) return 1u;
        if (old == key) return 0u;
        h = (h + 1u) & (kOwnTab - 1u);
    }
    return 0u;
}

constexpr uint32_t kOwnWin = 64;                   // seeds strictly inside the tile's range of asking positions that are kept as a list

// DYN: the device-planned run of the fused clustered + phased pipeline -- the candidate count and the contig offsets exist on the
// device only (the grid strides over the real tiles; the offsets, K + 1 <= 65 words, come into LDS once per workgroup)
template <bool DYN>
__global__ __launch_bounds__(256) void ef_finalize_own(const Params p)
{
    __shared__ uint32_t s_coff[kSmallK + 1];
    __shared__ uint32_t s_tab[kOwnTab];                        // the hash set of the contig's seeds
    __shared__ uint32_t s_key[kOwnKeys];                       // (rare path) the distinct seeds, compacted, then in sorted blocks of 64
    __shared__ uint32_t s_one[kOwnKeys];                       // (rare path) ... ascending
    __shared__ uint32_t s_win[kOwnWin];
    __shared__ uint32_t s_mm[4][2];                            // per wavefront: ~min and max + 1 of the asking positions
    __shared__ uint32_t s_red[4][3];                           // per wavefront: left seed + 1, ~right seed, seeds new to the set
    __shared__ uint32_t s_part[4];
    __shared__ uint32_t s_nwin;
    __shared__ __align__(8) uint32_t s_c2[kC2Quota * kC2Words];      // the tile's own group-summary slots
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    STAMP(2, 0);
    if (blockIdx.x == 0 && tid == 0) p.status[1] = 0;          // the summary pool's counter, for the next run's ef_classify
    const uint32_t n_cands = DYN ? *p.dyn_c : p.C;
    const uint32_t nK = DYN ? p.K : p.n_small;
    if (DYN) {
        if (tid <= nK) s_coff[tid] = p.ctg_off[tid];
        __syncthreads();
    }
    // contig k's first candidate: out of the kernel arguments (host-planned: scalar loads, no memory) or out of LDS
    auto coff = [&](uint32_t kk) -> uint32_t { return DYN ? s_coff[kk] : p.ctg_small[kk]; };
  for (uint32_t tile = blockIdx.x; (uint64_t)tile * 256u < n_cands; tile += gridDim.x) {
    const uint32_t c0 = tile * 256u, c = c0 + tid;
    const bool live = c < n_cands;
    const uint32_t last = min(c0 + 255u, n_cands - 1);
    uint32_t lo = 0, hi = nK;                                  // the last k with coff(k) <= c0
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (coff(mid) <= c0) lo = mid; else hi = mid;
    }
    const uint32_t keys_cap = (p.dbg & DUET_DBG_EF_OWN_SMALLTAB) ? 8u : kOwnKeys;
    const uint32_t win_cap = (p.dbg & DUET_DBG_EF_OWN_SMALLTAB) ? 2u : kOwnWin;
    const TileRecs recs{reinterpret_cast<const ulonglong4 *>(p.blk_rec)};
    // ONE round trip in front of everything: the candidate's code, PS and position (a candidate that needs the nearest seed would
    // fetch its position one trip later) and up to four tiles' seed records of the tile's first contig
    const uint32_t code = live ? (uint32_t)p.out_pred[c] : 0u;
    const uint32_t ps_in = live ? p.out_ps[c] : 0;
    const uint32_t c_pos = live ? p.cand_pos[c] : 0;
    uint32_t k_first = lo;
    while (k_first < nK && coff(k_first) == coff(k_first + 1)) ++k_first;       // (the tile's first contig with candidates)
    constexpr uint32_t kPre = 2;                               // records per thread that leave with the first trip: 512 tiles of a contig
    ulonglong4 rec[kPre] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, ext[kPre] = {{0, 0, 0, 0}, {0, 0, 0, 0}};      // a tile's count + entries 0 .. 2; entries 3 .. 6
    auto load_recs = [&](uint32_t b_lo, uint32_t b_hi) {
#pragma unroll
        for (uint32_t i = 0; i < kPre; ++i) {
            const uint32_t b = b_lo + tid + 256u * i;
            rec[i].x = 0;
            if (b <= b_hi) { rec[i] = recs[b]; ext[i] = recs.second(b); }
        }
    };
    if (k_first < nK) load_recs(coff(k_first) / kCandPerBlock, (coff(k_first + 1) - 1) / kCandPerBlock);
    // What a multi-PS candidate needs (:85-105, :148-155) would be a SECOND, dependent trip -- its group summary sits at a slot that
    // only its PS word names.  The tile's own kC2Quota summary slots (3.5 KB: the candidates of this very tile put theirs there) and
    // every candidate's own columns leave with the first trip instead, whether or not anybody will look at them; only a summary
    // in the shared pool (a tile with more than 64 multi-PS candidates) is fetched once the slot is known.
    const uint32_t c_o0 = live ? p.cand_off[c] : 0u, c_o1 = live ? p.cand_off[c + 1] : 0u;
    const uint32_t c_svread = live ? p.cand_svread[c] : 0u, c_refread = live ? p.cand_refread[c] : 0u;
    {
        constexpr uint32_t kPairs = kC2Quota * kC2Words / 2;                // 448 eight-byte pieces
        const uint2 *src = reinterpret_cast<const uint2 *>(p.c2rec + (size_t)tile * kC2Quota * kC2Words);
        uint2 *dst = reinterpret_cast<uint2 *>(s_c2);
        const uint2 a0 = src[tid], a1 = src[tid + 256u < kPairs ? tid + 256u : 0u];
        dst[tid] = a0;
        if (tid + 256u < kPairs) dst[tid + 256u] = a1;
    }
    for (uint32_t k = lo; k < nK && coff(k) <= last; ++k) {
        const uint32_t c_lo = coff(k), c_hi = coff(k + 1);
        if (c_lo == c_hi) continue;
        const uint32_t b_lo = c_lo / kCandPerBlock, b_hi = (c_hi - 1) / kCandPerBlock;
        if (k != k_first) load_recs(b_lo, b_hi);               // (in flight while the table is cleared)
#pragma unroll
        for (uint32_t i = 0; i < kOwnTab / 256u; ++i) s_tab[tid + 256u * i] = kEmpty;
        if (tid == 0) s_nwin = 0;
        // the range of positions for which this tile will ask "which seed is nearest" (:106-111): candidates of this contig that
        // need the nearest seed outright, or may (a multi-PS candidate without a winner)
        const bool mine = live && c >= c_lo && c < c_hi;
        const bool asks = mine && code >= 4u && !(code & kDivZero);
        {
            const uint32_t mn = wave_max_u32(asks ? ~c_pos : 0u), mx = wave_max_u32(asks ? c_pos + 1u : 0u);     // (a position is below 2^32 - 1)
            if (lane == 0) { s_mm[wave][0] = mn; s_mm[wave][1] = mx; }
        }
        lds_barrier();
        STAMP(2, 1);
        // ---- every entry of the contig: into the hash set (is this PS a seed, :91), and -- in the same sweep, from the registers
        // it sits in -- what the tile needs to answer "which seed is nearest" for positions in [pmin, pmax]: the largest seed
        // <= pmin, the smallest seed >= pmax, and the seeds strictly inside (with position-ordered candidates a handful; repeats
        // from neighbouring tiles do not hurt a minimum).
        uint32_t pmin, pmax;
        {
            const uint32_t a0 = max(max(s_mm[0][0], s_mm[1][0]), max(s_mm[2][0], s_mm[3][0]));
            const uint32_t a1 = max(max(s_mm[0][1], s_mm[1][1]), max(s_mm[2][1], s_mm[3][1]));
            pmin = a1 ? ~a0 : 0u;                                           // (nobody asks: any range serves)
            pmax = a1 ? a1 - 1u : 0u;
        }
        uint32_t fresh = 0, leftp = 0, rinv = 0;
        static_assert(kOwnTab == 1u << 11, "the hash takes 11 bits");
        auto slot_of = [](uint32_t key) -> uint32_t { return (key * 2654435761u) >> (32u - 11u); };
        auto is_new = [&](uint32_t key) {                                   // a seed new to the set: counted, and looked at once
            ++fresh;
            if (key <= pmin) leftp = max(leftp, key + 1u);
            if (key >= pmax) rinv = max(rinv, ~key);
            if (key > pmin && key < pmax) {
                const uint32_t at = atomicAdd(&s_nwin, 1u);
                if (at < kOwnWin) s_win[at] = key;
            }
        };
        auto insert = [&](uint64_t e) {
            const uint32_t cc = (uint32_t)(e >> 32), key = (uint32_t)e;        // (a PS is at most 2^32 - 2: kEmpty is free)
            if (cc >= c_lo && cc < c_hi && own_insert_slow(s_tab, key, slot_of(key))) is_new(key);
        };
        // The thread's up to 2 x 7 entries in four groups -- a tile's first three, its next four --, each group only when some lane
        // of the wavefront has one (most tiles have two or three entries, most wavefronts no second tile): a group's first probes
        // leave together, what comes back is looked at without a branch per entry (the branch per entry cost this phase 1,800
        // instructions and 2 us: selects instead; only a collision and a seed inside the window are rare enough to branch for).
        auto group = [&](const uint64_t (&e)[4], uint32_t first, uint32_t n_in, uint32_t cnt) {
            uint32_t key[4], old[4];
            bool ok[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t cc = (uint32_t)(e[j] >> 32);
                key[j] = (uint32_t)e[j];                                    // (a PS is at most 2^32 - 2: kEmpty is free)
                ok[j] = j < n_in && first + j < cnt && cc >= c_lo && cc < c_hi;
                old[j] = kEmpty;
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j)
                if (ok[j]) old[j] = atomicCAS(&s_tab[slot_of(key[j])], kEmpty, key[j]);
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                bool nw = ok[j] && old[j] == kEmpty;
                if (ok[j] && old[j] != kEmpty && old[j] != key[j])          // (a collision: the probe sequence, behind a call)
                    nw = own_insert_slow(s_tab, key[j], (slot_of(key[j]) + 1u) & (kOwnTab - 1u)) != 0u;
                fresh += nw ? 1u : 0u;
                leftp = max(leftp, (nw && key[j] <= pmin) ? key[j] + 1u : 0u);
                rinv = max(rinv, (nw && key[j] >= pmax) ? ~key[j] : 0u);
                if (nw && key[j] > pmin && key[j] < pmax) {
                    const uint32_t at = atomicAdd(&s_nwin, 1u);
                    if (at < kOwnWin) s_win[at] = key[j];
                }
            }
        };
#pragma unroll
        for (uint32_t i = 0; i < kPre; ++i) {
            const uint32_t cnt = b_lo + tid + 256u * i <= b_hi ? (uint32_t)rec[i].x : 0u;
            if (__any(cnt > 0u)) {
                const uint64_t e[4] = {rec[i].y, rec[i].z, rec[i].w, 0ull};
                group(e, 0u, 3u, cnt);
            }
            if (__any(cnt > 3u)) {
                const uint64_t e[4] = {ext[i].x, ext[i].y, ext[i].z, ext[i].w};
                group(e, 3u, 4u, cnt);
            }
        }
        // A tile with more than seven entries -- candidates of a sparse type in stage A0's type-major order: a tile of 256 of them
        // spans a quarter of the contig and carries a hundred phase sets -- has the rest in its overflow slots: the WAVEFRONT takes
        // them, 64 entries per trip side by side (one lane walking them alone was a chain of a hundred dependent loads in EVERY
        // workgroup, 35 of the kernel's 45 us on those candidates).
        auto overflow = [&](uint32_t b, uint32_t cnt, uint32_t from) {
            // (a few entries more: the lane itself, four loads in flight at a time -- many lanes do that side by side; a wavefront
            // that took its lanes' tiles one after the other made a 782-tile genome's step 83 us where this takes 28)
            constexpr uint32_t kLong = 24;
            if (cnt > from && cnt <= kLong) {
                const uint64_t *ov = p.seed_ent + (size_t)b * kCandPerBlock;
#pragma unroll 1
                for (uint32_t j = from; j < cnt; j += 4) {
                    uint64_t e[4];
#pragma unroll
                    for (uint32_t t = 0; t < 4; ++t) e[t] = ov[min(j + t, cnt - 1u)];
#pragma unroll
                    for (uint32_t t = 0; t < 4; ++t)
                        if (j + t < cnt) insert(e[t]);
                }
            }
            unsigned long long todo = __ballot(cnt > kLong);
            while (todo) {
                const uint32_t h = (uint32_t)__ffsll((long long)todo) - 1u;
                todo &= todo - 1ull;
                const uint32_t b_h = (uint32_t)__builtin_amdgcn_readlane((int)b, h), cnt_h = (uint32_t)__builtin_amdgcn_readlane((int)cnt, h);
                const uint64_t *ov = p.seed_ent + (size_t)b_h * kCandPerBlock;
                for (uint32_t j = from + lane; j < cnt_h; j += 64u) insert(ov[j]);
            }
        };
#pragma unroll 1
        for (uint32_t i = 0; i < kPre; ++i) {
            const uint32_t b = b_lo + tid + 256u * i;
            overflow(b, b <= b_hi ? (uint32_t)rec[i].x : 0u, 7u);
        }
        // (a contig of more than 512 tiles: the others' records one more trip later)
        for (uint32_t b0 = b_lo + 256u * kPre; b0 <= b_hi; b0 += 256u) {
            const uint32_t b = b0 + tid;
            ulonglong4 r = {0, 0, 0, 0};
            if (b <= b_hi) r = recs[b];
            const uint32_t cnt = (uint32_t)r.x;
            if (cnt > 0) insert(r.y);
            if (cnt > 1) insert(r.z);
            if (cnt > 2) insert(r.w);
            overflow(b, cnt, 3u);
        }
        {
            const uint32_t l = wave_max_u32(leftp), r = wave_max_u32(rinv), o = wave_sum(fresh);
            if (lane == 0) { s_red[wave][0] = l; s_red[wave][1] = r; s_red[wave][2] = o; }
        }
        lds_barrier();
        STAMP(2, 2);
        const uint32_t u = s_red[0][2] + s_red[1][2] + s_red[2][2] + s_red[3][2];                   // distinct seeds (a full table loses some: over)
        const uint32_t Lp = max(max(s_red[0][0], s_red[1][0]), max(s_red[2][0], s_red[3][0]));       // left seed + 1, 0: none
        const uint32_t Ri = max(max(s_red[0][1], s_red[1][1]), max(s_red[2][1], s_red[3][1]));       // ~right seed, 0: none
        const uint32_t n_win = s_nwin;
        const bool over = u > keys_cap;                                     // (a full table: u == kOwnTab > keys_cap)
        const bool sorted = !over && n_win > win_cap;                       // candidates out of position order: the whole set, ascending
        if (sorted) own_sort_set(s_tab, s_key, s_one, s_part, u);      // (a call: rare, and its registers are not the hot path's)
        STAMP(2, 5);
        // ---- this tile's candidates of contig k ------------------------------------------------------
        // :107-111 -- ties go to the larger seed
        auto nearest = [&](uint32_t pos) -> uint32_t {
            if (sorted) return nearest_ps(s_one, u, pos);
            uint32_t best = 0;
            int64_t bd = -1;
            auto look = [&](uint32_t v) {
                const int64_t d = llabs((int64_t)pos - (int64_t)v);
                if (bd < 0 || d < bd || (d == bd && v > best)) { bd = d; best = v; }
            };
            if (Lp) look(Lp - 1u);
            if (Ri) look(~Ri);
            for (uint32_t i = 0; i < n_win; ++i) look(s_win[i]);
            return best;
        };
        auto member = [&](uint32_t key) -> bool {                           // (the table has empty slots: u <= keys_cap < kOwnTab)
            uint32_t h = (key * 2654435761u) >> (32u - 11u);
            for (;;) {
                const uint32_t v = s_tab[h];
                if (v == key) return true;
                if (v == kEmpty) return false;
                h = (h + 1u) & (kOwnTab - 1u);
            }
        };
        if (mine) {
            if (u == 0) {                                                   // :209-210
                if (code != 0) p.out_pred[c] = 0;
                if (ps_in != 0) p.out_ps[c] = 0;
            } else if (code >= 4u) {
                if (code & kDivZero) {                                      // :123 would raise
                    atomicOr(&p.status[0], 1u);
                    p.out_pred[c] = 0;
                    p.out_ps[c] = 0;
                } else if (over) {
                    const OwnArgs a = {recs, p.seed_ent, p.read_tag, p.cand_pos, p.cand_off, p.cand_svread, p.cand_refread, p.mark_read, p.c2rec,
                                       p.out_ps, p.status, p.out_pred, c_lo, c_hi};
                    finalize_candidate_walk(a, c, code, ps_in);
                } else if (code & kClass2Slow) {                            // no group summary: the vote from the marks (:85-105)
                    Vote v;
                    uint32_t ps;
                    const uint32_t mb = p.cand_off[c], me_ = p.cand_off[c + 1];
                    own_vote_from_marks(p.mark_read, p.read_tag, mb, me_, s_tab, &v, &ps);
                    if (v.hap1 == 0 && v.hap2 == 0) ps = nearest(c_pos);    // :106
                    p.out_pred[c] = (uint8_t)decide(2, v, me_ - mb, p.cand_svread[c], p.cand_refread[c]);
                    p.out_ps[c] = ps;
                } else if (code & kClass2) {                                // :85-105, :148-155 from the summary that is already here
                    uint32_t w[kC2Words];
                    const uint32_t own0 = tile * kC2Quota;
                    if (ps_in >= own0 && ps_in < own0 + kC2Quota) {
#pragma unroll
                        for (int i = 0; i < kC2Words; ++i) w[i] = s_c2[(ps_in - own0) * kC2Words + i];
                    } else {                                                // (a slot of the shared pool)
                        const uint32_t *rec2 = p.c2rec + (size_t)ps_in * kC2Words;
#pragma unroll
                        for (int i = 0; i < kC2Words; ++i) w[i] = rec2[i];
                    }
                    Vote v = {0, 0, 0, 0, 0, 0};
                    uint32_t ps = 0, best = 0;
                    v.allhap = w[0];
                    const uint32_t ng = w[1];
#pragma unroll
                    for (int g = 0; g < kC2Groups; ++g) {
                        if ((uint32_t)g < ng && w[2 + 6 * g + 1] > best && member(w[2 + 6 * g])) {
                            best = w[2 + 6 * g + 1];
                            ps = w[2 + 6 * g];
                            v.hap1 = w[2 + 6 * g + 2]; v.hap2 = w[2 + 6 * g + 3];
                            v.t1 = w[2 + 6 * g + 4]; v.t2 = w[2 + 6 * g + 5];
                            v.hap0 = v.allhap - v.hap1 - v.hap2;           // only with a winner (:105)
                        }
                    }
                    if (v.hap1 == 0 && v.hap2 == 0) ps = nearest(c_pos);    // :106
                    p.out_pred[c] = (uint8_t)decide(2, v, c_o1 - c_o0, c_svread, c_refread);
                    p.out_ps[c] = ps;
                } else {                                                    // kNeedNearest
                    p.out_pred[c] = (uint8_t)(code & 3u);
                    p.out_ps[c] = nearest(c_pos);
                }
            }
        }
        STAMP(2, 6);
        __syncthreads();                                                    // the set is reused
    }
    if (!DYN) break;
  }
}

// plan time: ctg_start[c] = 1 for the first candidate of every non-empty contig
__global__ void plan_mark_starts(const uint32_t *ctg_off, uint32_t K, uint8_t *ctg_start)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K && ctg_off[k] < ctg_off[k + 1]) ctg_start[ctg_off[k]] = 1;
}

// device-planned runs, everything the plan needs in ONE launch: the workspace's copy of the contig offsets, the per-contig
// seed counts and the status words zeroed, and the contig of the first candidate of every 256-candidate block (binary search)
__global__ void plan_device(const uint32_t *ctg_off, uint32_t K, uint32_t B, uint32_t *ctg_off_copy, uint32_t *n_one_and_status,
                            uint32_t *blk_ctg)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= K) ctg_off_copy[b] = ctg_off[b];
    if (b < K + 8) n_one_and_status[b] = 0;
    if (b >= B) return;
    const uint32_t c = b * kCandPerBlock;
    uint32_t lo = 0, hi = K;                                   // the last k with ctg_off[k] <= c
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ctg_off[mid] <= c) lo = mid; else hi = mid;
    }
    blk_ctg[b] = lo;
}

// the same from the candidates' contig column (sorted by contig): the contig offsets by binary search, the tiles' contigs
// straight from the column
__global__ void plan_device_contigs(const uint16_t *cand_contig, const uint32_t *n_cands, uint32_t K, uint32_t B,
                                    uint32_t *ctg_off_copy, uint32_t *n_one_and_status, uint32_t *blk_ctg)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t N = *n_cands;
    if (t <= K) {
        uint32_t lo = 0, hi = N;                               // first candidate whose contig is >= t
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (cand_contig[mid] < t) lo = mid + 1; else hi = mid;
        }
        ctg_off_copy[t] = t == K ? N : lo;
    }
    if (t < K + 8) n_one_and_status[t] = 0;
    if (t < B) {
        const uint32_t c = t * kCandPerBlock;
        blk_ctg[t] = c < N ? (uint32_t)cand_contig[c] : 0u;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------

}  // namespace

thread_local std::string duet_g_last_error;

namespace {

inline int fail(duet_ctx *ctx, int code, const std::string &msg) { return duet_fail(ctx, code, msg); }
inline int reserve(duet_ctx *ctx, DevBuf &b, size_t bytes) { return duet_reserve(ctx, b, bytes); }

// (re)build the workspace for this contig layout; a no-op when it matches the cached plan.  The rebuild is
// ordered on `stream` (uploads from a pinned staging block, memsets, one small kernel): the device is only
// synchronised when a buffer has to grow or the context moves to another stream.
// group-summary slots: kC2Quota per tile and a pool of an eighth of the candidates for the tiles that need more
size_t c2_slots(uint32_t B, uint32_t C) { return (size_t)B * kC2Quota + (C / 8u > 256u ? C / 8u : 256u); }

int ensure_plan(duet_ctx *ctx, const duet_ef_problem *pr, hipStream_t stream)
{
    const uint32_t K = pr->n_contigs, C = pr->n_cands;
    if (ctx->plan_C == C && ctx->plan_off.size() == (size_t)K + 1 && ctx->plan_stream == stream &&
        memcmp(ctx->plan_off.data(), pr->cand_ctg_off, sizeof(uint32_t) * (K + 1)) == 0)
        return DUET_OK;
    const uint32_t B = (C + kCandPerBlock - 1) / kCandPerBlock;
    const size_t small_words = (size_t)(K + 1) + K + 8 + (size_t)B + 16 + (size_t)B * 16;      // (... 64-byte tile records, 64-byte aligned)
    const size_t want[6] = {small_words * 4, C, (size_t)B * kCandPerBlock * 8, ((size_t)C + K + 1) * 4, ((size_t)C + K + 1) * 4,
                            c2_slots(B, C) * kC2Words * 4};
    DevBuf *bufs[6] = {&ctx->ws_small, &ctx->ws_start, &ctx->ws_ent, &ctx->ws_one, &ctx->ws_tmp, &ctx->ws_c2};
    bool grow = ctx->plan_stream != stream;
    for (int i = 0; i < 6; ++i) grow = grow || want[i] > bufs[i]->cap;
    const size_t stage_bytes = ((size_t)K + 1 + B) * 4;
    grow = grow || stage_bytes > ctx->plan_stage_cap;
    int rc;
    if (grow) {
        // buffers of the previous plan may still be in use by queued work
        HIP_TRY(ctx, hipDeviceSynchronize());
        for (int i = 0; i < 6; ++i)
            if ((rc = reserve(ctx, *bufs[i], want[i]))) return rc;
        if (stage_bytes > ctx->plan_stage_cap) {
            if (ctx->plan_stage) HIP_TRY(ctx, hipHostFree(ctx->plan_stage));
            ctx->plan_stage = nullptr;
            ctx->plan_stage_cap = 0;
            HIP_TRY(ctx, hipHostMalloc((void **)&ctx->plan_stage, stage_bytes + stage_bytes / 4 + 4096, hipHostMallocDefault));
            ctx->plan_stage_cap = stage_bytes + stage_bytes / 4 + 4096;
        }
    } else {
        HIP_TRY(ctx, hipEventSynchronize(ctx->plan_ev));        // the staging block's previous upload has been read
    }
    uint32_t *w = (uint32_t *)ctx->ws_small.ptr;
    ctx->d_ctg_off = w;            w += K + 1;
    ctx->d_n_one = w;              w += K;
    ctx->d_status = w;             w += 8;
    ctx->d_blk_ctg = w;            w += B;
    w += ((uintptr_t)w & 63) ? (64 - ((uintptr_t)w & 63)) / 4 : 0;        // 32-byte aligned tile records
    ctx->d_blk_cnt = w;            // B records of 4 x u64
    // staging: ctg_off, then the contig of the first candidate of every 256-candidate block
    uint32_t *h_off = ctx->plan_stage, *h_blk = ctx->plan_stage + (K + 1);
    memcpy(h_off, pr->cand_ctg_off, sizeof(uint32_t) * (K + 1));
    {
        uint32_t k = 0;
        for (uint32_t b = 0; b < B; ++b) {
            const uint32_t c = b * kCandPerBlock;
            while (c >= pr->cand_ctg_off[k + 1]) ++k;
            h_blk[b] = k;
        }
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ctg_off, h_off, sizeof(uint32_t) * (K + 1), hipMemcpyHostToDevice, stream));
    if (B) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_blk_ctg, h_blk, sizeof(uint32_t) * B, hipMemcpyHostToDevice, stream));
    HIP_TRY(ctx, hipEventRecord(ctx->plan_ev, stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_n_one, 0, sizeof(uint32_t) * ((size_t)K + 8), stream));
    if (C) HIP_TRY(ctx, hipMemsetAsync(ctx->ws_start.ptr, 0, C, stream));
    hipLaunchKernelGGL(plan_mark_starts, dim3((K + 255) / 256), dim3(256), 0, stream, ctx->d_ctg_off, K,
                       (uint8_t *)ctx->ws_start.ptr);
    HIP_TRY(ctx, hipGetLastError());
    ctx->ef_seeds_stale = false;                               // (the saved kernel arguments describe the previous workspace)
    ctx->plan_off.assign(pr->cand_ctg_off, pr->cand_ctg_off + K + 1);
    ctx->plan_C = C;
    ctx->plan_stream = stream;
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

const char *duet_last_error(const duet_ctx *ctx) { return ctx ? ctx->err.c_str() : duet_g_last_error.c_str(); }

duet_ctx *duet_ctx_create(int device_id)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        duet_g_last_error = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return nullptr;
    }
    if (device_id < 0 || device_id >= n) {
        duet_g_last_error = "device id out of range";
        return nullptr;
    }
    if ((e = hipSetDevice(device_id)) != hipSuccess) {
        duet_g_last_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return nullptr;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) {
        duet_g_last_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return nullptr;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        duet_g_last_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return nullptr;
    }
    duet_ctx *ctx = new duet_ctx();
    ctx->device = device_id;
    if (const char *t = getenv("DUET_EF_HEAVY_T")) ctx->ef_heavy_t = (uint32_t)strtoul(t, nullptr, 10);   // diagnostic: sweeps of the threshold
    e = hipEventCreateWithFlags(&ctx->cl_fork, hipEventDisableTiming);
    for (int i = 0; i < 3 && e == hipSuccess; ++i) {
        e = hipStreamCreateWithFlags(&ctx->cl_side[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->cl_join[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->plan_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->rx_dtot, 8 * 2048 * sizeof(uint32_t));        // kDtotCopies x 256 (key sort) / x 2048 (record sort)
    if (e == hipSuccess) e = hipMemset(ctx->rx_dtot, 0, 8 * 2048 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->cl_flags, 64);
    if (e == hipSuccess) e = hipMemset(ctx->cl_flags, 0, 64);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_untagged, 64);
    if (e == hipSuccess) e = hipMemset(ctx->d_untagged, 0xFF, 64);                                  // the tag word of a mark without a tag
    if (e != hipSuccess || (e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        duet_g_last_error = std::string("context resources: ") + hipGetErrorString(e);
        duet_ctx_destroy(ctx);                               // releases whatever was created
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
    for (DevBuf &b : ctx->cl_ws) if (b.ptr) (void)hipFree(b.ptr);
    for (DevBuf &b : ctx->cl_in) if (b.ptr) (void)hipFree(b.ptr);
    for (DevBuf &b : ctx->cl_out) if (b.ptr) (void)hipFree(b.ptr);
    for (DevBuf &b : ctx->sv_ws) if (b.ptr) (void)hipFree(b.ptr);
    for (DevBuf &b : ctx->sv_in) if (b.ptr) (void)hipFree(b.ptr);
    for (DevBuf &b : ctx->sv_out) if (b.ptr) (void)hipFree(b.ptr);
    for (DevBuf &b : ctx->rows_ws) if (b.ptr) (void)hipFree(b.ptr);
    for (DevBuf &b : ctx->rows_in) if (b.ptr) (void)hipFree(b.ptr);
    if (ctx->eval_ws.ptr) (void)hipFree(ctx->eval_ws.ptr);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->rx_dtot) (void)hipFree(ctx->rx_dtot);
    if (ctx->d_untagged) (void)hipFree(ctx->d_untagged);
    if (ctx->cl_flags) (void)hipFree(ctx->cl_flags);
    for (int i = 0; i < 3; ++i) {
        if (ctx->cl_side[i]) (void)hipStreamDestroy(ctx->cl_side[i]);
        if (ctx->cl_join[i]) (void)hipEventDestroy(ctx->cl_join[i]);
    }
    if (ctx->cl_fork) (void)hipEventDestroy(ctx->cl_fork);
    if (ctx->sv_depth_off_ev) (void)hipEventDestroy(ctx->sv_depth_off_ev);
    if (ctx->plan_ev) (void)hipEventDestroy(ctx->plan_ev);
    if (ctx->plan_stage) (void)hipHostFree(ctx->plan_stage);
    delete ctx;
}

int duet_ctx_set_profiling(duet_ctx *ctx, int enabled)
{
    if (!ctx) return fail(nullptr, DUET_ERR_INVALID, "null context");
    if (ctx->ev_used && ctx->ev_mode != (enabled == 3 ? 1 : enabled))
        return fail(ctx, DUET_ERR_INVALID, "collect the pending profile before changing the mode");
    ctx->profiling = enabled < 0 ? 0 : (enabled > 3 ? 3 : enabled);
    ctx->prof_tick = 0;
    return DUET_OK;
}

#ifdef DUET_STAMPS
DUET_API int duet_dbg_stamps(duet_ctx *ctx, int enable, unsigned long long *host_out)
{
    const size_t bytes = (size_t)6 * 65536 * 8 * 8;        // [0..2] the E/F kernels, [3..5] stage A0's agglomeration chains (duet_cluster.hip)
    if (enable && !ctx->d_stamps) { if (hipMalloc((void **)&ctx->d_stamps, bytes) != hipSuccess) return -1; }
    if (enable) { (void)hipMemset(ctx->d_stamps, 0, bytes); return 0; }
    if (ctx->d_stamps && host_out) { (void)hipDeviceSynchronize(); (void)hipMemcpy(host_out, ctx->d_stamps, bytes, hipMemcpyDeviceToHost); }
    return 0;
}
#endif

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
    memset(&p, 0, sizeof(p));
    p.K = pr->n_contigs; p.C = pr->n_cands; p.M = pr->n_marks;
    p.read_tag = pr->n_reads ? pr->read_tag : (const uint64_t *)ctx->d_n_one;   // always >= 1 readable word
    p.n_reads = pr->n_reads; p.untagged = ctx->d_untagged;
    p.cand_pos = pr->cand_pos; p.cand_svlen = pr->cand_svlen; p.cand_svread = pr->cand_svread;
    p.cand_refread = pr->cand_refread; p.cand_gt_ok = pr->cand_gt_ok;
    p.cand_off = pr->cand_off; p.mark_read = pr->mark_read;
    p.svlen_thres = pr->svlen_thres; p.suppread_thres = pr->suppread_thres;
    p.ctg_off = ctx->d_ctg_off; p.ctg_start = (const uint8_t *)ctx->ws_start.ptr; p.blk_ctg = ctx->d_blk_ctg;
    p.blk_rec = (uint64_t *)ctx->d_blk_cnt; p.seed_ent = (uint64_t *)ctx->ws_ent.ptr;
    p.one_cap = pr->n_cands + pr->n_contigs;
    p.n_small = pr->n_contigs <= (uint32_t)kSmallK ? pr->n_contigs : 0;
    for (uint32_t k = 0; k <= p.n_small && p.n_small; ++k) p.ctg_small[k] = pr->cand_ctg_off[k];
    p.onebuf = (uint32_t *)ctx->ws_one.ptr; p.tmpbuf = (uint32_t *)ctx->ws_tmp.ptr;
    p.n_one = ctx->d_n_one; p.c2rec = (uint32_t *)ctx->ws_c2.ptr; p.status = ctx->d_status;
    p.c2_fixed = ((p.C + kCandPerBlock - 1) / kCandPerBlock) * kC2Quota;
    p.c2_cap = (uint32_t)c2_slots((p.C + kCandPerBlock - 1) / kCandPerBlock, p.C);
    p.out_pred = out_pred; p.out_ps = out_ps;
    p.dbg = ctx->dbg;
    p.heavy_t = (ctx->dbg & DUET_DBG_EF_HEAVY_ALL) ? 0u : ((ctx->dbg & DUET_DBG_EF_HEAVY_OFF) ? 0xFFFFFFFFu : ctx->ef_heavy_t);
    p.stamps = ctx->d_stamps;

    // Profiling: the start/stop events ride on the kernel's own dispatch packet (hipExtLaunchKernelGGL),
    // so hipEventElapsedTime is the kernel's execution time as rocprofv3 --kernel-trace reports it.
    // 6 events per run: {start, stop} x {classify, seed_sort, finalize}; mode 1 uses the first pair only.
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int prof = ctx->profiling;
    if (prof == 3) prof = (ctx->prof_tick++ % 8u == 0) ? 1 : 0;       // sampled: every 8th run carries the two events
    if (prof) {
        ctx->ev_mode = prof;                                        // 1 or 2 (a sampled run records like mode 1)
        while (ctx->ev_pool.size() < ctx->ev_used + 6) {
            hipEvent_t e;
            HIP_TRY(ctx, hipEventCreate(&e));
            ctx->ev_pool.push_back(e);
        }
        for (int i = 0; i < (prof == 2 ? 6 : 2); ++i) ev[i] = ctx->ev_pool[ctx->ev_used + i];
        ctx->ev_used += 6;
    }
    const uint32_t blocks = (pr->n_cands + kCandPerBlock - 1) / kCandPerBlock;
    // two launches where the problem is small (round 6): ef_finalize_own builds every tile's seed set itself, no ef_seed_sort
    // (the diagnostic bits that select a variant of ef_seed_sort / ef_finalize select the three launches with it)
    const uint32_t three = DUET_DBG_EF_OWN_OFF | DUET_DBG_EF_FIN_TPB2 | DUET_DBG_EF_FIN_TPB4 | DUET_DBG_EF_NO_SEED_HASH;
    const bool own = p.n_small && !(ctx->dbg & three) && (blocks <= kOwnMaxTiles || (ctx->dbg & DUET_DBG_EF_OWN_ALL));
    if (prof) ctx->ev_kmask.push_back(own ? 0x5 : 0x7);
    if (((uintptr_t)pr->mark_read & 15) == 0)
        hipExtLaunchKernelGGL(ef_classify<true>, dim3(blocks), dim3(kCandPerBlock), 0, stream, ev[0], ev[1], 0, p);
    else
        hipExtLaunchKernelGGL(ef_classify<false>, dim3(blocks), dim3(kCandPerBlock), 0, stream, ev[0], ev[1], 0, p);
    if (own) {
        hipExtLaunchKernelGGL(ef_finalize_own<false>, dim3(blocks), dim3(256), 0, stream, ev[4], ev[5], 0, p);
        ctx->ef_last_params.assign((const unsigned char *)&p, (const unsigned char *)&p + sizeof(p));
        ctx->ef_last_stream = stream;
        ctx->ef_seeds_stale = true;
    } else {
        ctx->ef_seeds_stale = false;
        hipExtLaunchKernelGGL(ef_seed_sort, dim3(pr->n_contigs), dim3(kSortThreads), 0, stream, ev[2], ev[3], 0, p);
        // Tiles per workgroup: the per-workgroup costs (launch, the tile's contigs, the seeds into LDS) once per 512 / 1024
        // candidates where there are enough of them to fill the chip anyway -- measured (tools/gpu/r4_fin.sh), 1 / 2 / 4 tiles:
        // 1e5 candidates 6.3 / 8.5 / 11.2 us, 2e6 20.6 / 18.4 / 18.8 us, 2e7 127 / 91 / 75 us
        const uint32_t C = pr->n_cands;
        int tpb = (!p.n_small || C < 1000000u) ? 1 : (C < 8000000u ? 2 : 4);
        if (p.n_small && (ctx->dbg & DUET_DBG_EF_FIN_TPB2)) tpb = 2;
        if (p.n_small && (ctx->dbg & DUET_DBG_EF_FIN_TPB4)) tpb = 4;
        if (tpb == 4) hipExtLaunchKernelGGL((ef_finalize<false, 4>), dim3((C + 1023) / 1024), dim3(256), 0, stream, ev[4], ev[5], 0, p);
        else if (tpb == 2) hipExtLaunchKernelGGL((ef_finalize<false, 2>), dim3((C + 511) / 512), dim3(256), 0, stream, ev[4], ev[5], 0, p);
        else hipExtLaunchKernelGGL((ef_finalize<false, 1>), dim3((C + 255) / 256), dim3(256), 0, stream, ev[4], ev[5], 0, p);
    }
    HIP_TRY(ctx, hipGetLastError());
    ctx->pending_check = true;
    return DUET_OK;
}

}  // extern "C"

// Device-planned run (used by duet_svim_phase_device): the candidate count and the contig offsets exist only on the
// device (d_n_cands, d_ctg_off[K+1]); every buffer and grid is sized for c_max candidates and the kernels read the real
// count.  Nothing here waits for the device.
//
// The workspace first: a producer that knows the candidates' contigs as it writes them (stage A0's cl_emit) fills in the
// plan itself -- the contig offsets [K+1], and K + 8 zeroed words (seeds per contig, status) behind them -- and no plan
// kernel runs (it cost the fused pipeline 6 us at 1.0 M marks, 9 us at 2e7, for K + 1 numbers).
int duet_ef_plan_on_device_prepare(duet_ctx *ctx, uint32_t K, uint32_t c_max, hipStream_t stream, uint32_t **ctg_off_out,
                                   uint32_t **zero_out)
{
    const uint32_t C = c_max;
    const uint32_t B = (C + kCandPerBlock - 1) / kCandPerBlock;
    const size_t small_words = (size_t)(K + 1) + K + 8 + (size_t)B + 16 + (size_t)B * 16;      // (... 64-byte tile records, 64-byte aligned)
    const size_t want[6] = {small_words * 4, C, (size_t)B * kCandPerBlock * 8, ((size_t)C + K + 1) * 4, ((size_t)C + K + 1) * 4,
                            c2_slots(B, C) * kC2Words * 4};
    DevBuf *bufs[6] = {&ctx->ws_small, &ctx->ws_start, &ctx->ws_ent, &ctx->ws_one, &ctx->ws_tmp, &ctx->ws_c2};
    bool grow = ctx->plan_stream != stream;
    for (int i = 0; i < 6; ++i) grow = grow || want[i] > bufs[i]->cap;
    int rc;
    if (grow) {
        HIP_TRY(ctx, hipDeviceSynchronize());                  // buffers of the previous plan may still be in use
        for (int i = 0; i < 6; ++i)
            if ((rc = reserve(ctx, *bufs[i], want[i]))) return rc;
    }
    uint32_t *w = (uint32_t *)ctx->ws_small.ptr;
    ctx->d_ctg_off = w;            w += K + 1;
    ctx->d_n_one = w;              w += K;
    ctx->d_status = w;             w += 8;
    ctx->d_blk_ctg = w;            w += B;
    w += ((uintptr_t)w & 63) ? (64 - ((uintptr_t)w & 63)) / 4 : 0;
    ctx->d_blk_cnt = w;
    ctx->plan_off.clear();                                     // the cached host-side plan no longer describes the workspace
    ctx->plan_C = 0;
    ctx->plan_stream = stream;
    if (ctg_off_out) *ctg_off_out = ctx->d_ctg_off;
    if (zero_out) *zero_out = ctx->d_n_one;
    return DUET_OK;
}

int duet_ef_run_planned_on_device(duet_ctx *ctx, const duet_ef_problem *pr, uint32_t c_max, const uint32_t *d_n_cands,
                                  const uint32_t *d_ctg_off, const uint16_t *d_cand_contig, uint8_t *out_pred, uint32_t *out_ps,
                                  hipStream_t stream, bool planned)
{
    const uint32_t K = pr->n_contigs, C = c_max;
    if (K == 0 || C == 0) return DUET_OK;
    const uint32_t B = (C + kCandPerBlock - 1) / kCandPerBlock;
    int rc;
    if ((rc = duet_ef_plan_on_device_prepare(ctx, K, c_max, stream, nullptr, nullptr))) return rc;
    if (!planned) {
        const uint32_t nthr = B > K + 8 ? B : K + 8;
        if (d_cand_contig)      // no contig offsets yet: they come out of the same launch
            hipLaunchKernelGGL(plan_device_contigs, dim3((nthr + 255) / 256), dim3(256), 0, stream, d_cand_contig, d_n_cands, K, B,
                               ctx->d_ctg_off, ctx->d_n_one, ctx->d_blk_ctg);
        else
            hipLaunchKernelGGL(plan_device, dim3((nthr + 255) / 256), dim3(256), 0, stream, d_ctg_off, K, B, ctx->d_ctg_off,
                               ctx->d_n_one, ctx->d_blk_ctg);
    }

    Params p;
    memset(&p, 0, sizeof(p));
    p.K = K; p.C = C; p.M = pr->n_marks;
    p.dyn_c = d_n_cands;
    p.cand_contig = d_cand_contig;
    p.read_tag = pr->n_reads ? pr->read_tag : (const uint64_t *)ctx->d_n_one;
    p.n_reads = pr->n_reads; p.untagged = ctx->d_untagged;
    p.cand_pos = pr->cand_pos; p.cand_svlen = pr->cand_svlen; p.cand_svread = pr->cand_svread;
    p.cand_refread = pr->cand_refread; p.cand_gt_ok = pr->cand_gt_ok;
    p.cand_off = pr->cand_off; p.mark_read = pr->mark_read;
    p.svlen_thres = pr->svlen_thres; p.suppread_thres = pr->suppread_thres;
    p.ctg_off = ctx->d_ctg_off; p.ctg_start = (const uint8_t *)ctx->ws_start.ptr; p.blk_ctg = ctx->d_blk_ctg;
    p.blk_rec = (uint64_t *)ctx->d_blk_cnt; p.seed_ent = (uint64_t *)ctx->ws_ent.ptr;
    p.one_cap = C + K;
    p.n_small = 0;
    p.onebuf = (uint32_t *)ctx->ws_one.ptr; p.tmpbuf = (uint32_t *)ctx->ws_tmp.ptr;
    p.n_one = ctx->d_n_one; p.c2rec = (uint32_t *)ctx->ws_c2.ptr; p.status = ctx->d_status;
    p.c2_fixed = ((p.C + kCandPerBlock - 1) / kCandPerBlock) * kC2Quota;
    p.c2_cap = (uint32_t)c2_slots((p.C + kCandPerBlock - 1) / kCandPerBlock, p.C);
    p.out_pred = out_pred; p.out_ps = out_ps;
    p.dbg = ctx->dbg;
    p.heavy_t = (ctx->dbg & DUET_DBG_EF_HEAVY_ALL) ? 0u : ((ctx->dbg & DUET_DBG_EF_HEAVY_OFF) ? 0xFFFFFFFFu : ctx->ef_heavy_t);
    p.stamps = ctx->d_stamps;
    // the kernels stride over the real tiles: an eighth of the bound's tiles (SV-like data: ~10 marks per candidate), at least 512
    const uint32_t G = B < 512u ? B : (B / 8u > 512u ? B / 8u : 512u);
    if (((uintptr_t)pr->mark_read & 15) == 0)
        hipLaunchKernelGGL((ef_classify<true, true>), dim3(G), dim3(kCandPerBlock), 0, stream, p);
    else
        hipLaunchKernelGGL((ef_classify<false, true>), dim3(G), dim3(kCandPerBlock), 0, stream, p);
    // The own-set finalize (ef_finalize_own) is NOT taken here by default: stage A0 hands its candidates over by type, then position,
    // and a tile of a sparse type (DUP, INV: a percent of the candidates each) spans a quarter of the contig and carries a hundred
    // seed entries that EVERY tile then reads -- measured on configs[1]'s marks: 35.6 us against 34.3 for E/F alone, 270.9 against
    // 269.1 us for the pipeline (in position order the same candidates take 23.4 against 28.8).  DUET_DBG_EF_OWN_ALL takes it
    // (tests/test_gpu_fused.py runs the pipeline both ways).
    const uint32_t three = DUET_DBG_EF_OWN_OFF | DUET_DBG_EF_FIN_TPB2 | DUET_DBG_EF_FIN_TPB4 | DUET_DBG_EF_NO_SEED_HASH;
    if (K <= (uint32_t)kSmallK && !(ctx->dbg & three) && (ctx->dbg & DUET_DBG_EF_OWN_ALL)) {
        hipLaunchKernelGGL(ef_finalize_own<true>, dim3(G), dim3(256), 0, stream, p);
    } else {
    hipLaunchKernelGGL(ef_seed_sort, dim3(K), dim3(kSortThreads), 0, stream, p);
    // (two tiles per workgroup where the bound says millions of candidates: the per-workgroup round trips once per 512)
    if (d_cand_contig && (C >= 8000000u || (ctx->dbg & DUET_DBG_EF_FIN_TPB2)))
        hipLaunchKernelGGL((ef_finalize<true, 2>), dim3(G), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL(ef_finalize<true>, dim3(G), dim3(256), 0, stream, p);
    }
    HIP_TRY(ctx, hipGetLastError());
    ctx->pending_check = true;
    return DUET_OK;
}

extern "C" {

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
    const size_t runs = ctx->ev_used / 6;
    double acc[DUET_N_KERNELS] = {0, 0, 0}, tot = 0;
    const int nk = ctx->ev_mode == 2 ? DUET_N_KERNELS : 1;
    for (size_t r = 0; r < runs; ++r) {
        hipEvent_t *ev = &ctx->ev_pool[6 * r];
        const unsigned mask = r < ctx->ev_kmask.size() ? ctx->ev_kmask[r] : 0x7u;      // (a kernel that did not run has no events: 0 us)
        HIP_TRY(ctx, hipEventSynchronize(ev[2 * nk - 1]));
        for (int i = 0; i < nk; ++i) {
            if (!((mask >> i) & 1u)) continue;
            float ms = 0;
            HIP_TRY(ctx, hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
            acc[i] += ms;
        }
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ev[0], ev[2 * nk - 1]));
        tot += ms;
    }
    if (runs) {
        for (int i = 0; i < DUET_N_KERNELS; ++i) stats->kernel_ms[i] = (float)(acc[i] / runs);
        stats->total_ms = (float)(tot / runs);
    }
    ctx->ev_kmask.clear();
    stats->n_profiled_runs = (uint32_t)runs;
    ctx->ev_used = 0;
    return DUET_OK;
}

int duet_ef_get_seed_ps(duet_ctx *ctx, uint32_t contig, uint32_t *out, uint32_t cap)
{
    if (!ctx) return fail(nullptr, DUET_ERR_INVALID, "null context");
    if (contig + 1 >= ctx->plan_off.size()) return fail(ctx, DUET_ERR_INVALID, "contig out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc_m = duet_ef_materialise_seeds(ctx);
    if (rc_m) return rc_m;
    HIP_TRY(ctx, hipDeviceSynchronize());
    uint32_t n = 0;
    HIP_TRY(ctx, hipMemcpy(&n, ctx->d_n_one + contig, 4, hipMemcpyDeviceToHost));
    const uint32_t take = n < cap ? n : cap;
    if (take && out)
        HIP_TRY(ctx, hipMemcpy(out, (uint32_t *)ctx->ws_one.ptr + ctx->plan_off[contig] + contig + 1, (size_t)take * 4,
                               hipMemcpyDeviceToHost));
    return (int)n;
}

}  // extern "C"

// the ascending seed arrays (onebuf, n_one) of the last run, when that run was the two-launch one: ef_seed_sort on its seed entries
int duet_ef_materialise_seeds(duet_ctx *ctx)
{
    if (!ctx->ef_seeds_stale || ctx->ef_last_params.size() != sizeof(Params)) return DUET_OK;
    Params p;
    memcpy(&p, ctx->ef_last_params.data(), sizeof(p));
    hipLaunchKernelGGL(ef_seed_sort, dim3(p.K), dim3(kSortThreads), 0, ctx->ef_last_stream, p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->ef_last_stream));
    ctx->ef_seeds_stale = false;
    return DUET_OK;
}

int duet_ef_validate(duet_ctx *ctx, const duet_ef_problem *pr) { return validate(ctx, pr, (const void *)1, (const void *)1); }

// host arrays of *pr -> the context's staging buffers (asynchronous on `s`); *d = the same problem with device pointers.
// Also reserves the result buffers h_out[0] (pred) / h_out[1] (ps).
int duet_ef_upload(duet_ctx *ctx, const duet_ef_problem *pr, duet_ef_problem *d, hipStream_t s)
{
    const uint32_t C = pr->n_cands, M = pr->n_marks, R = pr->n_reads;
    const void *src[8] = {pr->read_tag, pr->cand_pos, pr->cand_svlen, pr->cand_svread, pr->cand_refread,
                          pr->cand_gt_ok, pr->cand_off, pr->mark_read};
    const size_t bytes[8] = {(size_t)R * 8, (size_t)C * 4, (size_t)C * 4, (size_t)C * 4, (size_t)C * 4,
                             (size_t)C, ((size_t)C + 1) * 4, (size_t)M * 4};
    int rc;
    for (int i = 0; i < 8; ++i) {
        if ((rc = reserve(ctx, ctx->h_in[i], bytes[i] ? bytes[i] : 16))) return rc;
        if (bytes[i]) HIP_TRY(ctx, hipMemcpyAsync(ctx->h_in[i].ptr, src[i], bytes[i], hipMemcpyHostToDevice, s));
    }
    if ((rc = reserve(ctx, ctx->h_out[0], C ? C : 16))) return rc;
    if ((rc = reserve(ctx, ctx->h_out[1], C ? (size_t)C * 4 : 16))) return rc;
    *d = *pr;
    d->read_tag = (const uint64_t *)ctx->h_in[0].ptr;
    d->cand_pos = (const uint32_t *)ctx->h_in[1].ptr;
    d->cand_svlen = (const uint32_t *)ctx->h_in[2].ptr;
    d->cand_svread = (const uint32_t *)ctx->h_in[3].ptr;
    d->cand_refread = (const uint32_t *)ctx->h_in[4].ptr;
    d->cand_gt_ok = (const uint8_t *)ctx->h_in[5].ptr;
    d->cand_off = (const uint32_t *)ctx->h_in[6].ptr;
    d->mark_read = (const uint32_t *)ctx->h_in[7].ptr;
    return DUET_OK;
}

extern "C" {

int duet_ef_run_host(duet_ctx *ctx, const duet_ef_problem *pr, uint8_t *out_pred, uint32_t *out_ps, duet_ef_stats *stats)
{
    int rc = validate(ctx, pr, out_pred, out_ps);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->algorithmic_bytes = 12ull * pr->n_marks + 27ull * pr->n_cands + 8ull * pr->n_reads;
    }
    const uint32_t C = pr->n_cands;
    if (C == 0) return DUET_OK;
    hipStream_t s = ctx->own_stream;
    duet_ef_problem d;
    if ((rc = duet_ef_upload(ctx, pr, &d, s))) return rc;
    if ((rc = duet_ef_run_device(ctx, &d, (uint8_t *)ctx->h_out[0].ptr, (uint32_t *)ctx->h_out[1].ptr, s))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out_pred, ctx->h_out[0].ptr, C, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(out_ps, ctx->h_out[1].ptr, (size_t)C * 4, hipMemcpyDeviceToHost, s));
    if ((rc = duet_ef_check(ctx, s))) return rc;
    if (stats) {
        if ((rc = duet_ef_materialise_seeds(ctx))) return rc;
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
