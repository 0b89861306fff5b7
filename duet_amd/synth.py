# coding=utf-8
"""Deterministic synthetic inputs for Duet's step E/F (SV-mark / read-tag integration).

Everything is derived from a counter-based splitmix64 stream implemented with integer numpy
arithmetic only (no libm, no numpy Generator), so the same seed gives the same bytes on any host.
That matters because the golden sha256 values in tests/golden/ are produced in the development
container (where the reference can be imported) and re-checked on the GPU box from regenerated
inputs.

Two families:

* ``bench_contig`` / ``bench_genome`` -- the workload shapes of SURVEY.md section 8d (config 2:
  1 contig, 1e6 marks, 2e5 reads, 1e5 candidates; config 3: 24 contigs, 2e7 marks).  Pure array
  generation; can emit the SoA directly (``SynthContig.soa_parts``) or text (VCF + SAM/BAM).
* ``fuzz_case`` -- small adversarial cases: threshold-boundary PC values and read-count ratios,
  three caller dialects, duplicate SAM lines (last wins), duplicated RNAMES entries, unsorted
  positions, contigs without BAM, records on unlisted contigs, ``./.`` genotypes, ``SVLEN=.``.

Input formats follow what the reference reads: caller VCF as parsed by read_file.py:25-77 and
`samtools view` text as parsed by sv_phasing_fn.py:25-29 (last three fields HP:i, PC:i, PS:i).
"""

import os
import numpy as np

from duet_amd import bamio

MASK64 = (1 << 64) - 1
_GOLD = np.uint64(0x9E3779B97F4A7C15)

# hg19 lengths for the default contig universe (read_file.py:7-12 lists 1..22, X, Y).
HG19_LENGTHS = {
    '1': 249250621, '2': 243199373, '3': 198022430, '4': 191154276, '5': 180915260,
    '6': 171115067, '7': 159138663, '8': 146364022, '9': 141213431, '10': 135534747,
    '11': 135006516, '12': 133851895, '13': 115169878, '14': 107349540, '15': 102531392,
    '16': 90354753, '17': 81195210, '18': 78077248, '19': 59128983, '20': 63025520,
    '21': 48129895, '22': 51304566, 'X': 155270560, 'Y': 59373566,
}
DEFAULT_CONTIGS = [str(i) for i in range(1, 23)] + ['X', 'Y']


class SplitMix:
    """Counter-based splitmix64: value i of stream `seed` is mix(seed + (i+1)*GOLD)."""

    def __init__(self, seed):
        self.state = np.uint64(seed & MASK64)

    def u64(self, n):
        n = int(n)
        with np.errstate(over='ignore'):
            ctr = self.state + (np.arange(1, n + 1, dtype=np.uint64) * _GOLD)
            self.state = self.state + np.uint64(n) * _GOLD
            z = ctr
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        return z

    def below(self, n, hi):
        """n integers uniform in [0, hi) (modulo method; the bias is irrelevant here)."""
        return (self.u64(n) % np.uint64(hi)).astype(np.int64)

    def between(self, n, lo, hi):
        """n integers uniform in [lo, hi] inclusive."""
        return self.below(n, hi - lo + 1) + lo

    def chance(self, n, num, den):
        """n booleans, true with probability num/den."""
        return self.below(n, den) < num

    def one(self, hi):
        return int(self.below(1, hi)[0])


def _expish(rng, n, scale):
    """Integer-only 'exponential-like' draw with mean ~1.5*scale: geometric(1/2)*scale + U[0,scale)."""
    r = rng.u64(n)
    # number of trailing one-bits of r (geometric with p = 1/2), capped at 20
    k = np.zeros(n, dtype=np.int64)
    alive = np.ones(n, dtype=bool)
    for b in range(20):
        bit = ((r >> np.uint64(b)) & np.uint64(1)).astype(bool)
        alive &= bit
        k += alive
    return k * scale + rng.below(n, scale)


class SynthContig:
    """One contig's reads (SAM line order) and SV candidates (VCF file order)."""

    def __init__(self, label, spelled, length):
        self.label = label              # entry of the chrom list, e.g. '21'
        self.spelled = spelled          # CHROM column / BAM stem, e.g. 'chr21' or '21'
        self.length = int(length)
        self.has_bam = True
        # SAM lines
        self.line_name_id = None        # int64[L]  name id of each line (names may repeat)
        self.line_pos = None            # int64[L]
        self.line_tagged = None         # bool[L]
        self.line_hap = None            # int64[L]
        self.line_pc = None             # int64[L]
        self.line_ps = None             # int64[L]
        self.name_prefix = 'r%s_' % label
        # candidates
        self.cand_pos = None            # int64[C]
        self.cand_svlen = None          # int64[C]  as printed in INFO (signed); -2**62 means 'SVLEN=.'
        self.cand_svtype = None         # list[str]
        self.cand_svread = None         # int64[C]
        self.cand_refread = None        # int64[C]
        self.cand_gt = None             # list[str]
        self.cand_ref_dot = None        # bool[C] or None: print the reference-read field as '.' (parses as 0)
        self.cand_off = None            # int64[C+1]
        self.mark_name_id = None        # int64[M]  name id (>= n_names means a name with no SAM line)

    # -- names -----------------------------------------------------------------------------
    def name_of(self, name_id):
        return '%s%08x' % (self.name_prefix, int(name_id))

    def names_of(self, ids):
        p = self.name_prefix
        return [p + ('%08x' % int(i)) for i in ids]

    # -- tag table semantics (sv_phasing_fn.py:26-29: tagged lines only, later lines win) -----
    def tag_table(self):
        """-> (name_ids sorted unique int64[Rt], hap, pc, ps) of the reads that end up in the dict."""
        if not self.has_bam or self.line_name_id is None or len(self.line_name_id) == 0:
            z = np.zeros(0, dtype=np.int64)
            return z, z, z, z
        idx = np.nonzero(self.line_tagged)[0]
        ids = self.line_name_id[idx]
        # keep the LAST tagged line per name: stable sort by id, take last of each run
        order = np.argsort(ids, kind='stable')
        ids_s = ids[order]
        last = np.ones(len(ids_s), dtype=bool)
        last[:-1] = ids_s[1:] != ids_s[:-1]
        sel = idx[order[last]]
        return self.line_name_id[sel], self.line_hap[sel], self.line_pc[sel], self.line_ps[sel]

    def soa_parts(self):
        """Direct SoA for this contig (no text round trip).

        Returns dict(read_tag u64[Rt], cand_* arrays, cand_off u32-able int64[C+1], mark_read int64[M])
        with mark_read = index into this contig's read_tag or -1 (name not in the tag dict).
        """
        ids, hap, pc, ps = self.tag_table()
        read_tag = pack_tags(hap, pc, ps)
        if len(ids):
            where = np.searchsorted(ids, self.mark_name_id)
            where_c = np.minimum(where, len(ids) - 1)
            hit = ids[where_c] == self.mark_name_id
            mark_read = np.where(hit, where_c, -1).astype(np.int64)
        else:
            mark_read = np.full(len(self.mark_name_id), -1, dtype=np.int64)
        svlen = self.cand_svlen.copy()
        svlen[svlen == -(1 << 62)] = 0
        refread = self.cand_refread
        if self.cand_ref_dot is not None:
            refread = np.where(self.cand_ref_dot, 0, refread)
        return dict(read_tag=read_tag, cand_pos=self.cand_pos, cand_svlen_abs=np.abs(svlen),
                    cand_svread=self.cand_svread, cand_refread=refread,
                    cand_gt_ok=np.array([g != './.' for g in self.cand_gt], dtype=np.uint8),
                    cand_off=self.cand_off, mark_read=mark_read)


def pack_tags(hap, pc, ps):
    from duet_amd.engine import pack_tags as _pack
    return _pack(hap, pc, ps)


# ------------------------------------------------------------------------------------------
# benchmark-shaped generator
# ------------------------------------------------------------------------------------------

def bench_contig(label, n_reads, n_cands, seed, spelled=None, length=None, deg_lo=2, deg_hi=18,
                 ps_block=500000, window=24, literal_8d=False):
    """One contig of the section-8d workload: reads U[1,L), 80 % tagged, hap U{1,2},
    pc 97 % exp-like(mean ~600) / 3 % U[8101,20000], ps = floor(pos/ps_block)*ps_block+17;
    candidates at sorted positions, degree U{deg_lo..deg_hi}, marks drawn with replacement from the
    `window` reads nearest in position, 5 % of marks replaced by names that have no SAM line.
    literal_8d: SURVEY.md section 8d to the letter where the default departs from it -- pc = floor(Exp(mean 600)) (the default:
    geometric(1/2) * 400 + U[0, 400), capped at 8100) and 20 % of the marks' names absent (the default: 5 %)."""
    rng = SplitMix(seed)
    length = HG19_LENGTHS.get(label, 100000000) if length is None else length
    c = SynthContig(label, spelled or ('chr' + label), length)
    R, C = int(n_reads), int(n_cands)
    pos = np.sort(rng.between(R, 1, length - 1))
    c.line_name_id = np.arange(R, dtype=np.int64)
    c.line_pos = pos
    c.line_tagged = rng.chance(R, 4, 5)
    c.line_hap = rng.between(R, 1, 2)
    pc = _expish(rng, R, 400)
    if literal_8d:
        u = (rng.below(R, 1 << 52).astype(np.float64) + 0.5) / float(1 << 52)
        pc = np.floor(-600.0 * np.log(u)).astype(np.int64)
    hi = rng.chance(R, 3, 100)
    pc = np.where(hi, rng.between(R, 8101, 20000), np.minimum(pc, 8100))
    c.line_pc = pc
    c.line_ps = (pos // ps_block) * ps_block + 17
    # candidates
    cpos = np.sort(rng.between(C, 1, length - 1))
    deg = rng.between(C, deg_lo, deg_hi)
    off = np.zeros(C + 1, dtype=np.int64)
    np.cumsum(deg, out=off[1:])
    M = int(off[-1])
    i0 = np.searchsorted(pos, cpos)
    w = min(window, R)
    start = np.clip(i0 - w // 2, 0, R - w)
    mark = np.repeat(start, deg) + rng.below(M, w)
    foreign = rng.chance(M, 1, 5 if literal_8d else 20)
    mark = np.where(foreign, R + rng.below(M, 1 << 20), mark)
    c.cand_pos = cpos
    tsel = rng.below(C, 100)
    types = np.where(tsel < 49, 0, np.where(tsel < 98, 1, np.where(tsel < 99, 2, 3)))
    tnames = ['INS', 'DEL', 'DUP', 'INV']
    c.cand_svtype = [tnames[t] for t in types]
    mag = rng.between(C, 30, 5000)
    c.cand_svlen = np.where(types == 1, -mag, mag)
    c.cand_svread = deg.copy()
    c.cand_refread = rng.between(C, 0, 20)
    gsel = rng.below(C, 4)
    gts = ['0/1', '1/1', './.', '0/0']
    c.cand_gt = [gts[g] for g in gsel]
    c.cand_off = off
    c.mark_name_id = mark
    return c


def bench_genome(n_marks_target, seed, labels=None, reads_per_mark=0.2, mean_deg=10, deg_lo=2, deg_hi=18):
    """Config-3 style set: contigs with hg19 lengths, work split in proportion to length.  (deg_lo == deg_hi: every candidate
    with the same number of marks -- tools/prof_ef.py's upper bound for a load-balanced walk.)"""
    labels = DEFAULT_CONTIGS if labels is None else labels
    tot = float(sum(HG19_LENGTHS[l] for l in labels))
    out = []
    for i, l in enumerate(labels):
        frac = HG19_LENGTHS[l] / tot
        C = max(2, int(round(n_marks_target * frac / mean_deg)))
        R = max(32, int(round(n_marks_target * frac * reads_per_mark)))
        out.append(bench_contig(l, R, C, seed * 1000003 + i, deg_lo=deg_lo, deg_hi=deg_hi))
    return out


MARK_TYPES = {'DEL': 0, 'INS': 1, 'INV': 2, 'DUP': 3}


def raw_marks(contigs, seed, pos_jitter=40, span_jitter_pct=6, shuffle=True, reads_of=None, scan_order=False):
    """Raw SV marks (signatures) for the A0 clustering stage: every support-read mark of every candidate
    becomes one (contig, type, pos, span) record near its candidate -- pos +- pos_jitter, span within
    +- span_jitter_pct % of the candidate's |SVLEN| -- in shuffled order (the worst case for everything that gathers by
    mark index), or with scan_order in the order a scan of coordinate-sorted BAMs emits them: contig by contig, read by read,
    a read starting up to 10 kb in front of its mark.
    -> dict(contig u16[M], type u8[M], pos u32[M], span u32[M], truth int64[M] (global candidate index)
            [, read u32[M] = the mark's index into reads_of.read_tag or 0xFFFFFFFF, when reads_of (an EfSoA built
            from the same contigs) is given])"""
    rng = SplitMix(0xC1050000 + seed)
    parts = {k: [] for k in ('contig', 'type', 'pos', 'span', 'truth')}
    base = 0
    for ci, c in enumerate(contigs):
        deg = np.diff(c.cand_off)
        M = int(c.cand_off[-1])
        cand = np.repeat(np.arange(len(deg)), deg)
        mag = np.abs(np.where(c.cand_svlen == -(1 << 62), 60, c.cand_svlen))
        tcode = np.array([MARK_TYPES.get(t.split(':')[0], 2) for t in c.cand_svtype], dtype=np.int64)
        jit = rng.between(M, -pos_jitter, pos_jitter)
        pos = np.maximum(c.cand_pos[cand] + jit, 1)
        sp = mag[cand]
        sj = (sp * rng.between(M, -span_jitter_pct, span_jitter_pct)) // 100
        span = np.maximum(sp + sj, 1)
        parts['contig'].append(np.full(M, ci, dtype=np.int64))
        parts['type'].append(tcode[cand])
        parts['pos'].append(pos)
        parts['span'].append(span)
        parts['truth'].append(cand + base)
        base += len(deg)
    out = {k: np.concatenate(v) if v else np.zeros(0, dtype=np.int64) for k, v in parts.items()}
    if reads_of is not None:
        out['read'] = reads_of.mark_read.astype(np.int64)          # same (candidate-major) order as the marks above
    if scan_order and len(out['pos']):
        start = np.maximum(out['pos'] - rng.below(len(out['pos']), 10000), 0)
        perm = np.argsort(out['contig'] * (1 << 40) + start, kind='stable')
        out = {k: v[perm] for k, v in out.items()}
    elif shuffle and len(out['pos']):
        perm = np.argsort(rng.u64(len(out['pos'])), kind='stable')
        out = {k: v[perm] for k, v in out.items()}
    res = dict(contig=out['contig'].astype(np.uint16), type=out['type'].astype(np.uint8),
               pos=out['pos'].astype(np.uint32), span=out['span'].astype(np.uint32), truth=out['truth'])
    if 'read' in out:
        res['read'] = out['read'].astype(np.uint32)
    return res


def depth_bins(contigs, bin_width=1000, seed=1):
    """Synthetic binned coverage for the fused SVIM-mode pipeline: -> (depth u32[total], depth_off int64[K+1])."""
    rng = SplitMix(0xDE970000 + seed)
    off = [0]
    parts = []
    for c in contigs:
        nb = c.length // bin_width + 1
        parts.append(8 + rng.below(nb, 30))
        off.append(off[-1] + nb)
    return np.concatenate(parts).astype(np.uint32), np.array(off, dtype=np.int64)


# ------------------------------------------------------------------------------------------
# adversarial small cases
# ------------------------------------------------------------------------------------------

_PC_EDGE = [0, 1, 100, 300, 972, 973, 1000, 1369, 1370, 2000, 2400, 2401, 2500, 8100, 8101, 9000, 9720]
_RATIO_EDGE = [(6, 19), (3, 3), (10, 1), (3, 7), (8, 11), (4, 6), (5, 5), (9, 1), (9, 11), (3, 1),
               (18, 7), (5, 2), (2, 0), (3, 0), (4, 0), (13, 0), (12, 0), (20, 20), (4, 4), (1, 9)]
_TYPES = ['INS', 'DEL', 'DUP', 'INV', 'DUP:TANDEM', 'BND', 'DUP:INT']
_GTS = ['0/1', '1/1', './.', '0/0', '1/0', '.']


def fuzz_case(seed, labels=None, n_contigs=3, bare_bias=False):
    """Small adversarial multi-contig case. Returns list[SynthContig] (some without BAM, some
    spelled without the 'chr' prefix, plus one contig outside the default list).  bare_bias: spell three
    contigs in four exactly like their list entry (the natural case with -a, where the list already holds the
    names as the files spell them) instead of 'chr' + entry."""
    rng = SplitMix(0xF00D0000 + seed)
    pool = DEFAULT_CONTIGS if labels is None else labels
    picks = []
    while len(picks) < min(n_contigs, len(pool)):
        l = pool[rng.one(len(pool))]
        if l not in picks:
            picks.append(l)
    out = []
    for ci, l in enumerate(picks):
        spelled = ('chr' + l) if bool(rng.one(4)) != bool(bare_bias) else l
        length = 2000000 + rng.one(3000000)
        c = SynthContig(l, spelled, length)
        R = 20 + rng.one(120)
        C = 10 + rng.one(90)
        n_ps = 1 + rng.one(5)
        ps_vals = np.sort(rng.between(n_ps, 1, length - 1))
        # lines: R base reads + some duplicated names (supplementary alignments; later lines win)
        ndup = rng.one(1 + R // 4)
        ids = np.concatenate([np.arange(R, dtype=np.int64), rng.below(ndup, R)])
        L = len(ids)
        c.line_name_id = ids
        c.line_pos = rng.between(L, 1, length - 1)
        c.line_tagged = rng.chance(L, 3, 4)
        c.line_hap = rng.between(L, 1, 2)
        edge = rng.chance(L, 1, 2)
        pce = np.array(_PC_EDGE, dtype=np.int64)[rng.below(L, len(_PC_EDGE))]
        c.line_pc = np.where(edge, pce, rng.between(L, 0, 10000))
        # reads cluster into phase sets by position third, so most candidates see one PS
        c.line_ps = ps_vals[(ids * n_ps // max(R, 1)) % n_ps]
        flip = rng.chance(L, 1, 10)
        c.line_ps = np.where(flip, ps_vals[rng.below(L, n_ps)], c.line_ps)
        c.has_bam = rng.one(8) != 0
        # candidates (file order NOT sorted by position for a third of the cases)
        cpos = rng.between(C, 1, length - 1)
        if rng.one(3):
            cpos = np.sort(cpos)
        deg = rng.between(C, 1, 14)
        off = np.zeros(C + 1, dtype=np.int64)
        np.cumsum(deg, out=off[1:])
        M = int(off[-1])
        centre = rng.below(C, R)
        spread = 1 + rng.one(10)
        mark = (np.repeat(centre, deg) + rng.below(M, spread)) % R
        foreign = rng.chance(M, 1, 8)
        mark = np.where(foreign, R + rng.below(M, 50), mark)
        c.cand_pos = cpos
        c.cand_svtype = [_TYPES[t] if t < len(_TYPES) else ('INS' if t % 2 else 'DEL')
                         for t in rng.below(C, len(_TYPES) + 8)]
        mag = rng.between(C, 1, 400)
        small = rng.chance(C, 1, 10)
        mag = np.where(small, rng.between(C, 45, 55), mag + 49)
        svlen = np.where(np.array([t == 'DEL' for t in c.cand_svtype]), -mag, mag)
        svlen = np.where(rng.chance(C, 1, 40), -(1 << 62), svlen)      # 'SVLEN=.'
        c.cand_svlen = svlen
        redge = rng.chance(C, 1, 2)
        pairs = np.array(_RATIO_EDGE, dtype=np.int64)[rng.below(C, len(_RATIO_EDGE))]
        c.cand_svread = np.where(redge, pairs[:, 0], rng.between(C, 1, 30))
        c.cand_refread = np.where(redge, pairs[:, 1], rng.between(C, 0, 30))
        # keep svread + refread > 0 (the reference divides by it, sv_phasing_fn.py:123)
        gsel = rng.below(C, len(_GTS) + 6)
        c.cand_gt = [_GTS[g] if g < len(_GTS) else '0/1' for g in gsel]
        c.cand_ref_dot = rng.chance(C, 1, 25)
        c.cand_off = off
        c.mark_name_id = mark
        out.append(c)
    return out


# contig names for -a / --include_all_ctgs cases: what `tabix --list-chroms` prints for a pileup VCF (read_file.py:13-15).
# No name equals 'chr' + another one (such a list names one contig twice; the build refuses it).
ALL_CTG_POOL = ['chr1', 'chr2', 'chr10', 'chr19', 'chrX', 'chrM', 'chrUn_gl000220', 'chr6_cox_hap2', 'GL000192.1',
                'KI270728.1', 'MT', '21', 'HLA-A*01:01', 'chrEBV', 'scaffold_7', 'Y']


def fuzz_case_all_ctgs(seed, n_contigs=4):
    """A fuzz case for -a mode.  -> (contigs, listing, header_contigs): `listing` is the contig universe in the order
    `tabix --list-chroms` would print it (a shuffled subset of ALL_CTG_POOL, NOT karyotype order); the candidates sit on
    `n_contigs` of them; `header_contigs` are the ##contig lines of the caller VCF in a third order and include names
    that are not in the listing (with -a every ##contig line is copied in FILE order, write_file.py:38-41)."""
    rng = SplitMix(0xA11C0000 + seed)
    pool = list(ALL_CTG_POOL)
    order = np.argsort(rng.u64(len(pool)), kind='stable')
    listing = [pool[i] for i in order[:6 + rng.one(len(pool) - 6)]]
    contigs = fuzz_case(seed, labels=listing, n_contigs=min(n_contigs, len(listing)), bare_bias=True)
    spelled = {c.label: c.spelled for c in contigs}
    names = [spelled.get(l, l) for l in listing] + ['chrNotListed_1', 'decoy']
    horder = np.argsort(rng.u64(len(names)), kind='stable')
    header_contigs = [(names[i], 1000 + 37 * int(i)) for i in horder]
    return contigs, listing, header_contigs


# ------------------------------------------------------------------------------------------
# text writers (what the reference consumes)
# ------------------------------------------------------------------------------------------

DIALECTS = ('cutesv', 'sniffles', 'svim')


def _info_and_sample(dialect, svtype, svlen, pos, svread, refread, gt, names, rng_bits, ref_dot=False):
    svlen_s = '.' if svlen == -(1 << 62) else str(int(svlen))
    end = pos + (abs(int(svlen)) if svlen != -(1 << 62) and svtype != 'INS' else 0)
    rn = ','.join(names)
    if dialect == 'cutesv':
        info = 'PRECISE;SVTYPE=%s;SVLEN=%s;END=%d;CIPOS=-%d,%d;CILEN=-1,1;RE=%d;RNAMES=%s;STRAND=+-' % (
            svtype, svlen_s, end, rng_bits % 7, rng_bits % 5, svread, rn)
        fmt = 'GT:DR:DV:PL:GQ'
        dr = '.' if ref_dot else str(int(refread))
        sample = '%s:%s:%d:%d,%d,%d:%d' % (gt, dr, svread, rng_bits % 97, rng_bits % 13, rng_bits % 89,
                                          rng_bits % 61)
    elif dialect == 'sniffles':
        info = ('PRECISE;SVTYPE=%s;SVLEN=%s;END=%d;SUPPORT=%d;RNAMES=%s;COVERAGE=%d,%d,%d,%d,%d;'
                'STRAND=+-;AF=0.500;STDEV_LEN=%d.000;STDEV_POS=%d.000;SUPPORT_LONG=0' % (
                    svtype, svlen_s, end, svread, rn, rng_bits % 30, rng_bits % 31, rng_bits % 29,
                    rng_bits % 28, rng_bits % 27, rng_bits % 9, rng_bits % 11))
        fmt = 'GT:GQ:DR:DV'
        # NB the reference takes subfield 1 (= GQ) as the "ref read" count for this layout
        # (read_file.py:63-69), so the generator's refread goes there.
        sample = '%s:%s:%d:%d' % (gt, '.' if ref_dot else str(int(refread)), rng_bits % 40, svread)
    elif dialect == 'svim':
        info = 'SVTYPE=%s;END=%d;SVLEN=%s;SUPPORT=%d;STD_SPAN=%d.5;STD_POS=%d.25;READS=%s' % (
            svtype, end, svlen_s, svread, rng_bits % 17, rng_bits % 19, rn)
        fmt = 'GT:DP:AD'
        if gt == './.':
            sample = './.:.:.,.'
        else:
            sample = '%s:%d:%s,%d' % (gt, refread + svread, '.' if ref_dot else str(int(refread)), svread)
    else:
        raise ValueError('unknown dialect ' + dialect)
    return info, fmt, sample


def write_vcf(path, contigs, dialect='cutesv', header_contigs=None, extra_contig_records=True, seed=1):
    """Caller-style VCF. `header_contigs`: list of (name, length) for ##contig lines (default: the
    24 hg19 contigs spelled like the data, in karyotype order, plus chrM)."""
    rng = SplitMix(0xABCD0000 + seed)
    lines = ['##fileformat=VCFv4.2', '##source=synthetic-%s' % dialect,
             '##ALT=<ID=INS,Description="Insertion">', '##ALT=<ID=DEL,Description="Deletion">']
    spelled = {c.label: c.spelled for c in contigs}
    if header_contigs is None:
        header_contigs = []
        for l in DEFAULT_CONTIGS:
            header_contigs.append((spelled.get(l, 'chr' + l), HG19_LENGTHS[l]))
        header_contigs.append(('chrM', 16571))
    for name, ln in header_contigs:
        lines.append('##contig=<ID=%s,length=%d>' % (name, ln))
    lines.append('##INFO=<ID=SVTYPE,Number=1,Type=String,Description="Type of structural variant">')
    lines.append('##INFO=<ID=SVLEN,Number=1,Type=Integer,Description="Length of the SV">')
    lines.append('##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">')
    lines.append('#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE')
    n = 0
    for c in contigs:
        C = len(c.cand_pos)
        bits = rng.u64(C) >> np.uint64(40)
        for j in range(C):
            names = c.names_of(c.mark_name_id[c.cand_off[j]:c.cand_off[j + 1]])
            svtype = c.cand_svtype[j]
            info, fmt, sample = _info_and_sample(dialect, svtype, int(c.cand_svlen[j]), int(c.cand_pos[j]),
                                                 int(c.cand_svread[j]), int(c.cand_refread[j]),
                                                 c.cand_gt[j], names, int(bits[j]),
                                                 bool(c.cand_ref_dot[j]) if c.cand_ref_dot is not None else False)
            alt = '<%s>' % svtype
            ref = 'N'
            if dialect == 'cutesv' and svtype in ('INS', 'DEL') and (int(bits[j]) & 3) == 0:
                ref, alt = ('A', 'ACGTTGCA') if svtype == 'INS' else ('ACGTTGCA', 'A')
            lines.append('\t'.join([c.spelled, str(int(c.cand_pos[j])), '%s.%s.%d' % (dialect, svtype, n), ref,
                                    alt, '.' if dialect != 'svim' else str(int(bits[j]) % 60), 'PASS', info,
                                    fmt, sample]))
            n += 1
    if extra_contig_records:
        # records on a contig outside the default list vanish at parse (read_file.py:30)
        lines.append('\t'.join(['chrM', '100', 'x.0', 'N', '<DEL>', '.', 'PASS',
                                'PRECISE;SVTYPE=DEL;SVLEN=-80;END=180;RE=5;RNAMES=zz1,zz2;STRAND=+-',
                                'GT:DR:DV:PL:GQ', '0/1:3:5:1,2,3:9']))
    with open(path, 'w') as f:
        f.write('\n'.join(lines) + '\n')


def sam_lines(c):
    """`samtools view` style text for contig c (tagged reads end with HP:i, PC:i, PS:i)."""
    out = []
    names = c.names_of(c.line_name_id)
    for i in range(len(names)):
        core = '%s\t%d\t%s\t%d\t60\t*\t*\t0\t0\t*\t*\tNM:i:%d' % (
            names[i], 0 if i < len(names) // 2 else 2048, c.spelled, int(c.line_pos[i]), i % 50)
        if c.line_tagged[i]:
            core += '\tHP:i:%d\tPC:i:%d\tPS:i:%d' % (int(c.line_hap[i]), int(c.line_pc[i]), int(c.line_ps[i]))
        out.append(core)
    return out


def svim_sam_lines(c, seed, min_sv=40, pos_jitter=15, len_jitter_pct=4, split_every=7):
    """`samtools view` style text for contig c in which the SV evidence sits INSIDE the alignments (SVIM mode): the
    first line of every read that supports candidates carries one CIGAR insertion / deletion per supported INS / DEL
    candidate (jittered position and length), the other lines are plain matches.  Tags as in sam_lines().
    Every `split_every`-th INS / DEL candidate also gets one extra SPLIT read (two lines: primary + supplementary
    alignment, forward and reverse strand alternating) whose segments leave the event between them."""
    rng = SplitMix(0x51A70000 + seed)
    names = c.names_of(c.line_name_id)
    L = len(names)
    events = {}
    C = len(c.cand_pos)
    for j in range(C):
        t = c.cand_svtype[j]
        ln = abs(int(c.cand_svlen[j])) if int(c.cand_svlen[j]) > -(1 << 61) else 0
        if t not in ('INS', 'DEL') or ln < min_sv:
            continue
        for nid in c.mark_name_id[c.cand_off[j]:c.cand_off[j + 1]]:
            events.setdefault(int(nid), []).append((int(c.cand_pos[j]), t, ln))
    jit = rng.between(L * 4 + 8, 0, 2 * pos_jitter)
    ljit = rng.between(L * 4 + 8, -len_jitter_pct, len_jitter_pct)
    mapq = rng.between(L, 0, 60)
    seen = set()
    out = []
    u = 0
    for i in range(L):
        nid = int(c.line_name_id[i])
        ev = events.get(nid) if nid not in seen else None
        seen.add(nid)
        flag = 0 if i < L // 2 else 2048
        start = max(1, int(c.line_pos[i]))
        cigar = '%dM' % (200 + i % 300)
        if ev:
            ev = sorted(ev)
            start = max(1, ev[0][0] - 400 - (i % 200))
            ref = start
            ops = []
            for (p0, t, ln) in ev:
                p1 = p0 + int(jit[u % len(jit)]) - pos_jitter
                ln1 = max(1, ln * (100 + int(ljit[u % len(ljit)])) // 100)
                u += 1
                if p1 <= ref:                      # would overlap the previous event: this read does not show it
                    continue
                ops.append('%dM' % (p1 - ref))
                ref = p1
                if t == 'INS':
                    ops.append('%dI' % ln1)
                else:
                    ops.append('%dD' % ln1)
                    ref += ln1
            ops.append('%dM' % (300 + i % 100))
            cigar = ''.join(ops)
        core = '%s\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t*\t*\tNM:i:%d' % (names[i], flag, c.spelled, start, int(mapq[i]), cigar, i % 50)
        if c.line_tagged[i]:
            core += '\tHP:i:%d\tPC:i:%d\tPS:i:%d' % (int(c.line_hap[i]), int(c.line_pc[i]), int(c.line_ps[i]))
        out.append(core)
    if split_every:
        k = 0
        for j in range(C):
            t = c.cand_svtype[j]
            ln = abs(int(c.cand_svlen[j])) if int(c.cand_svlen[j]) > -(1 << 61) else 0
            p0 = int(c.cand_pos[j])
            if t not in ('INS', 'DEL') or ln < min_sv or p0 < 1000:
                continue
            k += 1
            if k % split_every:
                continue
            name = c.name_of(0x40000000 + j)
            rev = (k // split_every) % 2 == 1
            left_len, right_len = 500 + j % 40, 400 + j % 30
            gap_read = ln if t == 'INS' else 0              # bases of the read between the two segments
            right_pos = p0 + (ln if t == 'DEL' else 0)       # 0-based start of the right segment = p0 (+ deletion)
            total = left_len + gap_read + right_len
            if not rev:
                left = '%dM%dS' % (left_len, total - left_len)
                right = '%dH%dM' % (left_len + gap_read, right_len)
            else:                                          # reverse strand: SAM shows the reverse complement
                left = '%dM%dS' % (left_len, total - left_len)
                right = '%dS%dM' % (left_len + gap_read, right_len)
            tag = '\tHP:i:%d\tPC:i:%d\tPS:i:%d' % (1 + j % 2, 50 + j % 900, (p0 // 500000) * 500000 + 17) if j % 3 else ''
            fl = 16 if rev else 0
            out.append('%s\t%d\t%s\t%d\t60\t%s\t*\t0\t0\t*\t*\tNM:i:1%s' % (name, fl, c.spelled, p0 - left_len + 1, left, tag))
            out.append('%s\t%d\t%s\t%d\t60\t%s\t*\t0\t0\t*\t*\tNM:i:2%s' % (name, fl | 2048, c.spelled, right_pos + 1, right, tag))
        # round 4: tandem duplications and inversions only show in split reads -- candidates of those types (and every 11th
        # INS / DEL position, as a second event there) get `dup_reads` split reads each: a duplication's read runs to the end
        # of the duplicated stretch and starts over at its beginning; an inversion's read turns round at the breakpoint
        for j in range(C):
            t = c.cand_svtype[j].split(':')[0]
            ln = abs(int(c.cand_svlen[j])) if int(c.cand_svlen[j]) > -(1 << 61) else 0
            p0 = int(c.cand_pos[j])
            kind = t if t in ('DUP', 'INV') else (('DUP', 'INV')[j % 2] if j % 11 == 5 else None)
            if kind is None or ln < min_sv or ln > 90000 or p0 < 2000 or p0 + ln + 1000 >= c.length:
                continue
            for r in range(3 + j % 3):
                name = c.name_of(0x50000000 + 8 * j + r)
                jp, jl = (j * 7 + r * 3) % 21 - 10, (j * 5 + r) % 9 - 4
                s0, e0 = p0 + jp, p0 + jp + ln + jl             # 0-based [s0, e0): the duplicated / inverted stretch
                la, lb = 450 + (j + r) % 50, 380 + (j * 3 + r) % 60
                rev = (j + r) % 2 == 1
                tag = '\tHP:i:%d\tPC:i:%d\tPS:i:%d' % (1 + (j + r) % 2, 40 + (j * 13 + r) % 900, (p0 // 500000) * 500000 + 17) if (j + r) % 4 else ''
                if kind == 'DUP':
                    # read order: ... up to e0, then again from s0 ...   (reverse strand: the same two alignments, clips swapped)
                    a_pos, b_pos = e0 - la, s0
                    if not rev:
                        a_cig, b_cig = '%dM%dS' % (la, lb), '%dH%dM' % (la, lb)
                    else:
                        a_cig, b_cig = '%dS%dM' % (lb, la), '%dM%dH' % (lb, la)
                    fa = 16 if rev else 0
                    out.append('%s\t%d\t%s\t%d\t60\t%s\t*\t0\t0\t*\t*\tNM:i:3%s' % (name, fa, c.spelled, a_pos + 1, a_cig, tag))
                    out.append('%s\t%d\t%s\t%d\t60\t%s\t*\t0\t0\t*\t*\tNM:i:4%s' % (name, fa | 2048, c.spelled, b_pos + 1, b_cig, tag))
                else:
                    # forward up to s0's ... the read continues on the other strand from e0 backwards: the two right ends meet
                    a_pos, b_pos = s0 - la, e0 - lb
                    if not rev:
                        out.append('%s\t0\t%s\t%d\t60\t%dM%dS\t*\t0\t0\t*\t*\tNM:i:5%s' % (name, c.spelled, a_pos + 1, la, lb, tag))
                        out.append('%s\t2064\t%s\t%d\t60\t%dM%dH\t*\t0\t0\t*\t*\tNM:i:6%s' % (name, c.spelled, b_pos + 1, lb, la, tag))
                    else:
                        # the two left ends meet: first segment reverse at [e0, e0 + la), second forward at [s0, s0 + lb)
                        out.append('%s\t16\t%s\t%d\t60\t%dS%dM\t*\t0\t0\t*\t*\tNM:i:5%s' % (name, c.spelled, e0 + 1, lb, la, tag))
                        out.append('%s\t2048\t%s\t%d\t60\t%dH%dM\t*\t0\t0\t*\t*\tNM:i:6%s' % (name, c.spelled, s0 + 1, la, lb, tag))
    return out


def write_svim_workdir(home, contigs, seed=1, write_sam=True):
    """<home>/snp_phasing/<spelled>.bam (+ .bam.sam) with CIGAR-borne SV evidence: the input of SVIM mode."""
    os.makedirs(os.path.join(home, 'snp_phasing'), exist_ok=True)
    for c in contigs:
        if not c.has_bam:
            continue
        stem = os.path.join(home, 'snp_phasing', c.spelled + '.bam')
        lines = svim_sam_lines(c, seed)
        if write_sam:
            with open(stem + '.sam', 'w') as f:
                f.write(''.join(l + '\n' for l in lines))
        bamio.write_bam_from_sam_lines(stem, [(c.spelled, c.length)], lines)
    return home


def write_workdir(home, contigs, dialect='cutesv', seed=1, write_bam=True, write_sam=True, listing=None, **vcf_kw):
    """Lay out Duet's <OUTPUT> directory for step E/F (sv_phasing.py:12-14):
    <home>/sv_calling/variants.vcf and <home>/snp_phasing/<spelled>.bam (+ .bam.sam text that a
    `samtools view` shim can print).  `listing` (for -a mode): the contig names a
    `tabix --list-chroms` shim prints, written to <home>/snp_calling/pileup.vcf.gz.chroms beside an empty
    pileup.vcf.gz (read_file.py:13-15 only passes the path on)."""
    os.makedirs(os.path.join(home, 'sv_calling'), exist_ok=True)
    os.makedirs(os.path.join(home, 'snp_phasing'), exist_ok=True)
    if listing is not None:
        os.makedirs(os.path.join(home, 'snp_calling'), exist_ok=True)
        open(os.path.join(home, 'snp_calling', 'pileup.vcf.gz'), 'wb').close()
        with open(os.path.join(home, 'snp_calling', 'pileup.vcf.gz.chroms'), 'w') as f:
            f.write(''.join(n + '\n' for n in listing))
    write_vcf(os.path.join(home, 'sv_calling', 'variants.vcf'), contigs, dialect=dialect, seed=seed, **vcf_kw)
    for c in contigs:
        if not c.has_bam:
            continue
        stem = os.path.join(home, 'snp_phasing', c.spelled + '.bam')
        lines = sam_lines(c)
        if write_sam:
            with open(stem + '.sam', 'w') as f:
                f.write(''.join(l + '\n' for l in lines))
        if write_bam:
            bamio.write_bam_from_sam_lines(stem, [(c.spelled, c.length)], lines)
        elif not os.path.exists(stem):
            open(stem, 'wb').close()
    return home
