# coding=utf-8
"""TEST INFRASTRUCTURE ONLY -- plain-Python statement of SVIM-mode signature extraction and of the whole SVIM-mode
pipeline built from it (BAM text -> raw SV marks -> stage A0 -> adapter -> step E/F).

Parity status: UNPINNED against the reference.  The reference delegates signature extraction and clustering to the
external `svim alignment` binary (src/duet/sv_calling.py:13-15; svim 1.4.2, README.md:31,42), whose source is not under
/root/reference.  This file is the normative text of the repository's own deterministic rule, modelled on the
intra-alignment part of SVIM (SVIM_intra.py, analyze_cigar_indel): large insertions and deletions inside one alignment,
and on the insertion / deletion part of its inter-alignment analysis (SVIM_inter.py, analyze_read_segments): a read
whose consecutive segments (primary + supplementary alignments) leave a gap on the read or on the reference.

Rule, per `samtools view` text line of a contig's haplotagged BAM:
  * skipped: unmapped (flag 0x4), secondary (0x100), MAPQ < min_mapq (20);
  * walking the CIGAR from POS: M/=/X/N/D advance the reference; every I or D of at least min_sv_size (40) bases is
    one mark -- type INS=1 / DEL=0, pos = 1-based reference position where it starts, span = its length, read = the
    line's read name;
  * depth[b] = number of kept alignments [POS-1, end) that contain b * bin + bin // 2 (0-based), bin = 1000;
  * split reads: the kept alignments of one read name (in one contig's file) are its segments.  In read orientation a
    segment covers [qs, qe): qs = its leading clip (S or H) -- its TRAILING clip when the alignment is reverse (flag
    0x10) --, qe = qs + the bases it aligns (M, I, =, X).  Segments sorted by (qs, qe, line order); for consecutive
    segments a, b on the same strand: dread = b.qs - a.qe; dref = b.start - a.end (forward) or a.start - b.end
    (reverse); both must be >= -5 (overlap tolerance); dev = dread - dref; the event sits at the reference end of the
    left segment (a.end forward, b.end reverse), pos = that + 1: dev >= min_sv_size -> INS of span dev;
    -100000 <= dev <= -min_sv_size -> DEL of span -dev.
    Round 4 (SVIM_inter.py's tandem-duplication and inversion cases, simplified; type codes DEL 0 / INS 1 / INV 2 / DUP 3):
    same strand, dread >= -5 and dref < -5 -- the read goes BACK on the reference: a tandem duplication of the stretch both
    segments cover, [b.start, a.end) forward / [a.start, b.end) reverse: DUP, pos = its start + 1, span = its length, when
    min_sv_size <= span <= 100000;  opposite strands, dread >= -5: the segments meet at their right ends (a forward, b
    reverse: p1 = a.end, p2 = b.end) or at their left ends (a reverse, b forward: p1 = a.start, p2 = b.start): INV,
    pos = min(p1, p2) + 1, span = |p2 - p1|, when min_sv_size <= span <= 100000.
    These marks follow the contig's CIGAR marks, reads in order of first appearance.
The tag of a mark's read is looked up in the contig's tag table exactly like step E/F does (ef_oracle.tags_from_sam_text).
"""

import os
import re

import numpy as np

from oracle import ef_oracle

_CIG = re.compile(r'(\d+)([MIDNSHP=X])')


SEG_TOL = 5
SPLIT_MAX_DEL = 100000


def extract_from_sam_text(text, min_sv_size=40, min_mapq=20, depth_bin=1000):
    """-> (marks [(type, pos, span, read name)], depth list[int])."""
    marks = []
    diff = {}
    top = 0
    segs = {}                                           # read name -> [(qs, qe, line index, start, end, reverse)]
    for li, line in enumerate(text.split('\n')[:-1]):
        f = line.split('\t')
        flag, pos, mapq, cigar = int(f[1]), int(f[3]), int(f[4]), f[5]
        if flag & 0x104 or mapq < min_mapq or pos < 1 or cigar == '*':
            if cigar == '*' and not (flag & 0x104) and mapq >= min_mapq and pos >= 1:
                pass                                    # no CIGAR: no marks, zero-length span, no depth
            continue
        ref = pos - 1
        ops = [(int(n), op) for n, op in _CIG.findall(cigar)]
        lead = trail = 0
        for n, op in ops:
            if op not in 'SH':
                break
            lead += n
        for n, op in reversed(ops):
            if op not in 'SH':
                break
            trail += n
        aligned = sum(n for n, op in ops if op in 'MI=X')
        ref_len = sum(n for n, op in ops if op in 'MDN=X')
        if aligned:
            rev = bool(flag & 0x10)
            qs = trail if rev else lead
            segs.setdefault(f[0], []).append((qs, qs + aligned, li, pos - 1, pos - 1 + ref_len, rev))
        for n, op in ops:
            if op in 'ID':
                if n >= min_sv_size:
                    marks.append((1 if op == 'I' else 0, ref + 1, n, f[0]))
                if op == 'D':
                    ref += n
            elif op in 'MN=X':
                ref += n
        s, e, w, half = pos - 1, ref, depth_bin, depth_bin // 2
        if e > s:
            lo = 0 if s <= half else (s - half + w - 1) // w
            hi = 0 if e <= half else (e - half + w - 1) // w
            if hi > lo:
                diff[lo] = diff.get(lo, 0) + 1
                diff[hi] = diff.get(hi, 0) - 1
                top = max(top, hi)
    for name, sg in segs.items():                       # dict order = first appearance
        sg.sort()
        for a, b in zip(sg, sg[1:]):
            dread = b[0] - a[1]
            if a[5] != b[5]:
                if dread < -SEG_TOL:
                    continue
                p1, p2 = (a[3], b[3]) if a[5] else (a[4], b[4])
                if min_sv_size <= abs(p2 - p1) <= SPLIT_MAX_DEL:
                    marks.append((2, min(p1, p2) + 1, abs(p2 - p1), name))
                continue
            dref = (a[3] - b[4]) if a[5] else (b[3] - a[4])
            if dread < -SEG_TOL:
                continue
            if dref < -SEG_TOL:
                s0, e0 = (a[3], b[4]) if a[5] else (b[3], a[4])
                if min_sv_size <= e0 - s0 <= SPLIT_MAX_DEL:
                    marks.append((3, s0 + 1, e0 - s0, name))
                continue
            dev = dread - dref
            anchor = b[4] if a[5] else a[4]
            if dev >= min_sv_size:
                marks.append((1, anchor + 1, dev, name))
            elif min_sv_size <= -dev <= SPLIT_MAX_DEL:
                marks.append((0, anchor + 1, -dev, name))
    depth, run = [], 0
    for b in range(top):
        run += diff.get(b, 0)
        depth.append(run)
    return marks, depth


def extract_workdir(home, chroms, **kw):
    """Per contig of `chroms` (labels): marks with resolved tags and depth.
    -> dict(contig u16[M], type u8[M], pos u32[M], span u32[M], tag list[(hap, ps, pc) or None], depth [K] lists,
            tables [K] dicts)"""
    d = os.path.join(home, 'snp_phasing')
    tables = ef_oracle.load_tag_tables(d, chroms)
    out = dict(contig=[], type=[], pos=[], span=[], tag=[], name=[], depth=[], tables=tables)
    for k, c in enumerate(chroms):
        stem = None
        for cand in ('chr' + c + '.bam', c + '.bam'):
            if os.path.exists(os.path.join(d, cand)) or os.path.exists(os.path.join(d, cand + '.sam')):
                stem = os.path.join(d, cand)
                break
        if stem is None:
            out['depth'].append([])
            continue
        with open(stem + '.sam') as f:
            marks, depth = extract_from_sam_text(f.read(), **kw)
        out['depth'].append(depth)
        for (t, p, sp, name) in marks:
            out['contig'].append(k)
            out['type'].append(t)
            out['pos'].append(p)
            out['span'].append(sp)
            out['name'].append(name)
            out['tag'].append(tables[k].get(name))
    for key, dt in (('contig', np.uint16), ('type', np.uint8), ('pos', np.uint32), ('span', np.uint32)):
        out[key] = np.array(out[key], dtype=dt)
    return out


def pack_tag(t):
    """(hap, ps, pc) -> the u64 word of include/duet_ef.h (hap 1/2, anything else 3; pc saturated)."""
    hap, ps, pc = t
    code = hap if hap in (1, 2) else 3
    return (code << 62) | (min(pc, (1 << 30) - 2) << 32) | ps


def phase_workdir(home, chroms, svlen_thres=50, suppread_thres=2, max_dist=0.9, depth_bin=1000, **kw):
    """The whole SVIM-mode pipeline on the CPU: extraction -> cluster oracle -> adapter -> E/F oracle.
    -> dict(cand_contig, cand_type, cand_pos, cand_span, support, pred, ps)"""
    from duet_amd import engine
    from oracle import c_oracle
    ex = extract_workdir(home, chroms, depth_bin=depth_bin, **kw)
    K = len(chroms)
    # one global tag table: contig after contig, names in first-appearance order of the TAGGED lines (dict order)
    read_tag, index = [], []
    for k in range(K):
        ids = {}
        for name, t in ex['tables'][k].items():
            ids[name] = len(read_tag)
            read_tag.append(pack_tag(t))
        index.append(ids)
    mark_read = np.array([index[int(k)].get(n, 0xFFFFFFFF) for k, n in zip(ex['contig'], ex['name'])], dtype=np.uint32)
    cl = c_oracle.cluster(ex['contig'], ex['type'], ex['pos'], ex['span'], max_dist=max_dist)
    N = len(cl['cand_pos'])
    support = np.diff(cl['cand_off'].astype(np.int64))
    kk = cl['cand_contig'].astype(np.int64)
    d = np.zeros(N, dtype=np.int64)
    for i in range(N):
        dep = ex['depth'][int(kk[i])]
        if dep:
            d[i] = dep[min(int(cl['cand_pos'][i]) // depth_bin, len(dep) - 1)]
    soa = engine.EfSoA(cand_ctg_off=np.searchsorted(kk, np.arange(K + 1)), read_tag=np.array(read_tag, dtype=np.uint64),
                       cand_pos=cl['cand_pos'], cand_svlen=cl['cand_span'], cand_svread=support,
                       cand_refread=np.maximum(d - support, 0), cand_gt_ok=np.ones(N, dtype=np.uint8),
                       cand_off=cl['cand_off'], mark_read=mark_read[cl['order']] if len(mark_read) else mark_read)
    rc, pred, ps = c_oracle.ef(soa, svlen_thres, suppread_thres)
    if rc:
        raise ZeroDivisionError('division by zero')
    return dict(cand_contig=cl['cand_contig'], cand_type=cl['cand_type'], cand_pos=cl['cand_pos'], cand_span=cl['cand_span'],
                support=support, pred=pred, ps=ps)
