# coding=utf-8
"""Step E/F engine behind upstream's seam (mirror of src/duet/sv_phasing_fn.py).

    generate_phased_callset(vcf_path, sam_home, svlen_thres, suppread_thres, thread, include_all_ctgs)
        -> list of {ps, hp, chrom, pos, svlen, svtype, ref, alt}, sorted by (chrom text, pos)

has upstream's signature and return value (sv_phasing_fn.py:185-230); its body flattens the inputs
to the arrays of include/duet_ef.h, runs the HIP kernels (filter, PS-class, seed sets, vote,
T1-T5 decision -- :189-212) and assembles the rows.  There is no CPU implementation in this package.
"""

import logging
import os
import shlex
import shutil
import subprocess

import numpy as np

from duet_amd import bamio, engine
from duet_amd.read_file import init_chrom_list, parse_vcf

HP_TEXT = ('', '1|0', '0|1', '1|1')          # pred 1,2,3 -> HP field (sv_phasing_fn.py:217-222)


class TagTables(object):
    """Per contig: read name -> index into that contig's tag arrays (only reads that carry a PC tag,
    later alignments of a name overwrite earlier ones -- sv_phasing_fn.py:28-29)."""

    def __init__(self, n_contigs):
        self.index = [dict() for _ in range(n_contigs)]
        self.hap = [[] for _ in range(n_contigs)]
        self.pc = [[] for _ in range(n_contigs)]
        self.ps = [[] for _ in range(n_contigs)]

    def put(self, k, name, hap, pc, ps):
        idx = self.index[k].get(name)
        if idx is None:
            self.index[k][name] = len(self.hap[k])
            self.hap[k].append(hap)
            self.pc[k].append(pc)
            self.ps[k].append(ps)
        else:
            self.hap[k][idx], self.pc[k][idx], self.ps[k][idx] = hap, pc, ps

    def packed(self):
        """-> (read_tag u64[R_total], read_off int64[K+1])"""
        off = np.zeros(len(self.index) + 1, dtype=np.int64)
        parts = []
        for k in range(len(self.index)):
            off[k + 1] = off[k] + len(self.hap[k])
            parts.append(engine.pack_tags(self.hap[k], self.pc[k], self.ps[k]))
        tags = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint64)
        return tags.astype(np.uint64), off


def _tail_tokens_of_bam(path, thread):
    """(name, last three whitespace tokens) per alignment line, as `samtools view` text would give.
    Uses samtools when DUET_USE_SAMTOOLS=1 and it is on PATH (upstream's way, :25), else the
    built-in BGZF/BAM reader."""
    if os.environ.get('DUET_USE_SAMTOOLS') == '1' and shutil.which('samtools'):
        text = subprocess.check_output(shlex.split('samtools view -@' + str(thread) + ' ' + path)).decode('ascii')
        for line in text.split('\n')[:-1]:
            tok = line.split()
            yield tok[0], tok[-3:]
        return
    for name, mandatory, aux in bamio.iter_bam_records(path):
        yield name, bamio.tail_tokens(mandatory, aux)


def read_hap_bam(path, thread, include_all_ctgs):
    """Tag table per contig from <path>chr<c>.bam, else <path><c>.bam, else empty (:19-24)."""
    logging.info('extract SNP signatures')
    chrom_list = init_chrom_list(include_all_ctgs, path[:len(path) - 13])           # <home>/snp_phasing/
    tables = TagTables(len(chrom_list))
    for k, c in enumerate(chrom_list):
        bam = None
        for cand in (path + 'chr' + c + '.bam', path + c + '.bam'):
            if os.path.exists(cand):
                bam = cand
                break
        if bam is None:
            continue
        seen = False
        for name, tail in _tail_tokens_of_bam(bam, thread):
            seen = True
            if 'PC:i:' in tail[-2]:
                tables.put(k, name, int(tail[-3][5:]), int(tail[-2][5:]), int(tail[-1][5:]))
        logging.info(('  signatures extracted from ' if seen else '  no signature from ') + c)
    return tables


def generate_callinfo(caller_path, read_hap, include_all_ctgs):
    """Join (sv_phasing_fn.py:46-48) + flatten: -> (CallTable, EfSoA)."""
    logging.info('extract SV signatures')
    tab = parse_vcf(caller_path, include_all_ctgs)
    read_tag, read_off = read_hap.packed()
    absent = engine.MARK_ABSENT
    counts = np.fromiter((len(n) for n in tab.names), dtype=np.int64, count=len(tab))
    cand_off = np.zeros(len(tab) + 1, dtype=np.int64)
    np.cumsum(counts, out=cand_off[1:])
    mark = np.empty(int(cand_off[-1]), dtype=np.int64)
    at = 0
    for k, c in enumerate(tab.chrom_list):
        a, b = int(tab.ctg_off[k]), int(tab.ctg_off[k + 1])
        logging.info(('  signatures extracted from ' if b > a else '  no signature from ') + c)
        lookup = read_hap.index[k].get
        base = int(read_off[k])
        for i in range(a, b):
            for name in tab.names[i]:
                j = lookup(name)
                mark[at] = absent if j is None else base + j
                at += 1
    for col, what in ((tab.pos, 'POS'), (tab.svlen_abs, 'SVLEN'), (tab.svread, 'support count'),
                      (tab.refread, 'reference-read count')):
        if col.size and (int(col.min()) < 0 or int(col.max()) > 0xFFFFFFFF):
            raise ValueError(what + ' outside the 32-bit range of the device arrays')
    soa = engine.EfSoA(cand_ctg_off=tab.ctg_off, read_off=read_off, read_tag=read_tag, cand_pos=tab.pos,
                       cand_svlen=tab.svlen_abs, cand_svread=tab.svread, cand_refread=tab.refread,
                       cand_gt_ok=np.fromiter((g != './.' for g in tab.gt), dtype=np.uint8, count=len(tab)),
                       cand_off=cand_off, mark_read=mark)
    return tab, soa


def assemble_rows(tab, pred, ps, classes=None):
    """Rows for every candidate with pred != 0, in upstream's emission order -- contig, then PS-class
    0, 1, 2, then file order (sv_phasing_fn.py:206-228) -- then the stable (chrom text, pos) sort (:229).

    The emission order only matters for ties of the sort key.  Within one contig a tie needs equal
    CHROM text and equal POS; the class-major order is restored from `classes` when given, else ties
    keep file order."""
    idx = np.nonzero(pred)[0]
    if classes is not None and idx.size:
        ctg = np.searchsorted(tab.ctg_off, idx, side='right') - 1
        idx = idx[np.lexsort((idx, classes[idx], ctg))]
    rows = []
    for i in idx:
        i = int(i)
        svtype = tab.svtype[i]
        mag = int(tab.svlen_abs[i])
        rows.append({'ps': int(ps[i]), 'hp': HP_TEXT[int(pred[i])], 'chrom': tab.chrom[i], 'pos': int(tab.pos[i]),
                     'svlen': mag if svtype in ('INS', 'DUP') else -mag, 'svtype': svtype,
                     'ref': tab.ref[i], 'alt': tab.alt[i]})
    rows.sort(key=lambda r: (r['chrom'], r['pos']))
    return rows


def ps_classes(soa):
    """PS-class (0/1/2 = no/one/several distinct PS among tagged marks, :191-194) per candidate --
    host-side, vectorised; only used to order ties of the final sort exactly like upstream."""
    C = soa.n_cands
    cls = np.zeros(C, dtype=np.int64)
    if soa.n_marks == 0:
        return cls
    tagged = soa.mark_read != engine.MARK_ABSENT
    psv = np.zeros(soa.n_marks, dtype=np.int64)
    psv[tagged] = (soa.read_tag[soa.mark_read[tagged]] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    cand = np.repeat(np.arange(C), np.diff(soa.cand_off.astype(np.int64)))
    big = np.int64(1) << 40
    lo = np.full(C, big, dtype=np.int64)
    hi = np.full(C, -1, dtype=np.int64)
    np.minimum.at(lo, cand[tagged], psv[tagged])
    np.maximum.at(hi, cand[tagged], psv[tagged])
    cls[hi >= 0] = 1
    cls[(hi >= 0) & (lo != hi)] = 2
    return cls


def generate_phased_callset(vcf_path, sam_home, svlen_thres, suppread_thres, thread, include_all_ctgs,
                            ctx=None):
    tab, soa = generate_callinfo(vcf_path, read_hap_bam(sam_home, thread, include_all_ctgs), include_all_ctgs)
    logging.info('integrate read weight information')
    logging.info('calculate read weight statistics')
    logging.info('predict SV haplotypes in the callset')
    pred, ps = engine.run_ef(soa, svlen_thres, suppread_thres, ctx=ctx)
    # equal (chrom, pos) keys are rare; compute classes only when some emitted rows collide
    classes = None
    idx = np.nonzero(pred)[0]
    if idx.size > 1:
        key = np.stack([np.searchsorted(tab.ctg_off, idx, side='right') - 1, tab.pos[idx]])
        if np.unique(key, axis=1).shape[1] != idx.size:
            classes = ps_classes(soa)
    return assemble_rows(tab, pred, ps, classes)
