# coding=utf-8
"""Host-side ingest of the caller VCF into flat arrays (mirror of src/duet/read_file.py).

Same entry points as upstream -- init_chrom_list (read_file.py:6), read_file (:18), parse_vcf (:25) --
but parse_vcf yields one CallTable (structure of arrays over all contigs in callset order) instead
of nested Python lists, because its consumer is the C ABI of include/duet_ef.h.

Behaviour kept from upstream, including the accidental parts (numbers = SURVEY.md section 8a quirks):
whitespace tokenising of every line (Q3); CHROM matched as 'chr'+c or c (Q23); the INFO/FORMAT
layout of a contig is decided by that contig's FIRST record (Q5); Sniffles' GQ lands in the
"reference reads" column (Q4); 'SVLEN=.' / missing -> 0 and 'SVLEN=>n' accepted (Q20).
"""

import shlex
import subprocess

import numpy as np


def init_chrom_list(include_all_ctgs, home):
    """Contig universe: 1..22, X, Y -- or, with -a, `tabix --list-chroms` of the pileup VCF."""
    if not include_all_ctgs:
        return [str(n) for n in range(1, 23)] + ['X', 'Y']
    listing = subprocess.check_output(shlex.split('tabix --list-chroms ' + home + '/snp_calling/pileup.vcf.gz'))
    return listing.decode('ascii').split('\n')[:-1]


def read_file(vcf_path):
    """All lines of the file, stripped and whitespace-split."""
    with open(vcf_path, 'r') as fh:
        return [ln.strip().split() for ln in fh]


class CallTable(object):
    """Candidates of every listed contig, callset order. Numeric columns are numpy arrays; text
    columns stay Python lists (they are only copied to the output rows)."""

    def __init__(self, chrom_list):
        self.chrom_list = chrom_list
        self.ctg_off = np.zeros(len(chrom_list) + 1, dtype=np.int64)
        self.chrom = []
        self.ref = []
        self.alt = []
        self.svtype = []
        self.gt = []
        self.names = []                 # per candidate: list of read names (marks, list order)
        self.pos = None
        self.svlen_abs = None
        self.svread = None
        self.refread = None

    def __len__(self):
        return len(self.chrom)


def _first_with(items, *needles):
    for it in items:
        for nd in needles:
            if nd in it:
                return it
    return None


def _int_or_zero(txt):
    return 0 if txt == '.' else int(txt)


def _contig_columns(recs):
    """Derived columns of one contig's records (read_file.py:33-76), computed column by column over ALL records
    like upstream, so that a malformed record raises what upstream raises first.  The layout switches are taken from
    recs[0]; later records are parsed with that layout and raise like upstream if they lack it."""
    infos = [r[7].split(';') for r in recs]

    def lacks():
        # upstream silently drops the column and then fails on shifted indices (TypeError/IndexError)
        return ValueError('first record of contig %s lacks a support count, a read-name list or a '
                          '>=3-field sample column; the reference cannot process this layout' % recs[0][0])

    svlen = []
    for items in infos:                                            # :34-36
        it = _first_with(items, 'SVLEN=')
        if it is None or it == 'SVLEN=.':
            it = 'SVLEN=0'
        svlen.append(int(it[7:]) if '>' in it else int(it[6:]))
    svtype = []
    for items in infos:                                            # :38
        it = _first_with(items, 'SVTYPE=')
        if it is None:
            raise IndexError('list index out of range')            # upstream: [][0]
        svtype.append(it[7:])
    first_supp = _first_with(infos[0], 'SUPPORT=', 'SR=', 'RE=')
    if first_supp is None:
        raise lacks()
    supp_cut = 8 if 'SUPPORT=' in first_supp else 3
    svread = []
    for items in infos:                                            # :40-47
        it = _first_with(items, 'SUPPORT=', 'SR=', 'RE=')
        if it is None:
            raise IndexError('list index out of range')
        svread.append(int(it[supp_cut:]))
    first_rn = _first_with(infos[0], 'RNAMES=', 'READS=')
    if first_rn is None:
        raise lacks()
    rn_cut = 7 if 'RNAMES=' in first_rn else 6
    names = []
    for items in infos:                                            # :48-55
        it = _first_with(items, 'RNAMES=', 'READS=')
        if it is None:
            raise IndexError('list index out of range')
        names.append(it[rn_cut:].split(','))
    subs = [r[9].split(':') for r in recs]                          # :56
    first_fmt = subs[0]
    if len(first_fmt) < 3:
        raise lacks()
    if len(first_fmt) > 4:
        fmt_kind = 0                    # cuteSV   GT:DR:DV:PL:GQ -> DR
    elif first_fmt[-1].find(',') == -1:
        fmt_kind = 1                    # Sniffles GT:GQ:DR:DV    -> GQ (sic)
    else:
        fmt_kind = 2                    # SVIM     GT:DP:AD       -> AD[0]
    gt = [sub[0] for sub in subs]
    if fmt_kind == 2:                                              # :70-76
        refread = [_int_or_zero(sub[-1][:sub[-1].find(',')]) for sub in subs]
        for sub in subs:
            _int_or_zero(sub[-1][sub[-1].find(',') + 1:])           # parsed (and may raise) upstream too
    else:                                                          # :58-69
        refread = [_int_or_zero(sub[1]) for sub in subs]
        for sub in subs:
            _int_or_zero(sub[2])
    return svlen, svtype, svread, names, gt, refread


def parse_vcf(vcf_file, include_all_ctgs, tokens=None):
    """Caller VCF -> CallTable. `tokens` lets a caller reuse an existing read_file() result."""
    chrom_list = init_chrom_list(include_all_ctgs, vcf_file[:len(vcf_file) - 24])     # <home>/sv_calling/variants.vcf
    if tokens is None:
        tokens = read_file(vcf_file)
    # CHROM text -> contig index ('chr'+c and c both name contig c)
    owner = {}
    for k, c in enumerate(chrom_list):
        for nm in ('chr' + c, c):
            if owner.setdefault(nm, k) != k:
                # only reachable with -a when the listing holds both 'x' and 'chrx' (or a name twice):
                # upstream then evaluates such records once per alias; not supported here
                raise NotImplementedError('contig list names %r twice' % nm)
    per_ctg = [[] for _ in chrom_list]
    for t in tokens:
        k = owner.get(t[0])            # t[0] on a blank line raises IndexError, as upstream (read_file.py:30)
        if k is not None:
            per_ctg[k].append(t)

    tab = CallTable(chrom_list)
    pos, svlen_abs, svread_all, refread_all = [], [], [], []
    for k, recs in enumerate(per_ctg):
        if recs:
            svlen, svtype, svread, names, gt, refread = _contig_columns(recs)
            tab.chrom.extend(r[0] for r in recs)
            tab.ref.extend(r[3] for r in recs)
            tab.alt.extend(r[4] for r in recs)
            tab.svtype.extend(svtype)
            tab.gt.extend(gt)
            tab.names.extend(names)
            pos.extend(int(r[1]) for r in recs)
            svlen_abs.extend(abs(v) for v in svlen)
            svread_all.extend(svread)
            refread_all.extend(refread)
        tab.ctg_off[k + 1] = len(tab.chrom)
    for recs in per_ctg:
        for r in recs:
            if len(r) > 10:
                # upstream appends its derived columns AFTER the last token (read_file.py:37), so an 11th token
                # shifts them and generate_callinfo fails on the shifted indices (sv_phasing_fn.py:47,62)
                raise TypeError("'int' object is not iterable")
    tab.pos = np.array(pos, dtype=np.int64)
    tab.svlen_abs = np.array(svlen_abs, dtype=np.int64)
    tab.svread = np.array(svread_all, dtype=np.int64)
    tab.refread = np.array(refread_all, dtype=np.int64)
    return tab
