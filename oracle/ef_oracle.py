# coding=utf-8
"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of Duet's step E/F.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (duet_amd/) never does and fails loudly when the HIP library is missing.

What it restates (reference = /root/reference/src/duet, cited per function):
  text caller VCF + `samtools view` text of the haplotagged BAMs  ->  exact bytes of phased_sv.vcf.

Parity status: PINNED.  tests/test_oracle_golden.py checks this module against
  * the 38-row known-answer table captured from the imported reference's predict_hp
    (tests/golden/kat_predict_hp.json),
  * byte-for-byte phased_sv.vcf outputs of the imported reference on committed seeded inputs
    (tests/golden/cases/*), produced by tests/golden/make_golden.py in the development container.

Written for clarity, not speed: plain Python containers, one pass per concept.  The numbered quirks
(Q1..Q24) refer to SURVEY.md section 8a.
"""

import bisect
import os

PC_MAX = 8100          # sv_phasing_fn.py:76,88,201 -- reads with PC above this never vote
DEFAULT_CHROMS = [str(i) for i in range(1, 23)] + ['X', 'Y']     # read_file.py:7-12


# ---------------------------------------------------------------------------------------------
# I1  contig universe  (read_file.py:6-16)
# ---------------------------------------------------------------------------------------------

def chrom_list(include_all_ctgs=False, all_ctg_names=None):
    """Default: 1..22,X,Y. With -a the reference asks `tabix --list-chroms` on the pileup VCF;
    here the caller supplies that listing (`all_ctg_names`)."""
    if not include_all_ctgs:
        return list(DEFAULT_CHROMS)
    if all_ctg_names is None:
        raise ValueError('include_all_ctgs needs the contig listing')
    return list(all_ctg_names)


# ---------------------------------------------------------------------------------------------
# I2  read tag table  (sv_phasing_fn.py:11-34)
# ---------------------------------------------------------------------------------------------

def tags_from_sam_text(text):
    """`samtools view` text -> {read name: (hap, ps, pc)}.

    A line contributes iff its second-to-last whitespace token contains 'PC:i:' (:28); the last three
    tokens are then taken to be HP:i:, PC:i:, PS:i: in that order and sliced from character 5 (:29).
    Later lines overwrite earlier ones (Q2). The text is cut at '\\n' and the final piece dropped (:25)."""
    table = {}
    for line in text.split('\n')[:-1]:
        tok = line.split()
        if 'PC:i:' in tok[-2]:
            table[tok[0]] = (int(tok[-3][5:]), int(tok[-1][5:]), int(tok[-2][5:]))
    return table


def load_tag_tables(snp_phasing_dir, chroms):
    """Per contig: <dir>/chr<c>.bam else <dir>/<c>.bam else no table (:19-24). The oracle reads the
    text that a `samtools view` stand-in would print: the file <that bam>.sam next to it."""
    tables = []
    for c in chroms:
        stem = None
        for cand in ('chr' + c + '.bam', c + '.bam'):
            full = os.path.join(snp_phasing_dir, cand)
            if os.path.exists(full) or os.path.exists(full + '.sam'):     # fixtures keep only the text
                stem = full
                break
        if stem is None:
            tables.append({})
            continue
        with open(stem + '.sam', 'r') as f:
            tables.append(tags_from_sam_text(f.read()))
    return tables


# ---------------------------------------------------------------------------------------------
# I3  caller VCF  (read_file.py:18-77)
# ---------------------------------------------------------------------------------------------

def tokenise(vcf_path):
    """Every line stripped and split on whitespace (read_file.py:18-23), header lines included."""
    with open(vcf_path, 'r') as f:
        return [ln.strip().split() for ln in f.readlines()]


def _opt_int(txt):
    return 0 if txt == '.' else int(txt)


def contig_records(all_tokens, c):
    """Records of contig c with the derived columns appended in the reference's order.

    Returns a list of token lists; positions 10.. hold: svlen, svtype, [support], [read names],
    [gt, n1, n2].  The bracketed groups are appended only if the contig's FIRST record has them (Q5);
    when one is missing, everything after it shifts left, exactly as in the reference."""
    recs = [list(t) for t in all_tokens if t[0] in ('chr' + c, c)]        # t[0] on a blank line raises, as upstream
    if not recs:
        return recs
    infos = [r[7].split(';') for r in recs]
    # svlen: first INFO item containing 'SVLEN='; absent or 'SVLEN=.' counts as 0; 'SVLEN=>n' accepted (Q20)
    for r, items in zip(recs, infos):
        hit = [x for x in items if 'SVLEN=' in x]
        item = 'SVLEN=0' if (not hit or hit[0] == 'SVLEN=.') else hit[0]
        r.append(int(item[7:]) if '>' in item else int(item[6:]))
    for r, items in zip(recs, infos):
        r.append([x for x in items if 'SVTYPE=' in x][0][7:])
    # support count: SUPPORT= (Sniffles2/SVIM), or SR= / RE= (cuteSV) -- layout decided by record 0
    supp = [[x for x in items if ('SUPPORT=' in x or 'SR=' in x or 'RE=' in x)] for items in infos]
    if supp[0]:
        cut = 8 if 'SUPPORT=' in supp[0][0] else 3
        for r, s in zip(recs, supp):
            r.append(int(s[0][cut:]))
    # read names: RNAMES= (cuteSV, Sniffles2) or READS= (SVIM)
    rn = [[x for x in items if ('RNAMES=' in x or 'READS=' in x)] for items in infos]
    if rn[0]:
        cut = 7 if 'RNAMES=' in rn[0][0] else 6
        for r, s in zip(recs, rn):
            r.append(s[0][cut:].split(','))
    # sample column
    gts = [r[9].split(':') for r in recs]
    if len(gts[0]) > 4:                       # cuteSV GT:DR:DV:PL:GQ -> gt, DR, DV
        for r, g in zip(recs, gts):
            r.extend([g[0], _opt_int(g[1]), _opt_int(g[2])])
    elif len(gts[0]) >= 3:
        if gts[0][-1].find(',') == -1:        # Sniffles2 GT:GQ:DR:DV -> gt, GQ (sic, Q4), DR
            for r, g in zip(recs, gts):
                r.extend([g[0], _opt_int(g[1]), _opt_int(g[2])])
        else:                                 # SVIM GT:DP:AD -> gt, AD ref, AD alt
            for r, g in zip(recs, gts):
                last = g[-1]
                k = last.find(',')
                r.extend([g[0], _opt_int(last[:k]), _opt_int(last[k + 1:])])
    return recs


class Candidate(object):
    __slots__ = ('chrom', 'pos', 'ref', 'alt', 'svlen', 'svtype', 'marks', 'svread', 'gt', 'refread',
                 'contig_index')


def build_callset(all_tokens, chroms, tag_tables):
    """generate_callinfo (sv_phasing_fn.py:36-68): per contig in list order, per record in file
    order; each mark name is looked up ONLY in its own contig's table (Q24) -> (hap, ps, pc) or None."""
    out = []
    for k, c in enumerate(chroms):
        table = tag_tables[k]
        for r in contig_records(all_tokens, c):
            cd = Candidate()
            cd.contig_index = k
            cd.chrom, cd.pos, cd.ref, cd.alt = r[0], int(r[1]), r[3], r[4]
            cd.svlen = abs(r[10])
            cd.svtype = r[11]
            cd.svread = r[12]
            cd.marks = [table.get(name) for name in r[13]]
            cd.gt = r[14]
            cd.refread = r[15]
            out.append(cd)
    return out


# ---------------------------------------------------------------------------------------------
# E2-E4  filter, PS-class, seed sets  (sv_phasing_fn.py:189-203)
# ---------------------------------------------------------------------------------------------

def passes_filter(cd, svlen_thres, suppread_thres):
    return cd.svlen >= svlen_thres and cd.svread >= suppread_thres and cd.gt != './.'      # Q21


def ps_class(cd):
    """0 / 1 / 2 = no / one / several distinct PS among ALL tagged marks -- no PC test here (Q7)."""
    n = len(set(m[1] for m in cd.marks if m is not None))
    return 0 if n == 0 else (1 if n == 1 else 2)


def seed_ps(cd):
    """PS of the first mark that is tagged with pc <= PC_MAX, or None (:199-203)."""
    for m in cd.marks:
        if m is not None and m[2] <= PC_MAX:
            return m[1]
    return None


# ---------------------------------------------------------------------------------------------
# F1-F3  vote, features, decision  (sv_phasing_fn.py:70-183)
# ---------------------------------------------------------------------------------------------

def nearest_ps(sorted_ps, pos):
    """Element of the ascending seed array nearest to pos; ties go to the larger one (Q13; :107-111)."""
    i = bisect.bisect_left(sorted_ps, pos)
    lo = max(i - 1, 0)
    hi = min(i, len(sorted_ps) - 1)
    return sorted_ps[lo] if abs(pos - sorted_ps[lo]) < abs(pos - sorted_ps[hi]) else sorted_ps[hi]


def vote(cd, cls, seeds):
    """-> (hap1, hap2, hap0, allhap, t1, t2, ps). `seeds` is the contig's seed set (a set)."""
    hap1 = hap2 = hap0 = allhap = t1 = t2 = ps = 0
    voters = [m for m in cd.marks if m is not None and m[2] <= PC_MAX]
    if cls == 1:                                  # :74-84
        for hap, mps, pc in voters:
            ps = mps                              # last voter wins; single-valued in this class
            if hap == 1:
                hap1 += 1
                t1 += pc
            elif hap == 2:
                hap2 += 1
                t2 += pc
        allhap = hap1 + hap2
    elif cls == 2:                                # :85-105
        allhap = len(voters)
        order = []                                # phase sets in first-seen order
        acc = {}
        for hap, mps, pc in voters:
            if mps not in seeds:
                continue
            if mps not in acc:
                acc[mps] = [0, 0, 0, 0, 0]        # n, n1, n2, sum1, sum2
                order.append(mps)
            a = acc[mps]
            a[0] += 1
            a[hap] += 1                           # hap is 1 or 2 for WhatsHap diploid tags
            a[2 + hap] += pc
        best = 0
        for mps in order:                         # strict '>' : first-seen wins ties (Q12)
            a = acc[mps]
            if a[0] > best:
                best = a[0]
                hap1, hap2, t1, t2, ps = a[1], a[2], a[3], a[4], mps
                hap0 = allhap - hap1 - hap2       # only assigned when a winner exists (Q11)
    if cls == 0 or (hap1 == 0 and hap2 == 0):     # :106-111
        ps = nearest_ps(sorted(seeds), cd.pos)
    return hap1, hap2, hap0, allhap, t1, t2, ps


def decide(cd, cls, seeds):
    """predict_hp (sv_phasing_fn.py:142-183) -> (pred in 0..3, ps). Python floats are IEEE binary64."""
    hap1, hap2, hap0, allhap, t1, t2, ps = vote(cd, cls, seeds)
    deg = len(cd.marks)
    hapread_ratio = allhap / deg                                   # :112  (Q6: by list length)
    a1 = t1 / hap1 if hap1 > 0 else 0
    a2 = t2 / hap2 if hap2 > 0 else 0
    sv_ratio = cd.svread / (cd.svread + cd.refread)                # :123  (ZeroDivisionError possible, Q22)
    lo, hi = min(t1, t2), max(t1, t2)
    totsc_ratio = hi / lo if lo > 0 else 0
    onehap_totsc = hi if lo == 0 else 0
    avgsc_diff = abs(a2 - a1)
    pred = 0
    if cls == 0:
        if sv_ratio == 1 and cd.svread >= 4:
            pred = 3
    elif cls == 2:
        if sv_ratio >= 0.72:
            if avgsc_diff <= 1369.50:
                pred = 3 if cd.svread >= 3 else 0
            else:
                pred = 3 if hap0 >= 6 else 0
    else:
        # :157-158 assigns 0 and falls through: the sv_num >= 20 test filters nothing (Q10)
        gate = hapread_ratio > 0.75 or avgsc_diff <= 2400
        if onehap_totsc != 0:
            if sv_ratio <= 0.24:
                pred = 0
            elif sv_ratio <= 0.9:
                if gate:
                    pred = 1 if a1 > 0 else 2
            else:
                if gate:
                    pred = 3
        else:
            if sv_ratio <= 0.3:
                pred = 0
            elif sv_ratio <= 0.45:
                pred = 0 if cd.refread > 10 else (1 if t1 > t2 else 2)
            elif sv_ratio <= 0.75:
                pred = 3 if totsc_ratio <= 9.72 else (1 if t1 > t2 else 2)
            else:
                pred = 3
    return pred, ps


# ---------------------------------------------------------------------------------------------
# F0, S1  driver and sort  (sv_phasing_fn.py:185-230)
# ---------------------------------------------------------------------------------------------

HP_TEXT = {1: '1|0', 2: '0|1', 3: '1|1'}


def phase_callset(callset, chroms, svlen_thres, suppread_thres, want_trace=False):
    """-> sorted list of output rows (dicts). With want_trace also returns, per input candidate in
    callset order, (kept, class, pred, ps) -- pred/ps are None when predict_hp was never called.

    Candidates are attributed to a contig by their CHROM text ('chr'+c or c), as upstream does
    (:198, :208), not by the table they were joined against."""
    kept = [passes_filter(cd, svlen_thres, suppread_thres) for cd in callset]
    cls = [ps_class(cd) if k else None for cd, k in zip(callset, kept)]
    spell = [('chr' + c, c) for c in chroms]
    seeds = [set() for _ in chroms]
    for ctg in range(len(chroms)):
        for cd, k, p in zip(callset, kept, cls):
            if k and p == 1 and cd.chrom in spell[ctg]:
                s = seed_ps(cd)
                if s is not None:
                    seeds[ctg].add(s)
    trace = [[k, p, None, None] for k, p in zip(kept, cls)]
    rows = []
    for ctg in range(len(chroms)):
        if not seeds[ctg]:
            continue                                                  # Q9
        for want in (0, 1, 2):
            for i, cd in enumerate(callset):
                if cd.chrom not in spell[ctg] or not kept[i] or cls[i] != want:
                    continue
                pred, ps = decide(cd, want, seeds[ctg])
                trace[i][2], trace[i][3] = pred, int(ps)
                if pred == 0:
                    continue
                signed = cd.svlen if cd.svtype in ('INS', 'DUP') else -cd.svlen      # Q16
                rows.append(dict(ps=ps, hp=HP_TEXT[pred], chrom=cd.chrom, pos=cd.pos, svlen=signed,
                                 svtype=cd.svtype, ref=cd.ref, alt=cd.alt))
    rows.sort(key=lambda r: (r['chrom'], r['pos']))                   # stable; chrom compared as text (Q15)
    return (rows, trace) if want_trace else rows


# ---------------------------------------------------------------------------------------------
# W1, W2  output text  (write_file.py:6-45)
# ---------------------------------------------------------------------------------------------

_FIXED_HEADER = (
    '##fileformat=VCFv4.2\n'
    '##source=Duet\n'
    '##ALT=<ID=INS,Description="Insertion of novel sequence relative to the reference">\n'
    '##ALT=<ID=DEL,Description="Deletion relative to the reference">\n'
    '##FILTER=<ID=PASS,Description="SV calls passed phasing criterion">\n'
    '##INFO=<ID=SVLEN,Number=1,Type=Integer,Description="Estimated length of the variant">\n'
    '##FORMAT=<ID=HP,Number=1,Type=String,Description="Haplotype of the SV call">\n'
    '##FORMAT=<ID=PS,Number=1,Type=String,Description="Phase set which the SV call belongs to">\n'
)
_COLUMNS = '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tVALUE\n'


def header_text(all_tokens, chroms, include_all_ctgs):
    """Fixed lines, then the caller VCF's ##contig lines (first whitespace token of each): default
    mode walks the 24 listed contigs in list order and copies every line naming chr<c> or <c> (Q19);
    -a mode copies every ##contig line in file order."""
    out = [_FIXED_HEADER]
    if not include_all_ctgs:
        for c in chroms[:24]:
            a, b = '##contig=<ID=chr' + c + ',', '##contig=<ID=' + c + ','
            for t in all_tokens:
                if a in t[0] or b in t[0]:
                    out.append(t[0] + '\n')
    else:
        for t in all_tokens:
            if '##contig=<ID=' in t[0]:
                out.append(t[0] + '\n')
    out.append(_COLUMNS)
    return ''.join(out)


def rows_text(rows):
    """One line per phased call, ids renumbered Duet.1..N after the sort (Q18), INFO with literal
    angle brackets around the type (Q17), FORMAT HP:PS."""
    out = []
    for i, r in enumerate(rows):
        out.append('%s\t%d\tDuet.%d\t%s\t%s\t.\tPASS\tSVLEN=%d;SVTYPE=<%s>\tHP:PS\t%s:%d\n' % (
            r['chrom'], r['pos'], i + 1, r['ref'], r['alt'], r['svlen'], r['svtype'], r['hp'], int(r['ps'])))
    return ''.join(out)


def sv_phasing_text(home, svlen_thres=50, suppread_thres=2, include_all_ctgs=False, all_ctg_names=None,
                    want_trace=False):
    """sv_phasing (sv_phasing.py:8-20) as a pure function: <home>/sv_calling/variants.vcf and
    <home>/snp_phasing/*.bam.sam -> the text of <home>/phased_sv.vcf."""
    chroms = chrom_list(include_all_ctgs, all_ctg_names)
    toks = tokenise(os.path.join(home, 'sv_calling', 'variants.vcf'))
    head = header_text(toks, chroms, include_all_ctgs)
    tables = load_tag_tables(os.path.join(home, 'snp_phasing'), chroms)
    callset = build_callset(toks, chroms, tables)
    res = phase_callset(callset, chroms, svlen_thres, suppread_thres, want_trace)
    if want_trace:
        return head + rows_text(res[0]), res[1], callset
    return head + rows_text(res)
