#!/usr/bin/env python3
# coding=utf-8
"""Accuracy evaluator for phased SV callsets (restatement of src/scripts/evaluation.py).

Same command line and the same ten numbers: precision / recall / F1 of SV calling, genotyping and phasing of a
callset against a truth set, each call matched to the NEAREST same-type truth call of its contig (ties to the
right-hand neighbour, evaluation.py:117-125) within --refdist and a length ratio of at least --pctsim, phasing
scored per phase set with the better of the two haplotype labelings (evaluation.py:143-148).

Differences in form only: records are flattened into per-(contig, type) position-sorted numpy arrays once, and
each phase set is matched with one vectorised `searchsorted` instead of Python loops over dicts.

`evaluation_gpu` is the same scoring on the MI355X (duet_eval_run_host, duet_amd/csrc/duet_eval.hip): the host flattens the
parsed records (`flatten`), the device matches, builds the id sets and picks each phase set's labelling, and the six set
sizes come back; the ten numbers are the same binary64 quotients.  `--gpu` on the command line selects it.
"""

import argparse
import sys

import numpy as np

LABELS = [str(i) for i in range(1, 23)] + ['X', 'Y']
CHROMS = ['chr' + c for c in LABELS]
_ALT_TYPES = ('<INS>', '<DEL>', '<DUP:TANDEM>', '<DUP:INT>', '<DUP>')


def _tokens(path):
    with open(path, 'r') as fh:
        return [ln.strip().split() for ln in fh]


def parse_bed(path):
    """Per contig: list of closed integer ranges (evaluation.py:25-33)."""
    spans = {c: [] for c in LABELS}
    for t in _tokens(path):
        if t[0][:3] == 'chr' and t[0][3:] in spans:
            spans[t[0][3:]].append((int(t[1]), int(t[2])))
    return spans


def parse_vcf(vcf_path, skip_phasing, bed_path=''):
    """-> list of dicts {chr, pos, id, hp, ps, len, type}, file order (evaluation.py:35-97)."""
    bed = parse_bed(bed_path) if bed_path != '' else None
    out = []
    for s in _tokens(vcf_path):
        if s[0][0] == '#':
            continue
        if s[0][3:] not in LABELS:          # whatever the first three characters are (evaluation.py:43)
            continue
        if 'SVLEN=.' in s[7]:
            continue
        if not any(k in s[7] or k in s[4] for k in ('INS', 'DEL', 'DUP')):
            continue
        hp = s[-1][:3]
        if hp[0] == '.':
            hp = '0' + hp[1:]
        if hp[2] == '.':
            hp = hp[:2] + '0'
        if hp[1] == '/':
            if not skip_phasing and hp != '1/1':
                continue
            hp = hp[0] + '|' + hp[2]
            ps = s[0]
        else:
            ps = s[0] + '_' + s[-1][s[-1].rfind(':'):]
        rec = {'chr': s[0], 'pos': int(s[1]), 'id': s[2] + s[0] + s[1], 'hp': hp, 'ps': ps}
        items = s[7].split(';')
        if 'SVLEN' in s[7]:
            item = [x for x in items if 'SVLEN' in x][0]
            rec['len'] = abs(int(item[7:])) if 'SVLEN=>' in s[7] else abs(int(item[6:]))
            rec['type'] = s[4][1:-1] if s[4] in _ALT_TYPES else [x for x in items if 'SVTYPE' in x][0][7:]
            if 'DUP' in rec['type']:
                rec['type'] = 'INS'
        else:
            d = len(s[3]) - len(s[4])
            if d > 0:
                rec['len'], rec['type'] = d, 'DEL'
            if d < 0:
                rec['len'], rec['type'] = -d, 'INS'
        if bed is not None:
            p = int(s[1])
            if not any(a <= p <= b for a, b in bed[s[0][3:]]):
                continue
        if rec['len'] < 50 or rec['hp'] == '0|0':                     # KeyError for equal-length REF/ALT, as upstream
            continue
        out.append(rec)
    return out


def _nearest(base_pos, call_pos):
    """Index of the nearest base position per call; ties and the insertion point rule of evaluation.py:117-125."""
    idx = np.searchsorted(base_pos, call_pos)
    n = len(base_pos)
    if n == 0:
        raise IndexError('list index out of range')                   # upstream indexes the empty truth list
    at_end = idx == n
    left = np.maximum(idx - 1, 0)
    right = np.minimum(idx, n - 1)
    take_left = at_end | ((idx > 0) & (np.abs(call_pos - base_pos[right]) > np.abs(call_pos - base_pos[left])))
    return np.where(take_left, left, right)


def evaluation(baseinfo, callinfo, threshold_tp_range, ratio):
    call_tp, call_gt, call_hp, base_tp, base_gt, base_hp = set(), set(), set(), set(), set(), set()
    avg_sv_num = len(callinfo) / len(set(s['ps'] for s in callinfo))
    for chrom in CHROMS:
        base = {}
        for svtype in ('INS', 'DEL'):
            rows = sorted((s for s in baseinfo if s['chr'] == chrom and s['type'] == svtype), key=lambda r: r['pos'])
            base[svtype] = (rows, np.array([r['pos'] for r in rows], dtype=np.int64),
                            np.array([r['len'] for r in rows], dtype=np.int64))
        on_chrom = [s for s in callinfo if s['chr'] == chrom]
        for ps in set(s['ps'] for s in on_chrom):
            same_c, same_b, flip_c, flip_b = set(), set(), set(), set()
            for svtype in ('INS', 'DEL'):
                calls = [s for s in on_chrom if s['ps'] == ps and s['type'] == svtype]
                if not calls:
                    continue
                rows, bpos, blen = base[svtype]
                cpos = np.array([c['pos'] for c in calls], dtype=np.int64)
                clen = np.array([c['len'] for c in calls], dtype=np.int64)
                j = _nearest(bpos, cpos)
                ok = (np.abs(cpos - bpos[j]) <= threshold_tp_range) & \
                     (np.minimum(clen, blen[j]) / np.maximum(clen, blen[j]) >= ratio)
                for ci in np.nonzero(ok)[0]:
                    c, b = calls[int(ci)], rows[int(j[ci])]
                    call_tp.add(c['id'])
                    base_tp.add(b['id'])
                    het = ('1|0', '0|1')
                    if (c['hp'] in het and b['hp'] in het) or c['hp'] == b['hp'] == '1|1':
                        call_gt.add(c['id'])
                        base_gt.add(b['id'])
                    if c['hp'] == b['hp']:
                        same_c.add(c['id'])
                        same_b.add(b['id'])
                    if c['hp'] == b['hp'] == '1|1' or (c['hp'], b['hp']) in (('0|1', '1|0'), ('1|0', '0|1')):
                        flip_c.add(c['id'])
                        flip_b.add(b['id'])
            if len(same_c) + len(same_b) > len(flip_c) + len(flip_b):
                call_hp |= same_c
                base_hp |= same_b
            else:
                call_hp |= flip_c
                base_hp |= flip_b

    def prf(tp_c, tp_b):
        p, r = len(tp_c) / len(callinfo), len(tp_b) / len(baseinfo)
        return p, r, 2 * p * r / (p + r)

    return (avg_sv_num,) + prf(call_tp, base_tp) + prf(call_gt, base_gt) + prf(call_hp, base_hp)


_HP_FIXED = {'1|0': 0, '0|1': 1, '1|1': 2}


def flatten(baseinfo, callinfo):
    """The parsed records as the flat arrays of include/duet_ef.h's duet_eval_problem.  Raises IndexError where upstream
    indexes an empty truth list (a call whose (contig, type) has no truth record, evaluation.py:120-125)."""
    hp_code = dict(_HP_FIXED)

    def code(hp):
        if hp not in hp_code:
            hp_code[hp] = len(hp_code)
            if len(hp_code) > 255:
                raise ValueError('more than 255 distinct haplotype strings')
        return hp_code[hp]

    types = {'INS': 0, 'DEL': 1}
    contig = {c: k for k, c in enumerate(CHROMS)}
    base_uid, call_uid, groups = {}, {}, {}
    lists = [[] for _ in range(2 * len(CHROMS))]
    for r in baseinfo:
        base_uid.setdefault(r['id'], len(base_uid))
        if r['chr'] in contig and r['type'] in types:
            lists[2 * contig[r['chr']] + types[r['type']]].append(r)
    base_off = np.zeros(len(lists) + 1, dtype=np.int64)
    rows = []
    for i, lst in enumerate(lists):
        lst.sort(key=lambda r: r['pos'])                 # stable, like sorted(..., key=itemgetter('pos')) upstream
        rows.extend(lst)
        base_off[i + 1] = len(rows)
    a = dict(base_off=base_off, base_pos=[r['pos'] for r in rows], base_len=[r['len'] for r in rows],
             base_uid=[base_uid[r['id']] for r in rows], base_hp=[code(r['hp']) for r in rows])
    ck, cp, cl, cu, cg, ch = [], [], [], [], [], []
    for r in callinfo:
        call_uid.setdefault(r['id'], len(call_uid))
        if r['chr'] not in contig:
            continue
        g = groups.setdefault((r['chr'], r['ps']), len(groups))
        if r['type'] not in types:
            continue
        key = 2 * contig[r['chr']] + types[r['type']]
        if base_off[key + 1] == base_off[key]:
            raise IndexError('list index out of range')
        ck.append(key); cp.append(r['pos']); cl.append(r['len']); cu.append(call_uid[r['id']]); cg.append(g); ch.append(code(r['hp']))
    a.update(call_key=ck, call_pos=cp, call_len=cl, call_uid=cu, call_group=cg, call_hp=ch, n_groups=len(groups),
             n_base_uid=len(base_uid), n_call_uid=len(call_uid))
    for k in ('base_pos', 'base_len', 'call_pos', 'call_len'):
        v = np.asarray(a[k], dtype=np.int64)
        if v.size and (int(v.min()) < 0 or int(v.max()) > 0xFFFFFFFF):
            raise ValueError(k + ' outside the 32-bit range of the device arrays')
    return a


def evaluation_gpu(baseinfo, callinfo, threshold_tp_range, ratio, ctx=None):
    """evaluation() on the GPU: same ten numbers, same exceptions."""
    from duet_amd import engine
    avg_sv_num = len(callinfo) / len(set(s['ps'] for s in callinfo))
    a = flatten(baseinfo, callinfo)
    if ctx is None:
        ctx = engine.default_context()
    if threshold_tp_range < 0:
        a['call_key'] = []                               # abs(...) <= negative never holds: nothing matches
        for k in ('call_pos', 'call_len', 'call_uid', 'call_group', 'call_hp'):
            a[k] = []
    n = ctx.eval_counts(a, max(int(threshold_tp_range), 0), ratio)

    def prf(tp_c, tp_b):
        p, r = tp_c / len(callinfo), tp_b / len(baseinfo)
        return p, r, 2 * p * r / (p + r)

    return (avg_sv_num,) + prf(n.call_tp, n.base_tp) + prf(n.call_gt, n.base_gt) + prf(n.call_hp, n.base_hp)


def parse_args(argv):
    ap = argparse.ArgumentParser(description='precision / recall / F1 of SV calling, genotyping and phasing against a truth set')
    ap.add_argument('callset', type=str, help='VCF of phased SV calls to score')
    ap.add_argument('truthset', type=str, help='VCF of the phased truth set')
    ap.add_argument('-r', '--refdist', type=int, default=1000,
                    help='a call matches a truth call at most this many bp away [%(default)s]')
    ap.add_argument('-p', '--pctsim', type=float, default=0,
                    help='minimum shorter/longer SV length ratio for a match [%(default)s]')
    ap.add_argument('-b', '--bed_file', type=str, help='BED file; only calls inside its regions are scored')
    ap.add_argument('--skip_phasing', action='store_true',
                    help='score calling and genotyping only')
    ap.add_argument('--gpu', action='store_true', help='score on the MI355X (libduet_ef.so) instead of with numpy')
    return ap.parse_args(argv)


def main(argv):
    args = parse_args(argv)
    bed = args.bed_file or ''
    truth, calls = (parse_vcf(path, args.skip_phasing, bed) for path in (args.truthset, args.callset))
    res = (evaluation_gpu if args.gpu else evaluation)(truth, calls, args.refdist, args.pctsim)
    # upstream's report lines, verbatim (evaluation.py:184-189)
    if not args.skip_phasing:
        print('Average SV number per phase set is', res[0])
    for what, at in (('calling', 1), ('genotyping', 4), ('phasing', 7)):
        if what == 'phasing' and args.skip_phasing:
            continue
        print('The precision, recall and F1 score of SV %s are' % what, res[at], res[at + 1], res[at + 2])


if __name__ == '__main__':
    main(sys.argv[1:])
