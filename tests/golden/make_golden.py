#!/usr/bin/env python3
# coding=utf-8
"""Generate the golden fixtures in this directory from the UNMODIFIED reference.

Runs ONLY in the development container, where /root/reference exists.  It imports the reference's
own step-E/F modules (PYTHONPATH=/root/reference/src), replaces the `samtools view` subprocess with
a PATH shim that prints `<bam>.sam`, and records what the reference returns.
Nothing from the reference is copied: the fixtures are inputs made by duet_amd.synth and outputs
captured from the run.

    python tests/golden/make_golden.py            # everything (a few minutes)
    python tests/golden/make_golden.py --quick    # skip the 1M-mark config

Outputs
    kat_predict_hp.json      the 38 known-answer rows of SURVEY.md section 8c, re-captured
    kat_random.npz           20,000 random boundary-biased candidates through predict_hp
    cases/<name>/            full small work dirs: sv_calling/variants.vcf, snp_phasing/*.bam.sam,
                             expected phased_sv.vcf
    seeded.json              per (seed, dialect): sha256 of the inputs and of the expected output
                             (inputs are regenerated from the seed by the tests)
"""

import argparse
import hashlib
import json
import os
import shutil
import stat
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_SRC = '/root/reference/src'
sys.path.insert(0, REPO)

from duet_amd import synth  # noqa: E402


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, 'rb') as f:
        for blk in iter(lambda: f.read(1 << 20), b''):
            h.update(blk)
    return h.hexdigest()


def inputs_digest(home):
    """sha256 over the VCF and every .sam text in name order."""
    h = hashlib.sha256()
    with open(os.path.join(home, 'sv_calling', 'variants.vcf'), 'rb') as f:
        h.update(f.read())
    d = os.path.join(home, 'snp_phasing')
    for n in sorted(os.listdir(d)):
        if n.endswith('.sam'):
            h.update(n.encode())
            with open(os.path.join(d, n), 'rb') as f:
                h.update(f.read())
    return h.hexdigest()


def install_shims(tmp):
    shim = os.path.join(tmp, 'shim')
    os.makedirs(shim, exist_ok=True)
    p = os.path.join(shim, 'samtools')
    with open(p, 'w') as f:
        f.write('#!/bin/sh\nfor a; do last="$a"; done\ncat "$last.sam"\n')
    os.chmod(p, os.stat(p).st_mode | stat.S_IEXEC)
    p = os.path.join(shim, 'tabix')
    with open(p, 'w') as f:
        f.write('#!/bin/sh\nfor a; do last="$a"; done\ncat "$last.chroms"\n')
    os.chmod(p, os.stat(p).st_mode | stat.S_IEXEC)
    os.environ['PATH'] = shim + os.pathsep + os.environ['PATH']


def run_reference(home, svlen_thres=50, supp_thres=2, all_ctgs=False):
    from duet.sv_phasing import sv_phasing
    sv_phasing(home, svlen_thres, supp_thres, 4, all_ctgs)
    return os.path.join(home, 'phased_sv.vcf')


# ---------------------------------------------------------------------------------------------
# known-answer table
# ---------------------------------------------------------------------------------------------

def _marks(spec):
    """'1@500:300*2, u*3' -> reference svreadinfo list."""
    out = []
    n = 0
    for part in [p.strip() for p in spec.split(',') if p.strip()]:
        rep = 1
        if '*' in part:
            part, r = part.split('*')
            rep = int(r)
        for _ in range(rep):
            n += 1
            if part == 'u':
                out.append(['n%d' % n])
            else:
                hap, rest = part.split('@')
                ps, pc = rest.split(':')
                out.append(['n%d' % n, int(hap), int(ps), int(pc)])
    return out


KAT_ROWS = [
    # P, marks, svread, refread, pos
    (0, 'u*4', 4, 0, 7000), (0, 'u*3', 3, 0, 7000), (0, 'u*5', 5, 1, 7000), (0, 'u*4', 4, 0, 4250),
    (0, 'u*4', 4, 0, 10), (0, 'u*4', 4, 0, 99999),
    (1, '1@500:300*2', 6, 19, 1000), (1, '1@500:300*2, u', 3, 3, 1000), (1, '2@500:300*2, u', 3, 3, 1000),
    (1, '1@500:2500, u*3', 4, 4, 1000), (1, '1@500:2400, u*3', 4, 4, 1000), (1, '1@500:300*2', 10, 1, 1000),
    (1, '1@500:2500, u*3', 10, 1, 1000), (1, '1@500:300, 2@500:100', 3, 7, 1000),
    (1, '1@500:300, 2@500:100', 8, 11, 1000), (1, '1@500:300, 2@500:100', 4, 6, 1000),
    (1, '1@500:100, 2@500:100', 4, 6, 1000), (1, '1@500:972, 2@500:100', 5, 5, 1000),
    (1, '1@500:973, 2@500:100', 5, 5, 1000), (1, '2@500:973, 1@500:100', 5, 5, 1000),
    (1, '1@500:300, 2@500:100', 9, 1, 1000), (1, '1@500:9000, 2@500:8101', 9, 1, 7000),
    (1, '1@500:8100, 1@500:8101', 3, 3, 1000), (1, '1@500:0*2', 4, 6, 1000), (1, '1@500:300*20', 20, 20, 1000),
    (1, '1@500:300*2', 9, 1, 1000), (1, '1@500:9720, 2@500:1000', 3, 1, 1000),
    (1, '1@500:300, 2@500:100', 9, 11, 1000),
    (2, '1@500:300, 2@8000:100, u', 3, 1, 1000), (2, '1@500:300, 2@8000:100', 2, 0, 1000),
    (2, '1@500:300, 2@8000:100, u', 5, 2, 1000), (2, '1@500:300, 2@8000:100, u', 18, 7, 1000),
    (2, '1@500:2000*7, 2@777:10*6', 13, 0, 1000), (2, '1@500:2000*7, 2@777:10*5', 12, 0, 1000),
    (2, '2@8000:100, 1@500:300, u', 3, 0, 1000), (2, '2@8000:100, 1@500:300, 1@500:100', 3, 0, 1000),
    (2, '2@777:100, 1@888:300, u', 3, 0, 7000), (2, '1@500:1370, 1@500:1369, 2@8000:0', 3, 0, 1000),
]


def make_kat():
    from duet.sv_phasing_fn import predict_hp
    oneps = {500, 8000, 20000}
    rows = []
    for cls, spec, svread, refread, pos in KAT_ROWS:
        call = dict(svreadinfo=_marks(spec), pos=pos, svread=svread, refread=refread)
        pred, ps = predict_hp(call, cls, oneps)
        rows.append(dict(cls=cls, marks=spec, svread=svread, refread=refread, pos=pos,
                         pred=int(pred), ps=int(ps)))
    with open(os.path.join(HERE, 'kat_predict_hp.json'), 'w') as f:
        json.dump(dict(oneps=sorted(oneps), rows=rows), f, indent=1)
    print('kat_predict_hp.json: %d rows' % len(rows))


def make_kat_random(n=20000, seed=7):
    """Random candidates straight through the reference's predict_hp; PC values and read-count pairs
    concentrated on the decision thresholds. Stored as flat arrays (CSR over marks)."""
    from duet.sv_phasing_fn import predict_hp
    rng = synth.SplitMix(seed)
    pc_edge = np.array(synth._PC_EDGE, dtype=np.int64)
    ratio_edge = np.array(synth._RATIO_EDGE, dtype=np.int64)
    n_sets = 16
    oneps_sets = []
    for s in range(n_sets):
        k = 1 + rng.one(6)
        oneps_sets.append(sorted(set(int(x) for x in rng.between(k, 1, 40000))))
    cls_a = np.zeros(n, dtype=np.int64)
    set_a = np.zeros(n, dtype=np.int64)
    pos_a = np.zeros(n, dtype=np.int64)
    svr_a = np.zeros(n, dtype=np.int64)
    ref_a = np.zeros(n, dtype=np.int64)
    off = np.zeros(n + 1, dtype=np.int64)
    m_tag, m_hap, m_ps, m_pc = [], [], [], []
    pred_a = np.zeros(n, dtype=np.int64)
    ps_a = np.zeros(n, dtype=np.int64)
    for i in range(n):
        si = rng.one(n_sets)
        ops = oneps_sets[si]
        deg = 1 + rng.one(12)
        want = rng.one(3)
        foreign_ps = [int(x) for x in rng.between(3, 1, 40000)]
        marks = []
        pool = ops + foreign_ps
        base_ps = pool[rng.one(len(pool))]
        for j in range(deg):
            if want == 0 or rng.one(4) == 0:
                marks.append(['r%d' % j])
                continue
            hap = 1 + rng.one(2)
            pc = int(pc_edge[rng.one(len(pc_edge))]) if rng.one(2) else rng.one(9500)
            ps = base_ps if (want == 1 or rng.one(3)) else pool[rng.one(len(pool))]
            marks.append(['r%d' % j, hap, ps, pc])
        distinct = len(set(m[2] for m in marks if len(m) > 1))
        cls = 0 if distinct == 0 else (1 if distinct == 1 else 2)
        if rng.one(2):
            svread, refread = [int(x) for x in ratio_edge[rng.one(len(ratio_edge))]]
        else:
            svread, refread = 1 + rng.one(30), rng.one(30)
        pos = rng.one(45000)
        call = dict(svreadinfo=marks, pos=pos, svread=svread, refread=refread)
        pred, ps = predict_hp(call, cls, set(ops))
        cls_a[i], set_a[i], pos_a[i], svr_a[i], ref_a[i] = cls, si, pos, svread, refread
        pred_a[i], ps_a[i] = int(pred), int(ps)
        off[i + 1] = off[i] + deg
        for m in marks:
            if len(m) > 1:
                m_tag.append(1); m_hap.append(m[1]); m_ps.append(m[2]); m_pc.append(m[3])
            else:
                m_tag.append(0); m_hap.append(0); m_ps.append(0); m_pc.append(0)
    so = np.zeros(n_sets + 1, dtype=np.int64)
    sv = []
    for s, ops in enumerate(oneps_sets):
        so[s + 1] = so[s] + len(ops)
        sv.extend(ops)
    np.savez_compressed(os.path.join(HERE, 'kat_random.npz'), cls=cls_a.astype(np.int8), oneps_set=set_a.astype(np.int16),
                        pos=pos_a.astype(np.int32), svread=svr_a.astype(np.int32), refread=ref_a.astype(np.int32),
                        off=off.astype(np.int32), m_tagged=np.array(m_tag, dtype=np.int8),
                        m_hap=np.array(m_hap, dtype=np.int8), m_ps=np.array(m_ps, dtype=np.int32),
                        m_pc=np.array(m_pc, dtype=np.int32), oneps_off=so.astype(np.int32),
                        oneps_val=np.array(sv, dtype=np.int32), pred=pred_a.astype(np.int8), ps=ps_a.astype(np.int32))
    hist = {}
    for c, p in zip(cls_a, pred_a):
        hist[(int(c), int(p))] = hist.get((int(c), int(p)), 0) + 1
    print('kat_random.npz: %d candidates; (class,pred) histogram %s' % (n, sorted(hist.items())))


# ---------------------------------------------------------------------------------------------
# work-dir cases
# ---------------------------------------------------------------------------------------------

def fuzz_contigs(seed):
    return synth.fuzz_case(seed, n_contigs=2 + seed % 3)


FULL_CASES = [  # (name, seed, dialect, svlen_thres, supp_thres)
    ('fuzz_cutesv_s1', 1, 'cutesv', 50, 2), ('fuzz_cutesv_s2', 2, 'cutesv', 50, 2),
    ('fuzz_cutesv_s3', 3, 'cutesv', 30, 1), ('fuzz_cutesv_s4', 4, 'cutesv', 50, 3),
    ('fuzz_sniffles_s5', 5, 'sniffles', 50, 2), ('fuzz_sniffles_s6', 6, 'sniffles', 50, 2),
    ('fuzz_svim_s7', 7, 'svim', 50, 2), ('fuzz_svim_s8', 8, 'svim', 40, 2),
    ('fuzz_cutesv_s9', 9, 'cutesv', 50, 2), ('fuzz_sniffles_s10', 10, 'sniffles', 50, 5),
]


def build_case(home, name_or_seed, dialect, kind='fuzz'):
    if kind == 'fuzz':
        contigs = fuzz_contigs(name_or_seed)
    elif kind == 'chr21':
        contigs = [synth.bench_contig('21', 1500, 1500, name_or_seed, deg_lo=2, deg_hi=14)]
    elif kind == 'config2':
        contigs = [synth.bench_contig('1', 200000, 100000, name_or_seed)]
    elif kind == 'genome_small':
        contigs = synth.bench_genome(200000, name_or_seed)
    else:
        raise ValueError(kind)
    synth.write_workdir(home, contigs, dialect=dialect, seed=int(name_or_seed), write_bam=False)
    return contigs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--quick', action='store_true')
    args = ap.parse_args()
    if not os.path.isdir(REF_SRC):
        sys.exit('reference not present: this script only runs in the development container')
    sys.path.insert(0, REF_SRC)
    tmp = tempfile.mkdtemp(prefix='duet_golden_')
    install_shims(tmp)
    make_kat()
    make_kat_random()

    cases_dir = os.path.join(HERE, 'cases')
    if os.path.isdir(cases_dir):
        shutil.rmtree(cases_dir)
    os.makedirs(cases_dir)
    for name, seed, dialect, sl, sr in FULL_CASES:
        home = os.path.join(cases_dir, name)
        build_case(home, seed, dialect)
        run_reference(home, sl, sr)
        with open(os.path.join(home, 'params.json'), 'w') as f:
            json.dump(dict(seed=seed, dialect=dialect, svlen_thres=sl, suppread_thres=sr), f)
        for n in os.listdir(os.path.join(home, 'snp_phasing')):     # keep only the text; BAMs are regenerated
            if n.endswith('.bam'):
                os.remove(os.path.join(home, 'snp_phasing', n))
        nrows = sum(1 for l in open(os.path.join(home, 'phased_sv.vcf')) if not l.startswith('#'))
        print('case %s: %d rows' % (name, nrows))

    seeded = []
    plan = [('fuzz', s, d, 50, 2) for s in range(100, 160) for d in synth.DIALECTS]
    plan += [('chr21', 21, d, 50, 2) for d in synth.DIALECTS]
    plan += [('genome_small', 3, 'cutesv', 50, 2)]
    if not args.quick:
        plan += [('config2', 1, 'cutesv', 50, 2)]
    for kind, seed, dialect, sl, sr in plan:
        home = os.path.join(tmp, 'w_%s_%d_%s' % (kind, seed, dialect))
        build_case(home, seed, dialect, kind)
        out = run_reference(home, sl, sr)
        nrows = sum(1 for l in open(out) if not l.startswith('#'))
        seeded.append(dict(kind=kind, seed=seed, dialect=dialect, svlen_thres=sl, suppread_thres=sr,
                           inputs_sha256=inputs_digest(home), output_sha256=sha256_file(out), rows=nrows))
        print('seeded %s/%d/%s: %d rows' % (kind, seed, dialect, nrows))
        shutil.rmtree(home)
    with open(os.path.join(HERE, 'seeded.json'), 'w') as f:
        json.dump(seeded, f, indent=1)
    shutil.rmtree(tmp)


if __name__ == '__main__':
    main()
