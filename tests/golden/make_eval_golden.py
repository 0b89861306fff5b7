#!/usr/bin/env python3
# coding=utf-8
"""Golden vectors for duet_amd/evaluation.py from the UNMODIFIED reference evaluator
(/root/reference/src/scripts/evaluation.py, imported here only).  Inputs: small phased callsets produced by
the oracle from seeded synthetic work dirs, and truth sets derived from them by a seeded perturbation."""
import importlib.util
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from duet_amd import synth          # noqa: E402
from oracle import ef_oracle        # noqa: E402


def make_pair(seed, out_dir):
    rng = synth.SplitMix(0xE7A10000 + seed)
    home = tempfile.mkdtemp(prefix='duet_eval_')
    contigs = [synth.bench_contig(l, 700, 350, seed * 7 + i, deg_lo=2, deg_hi=12, length=3000000)
               for i, l in enumerate(('1', '2', 'X'))]
    synth.write_workdir(home, contigs, dialect='cutesv', seed=seed, write_bam=False)
    call_text = ef_oracle.sv_phasing_text(home, 50, 2)
    rows = [l.split('\t') for l in call_text.splitlines() if not l.startswith('#')]
    truth = ['##fileformat=VCFv4.2', '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE']
    flips = {'1|0': '0|1', '0|1': '1|0', '1|1': '1|1'}
    for i, r in enumerate(rows):
        u = rng.one(100)
        if u < 12:
            continue                                            # missing from the truth
        pos = int(r[1]) + rng.one(2401) - 1200                  # some beyond --refdist
        svlen = int(r[7].split(';')[0][6:])
        if rng.one(5) == 0:
            svlen = int(svlen * (0.4 + rng.one(120) / 100.0)) or 60
        typ = r[7].split('SVTYPE=<')[1][:-1]
        hp, ps = r[9].split(':')
        v = rng.one(10)
        if v == 0:
            hp = flips[hp]
        elif v == 1:
            hp = '1|1' if hp != '1|1' else '1|0'
        gt = hp if rng.one(7) else hp.replace('|', '/')
        alt = '<%s>' % typ if rng.one(3) else ('ACGT' * (abs(svlen) // 4 + 1) if typ == 'INS' else 'A')
        ref = 'N' if alt.startswith('<') or typ == 'INS' else 'ACGT' * (abs(svlen) // 4 + 1)
        info = 'SVTYPE=%s;SVLEN=%d' % (typ, svlen) if alt.startswith('<') else 'SVTYPE=%s' % typ
        if alt.startswith('<') is False and typ == 'INS':
            ref = 'A'
        truth.append('\t'.join([r[0], str(max(pos, 1)), 't%d' % i, ref, alt, '.', 'PASS', info, 'GT:PS',
                                '%s:%s' % (gt, ps)]))
    for j in range(25):                                          # false negatives for the callset
        c = ('chr1', 'chr2', 'chrX')[rng.one(3)]
        truth.append('\t'.join([c, str(1 + rng.one(2900000)), 'x%d' % j, 'N', '<DEL>', '.', 'PASS',
                                'SVTYPE=DEL;SVLEN=-%d' % (60 + rng.one(900)), 'GT:PS', '0|1:%d' % (1 + rng.one(2000000))]))
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, 'call.vcf'), 'w') as f:
        f.write(call_text)
    with open(os.path.join(out_dir, 'truth.vcf'), 'w') as f:
        f.write('\n'.join(truth) + '\n')
    with open(os.path.join(out_dir, 'regions.bed'), 'w') as f:
        for c in ('chr1', 'chr2', 'chrX'):
            f.write('%s\t200000\t1400000\n%s\t1800000\t2600000\n' % (c, c))


def main():
    spec = importlib.util.spec_from_file_location('ref_eval', '/root/reference/src/scripts/evaluation.py')
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = []
    for seed in (1, 2, 3):
        d = os.path.join(HERE, 'eval', 'pair%d' % seed)
        make_pair(seed, d)
        for skip in (False, True):
            for bed in ('', os.path.join(d, 'regions.bed')):
                for refdist, pct in ((1000, 0.0), (500, 0.7)):
                    res = ref.evaluation(ref.parse_vcf(os.path.join(d, 'truth.vcf'), skip, bed),
                                         ref.parse_vcf(os.path.join(d, 'call.vcf'), skip, bed), refdist, pct)
                    out.append(dict(pair=seed, skip_phasing=skip, bed=bool(bed), refdist=refdist, pctsim=pct,
                                    result=[float(x) for x in res]))
    with open(os.path.join(HERE, 'eval', 'expected.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print(len(out), 'evaluations;', out[0]['result'])


if __name__ == '__main__':
    main()
