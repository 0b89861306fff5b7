# coding=utf-8
"""-m gpu: the accuracy evaluator on the device (duet_eval_run_host) against the numbers captured from the reference's own
evaluator (tests/golden/eval/expected.json, made by tests/golden/make_eval_golden.py) and against the numpy restatement on
larger synthetic callsets (duplicate ids, several phase sets per contig, ties between the two labellings)."""
import json
import os

import numpy as np
import pytest

from duet_amd import evaluation as E
from duet_amd import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_reference_numbers():
    d = os.path.join(H.GOLDEN, 'eval')
    with open(os.path.join(d, 'expected.json')) as f:
        cases = json.load(f)
    assert len(cases) == 24
    for c in cases:
        pair = os.path.join(d, 'pair%d' % c['pair'])
        bed = os.path.join(pair, 'regions.bed') if c['bed'] else ''
        truth = E.parse_vcf(os.path.join(pair, 'truth.vcf'), c['skip_phasing'], bed)
        calls = E.parse_vcf(os.path.join(pair, 'call.vcf'), c['skip_phasing'], bed)
        got = E.evaluation_gpu(truth, calls, c['refdist'], c['pctsim'])
        assert [float(x) for x in got] == c['result'], c          # same sets, same float divisions -> exact


def synthetic_records(seed, n_truth, n_calls):
    rng = synth.SplitMix(0xE7A10000 + seed)
    hps = ['1|0', '0|1', '1|1', '1|2']

    def recs(n, tag, jitter):
        out = []
        chrom = rng.below(n, 26)
        pos = rng.between(n, 1, 200000)
        ln = rng.between(n, 50, 400)
        ty = rng.below(n, 5)
        hp = rng.below(n, 16)
        ps = rng.below(n, 6)
        dup = rng.below(n, 12)
        for i in range(n):
            c = 'chr%s' % (E.LABELS[int(chrom[i])] if chrom[i] < 24 else ('M', 'Un')[int(chrom[i]) - 24])
            p = int(pos[i]) // 37 * 37 + (jitter if i % 3 == 0 else 0)          # many exact position ties
            rid = '%s%d' % (tag, i if dup[i] else max(i - 1, 0))                # some id strings occur twice
            out.append({'chr': c, 'pos': p, 'id': rid + c + str(p), 'hp': hps[int(hp[i]) % 3 if hp[i] < 15 else 3],
                        'ps': '%s_:%d' % (c, int(ps[i]) * 50000), 'len': int(ln[i]),
                        'type': ('INS', 'DEL', 'INS', 'DEL', 'INV')[int(ty[i])]})
        return out

    truth = recs(n_truth, 't', 0)
    # every listed contig needs at least one truth record of each type, else upstream raises
    for k, c in enumerate(E.CHROMS):
        for t in ('INS', 'DEL'):
            truth.append({'chr': c, 'pos': 100 + k, 'id': 'fill%s%s' % (c, t), 'hp': '1|1', 'ps': c, 'len': 60, 'type': t})
    return truth, recs(n_calls, 'c', 5)


@pytest.mark.parametrize('seed,n_truth,n_calls,refdist,ratio', [(1, 300, 400, 1000, 0.0), (2, 3000, 5000, 100, 0.7),
                                                                 (3, 50000, 80000, 40, 0.5), (4, 2000, 2500, 3, 0.9)])
def test_matches_the_numpy_restatement(seed, n_truth, n_calls, refdist, ratio):
    truth, calls = synthetic_records(seed, n_truth, n_calls)
    want = E.evaluation(truth, calls, refdist, ratio)
    got = E.evaluation_gpu(truth, calls, refdist, ratio)
    assert [float(x) for x in got] == [float(x) for x in want]


def test_empty_truth_list_raises_like_upstream():
    truth, calls = synthetic_records(5, 50, 50)
    truth = [r for r in truth if not (r['chr'] == 'chr1' and r['type'] == 'INS')]
    calls.append({'chr': 'chr1', 'pos': 5, 'id': 'x', 'hp': '1|1', 'ps': 'chr1', 'len': 60, 'type': 'INS'})
    with pytest.raises(IndexError):
        E.evaluation(truth, calls, 1000, 0.0)
    with pytest.raises(IndexError):
        E.evaluation_gpu(truth, calls, 1000, 0.0)
