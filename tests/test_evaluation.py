# coding=utf-8
"""duet_amd/evaluation.py against numbers captured from the reference evaluator (tests/golden/make_eval_golden.py)."""
import json
import os

from duet_amd import evaluation as E
from tests import helpers as H


def test_matches_reference_numbers():
    d = os.path.join(H.GOLDEN, 'eval')
    with open(os.path.join(d, 'expected.json')) as f:
        cases = json.load(f)
    assert len(cases) == 24
    for c in cases:
        pair = os.path.join(d, 'pair%d' % c['pair'])
        bed = os.path.join(pair, 'regions.bed') if c['bed'] else ''
        got = E.evaluation(E.parse_vcf(os.path.join(pair, 'truth.vcf'), c['skip_phasing'], bed),
                           E.parse_vcf(os.path.join(pair, 'call.vcf'), c['skip_phasing'], bed), c['refdist'], c['pctsim'])
        assert [float(x) for x in got] == c['result'], c          # same sets, same float divisions -> exact


def test_cli_prints_like_upstream(capsys):
    pair = os.path.join(H.GOLDEN, 'eval', 'pair1')
    E.main([os.path.join(pair, 'call.vcf'), os.path.join(pair, 'truth.vcf')])
    out = capsys.readouterr().out.splitlines()
    assert out[0].startswith('Average SV number per phase set is ')
    assert out[1].startswith('The precision, recall and F1 score of SV calling are ')
    assert out[3].startswith('The precision, recall and F1 score of SV phasing are ')
