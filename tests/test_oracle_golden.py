# coding=utf-8
"""Pin the CPU oracle (oracle/ef_oracle.py) to outputs captured from the imported reference."""
import json
import os

import numpy as np
import pytest

from oracle import ef_oracle as O
from tests import helpers as H


def _kat_candidate(spec, pos, svread, refread):
    cd = O.Candidate()
    cd.pos, cd.svread, cd.refread = pos, svread, refread
    marks = []
    for part in [p.strip() for p in spec.split(',') if p.strip()]:
        rep = 1
        if '*' in part:
            part, r = part.split('*')
            rep = int(r)
        for _ in range(rep):
            if part == 'u':
                marks.append(None)
            else:
                hap, rest = part.split('@')
                ps, pc = rest.split(':')
                marks.append((int(hap), int(ps), int(pc)))
    cd.marks = marks
    return cd


def test_known_answer_table(golden_dir):
    with open(os.path.join(golden_dir, 'kat_predict_hp.json')) as f:
        kat = json.load(f)
    seeds = set(kat['oneps'])
    assert len(kat['rows']) == 38
    for i, r in enumerate(kat['rows']):
        cd = _kat_candidate(r['marks'], r['pos'], r['svread'], r['refread'])
        assert O.ps_class(cd) == r['cls'], i
        assert O.decide(cd, r['cls'], seeds) == (r['pred'], r['ps']), (i, r)


def test_random_known_answers(golden_dir):
    with np.load(os.path.join(golden_dir, 'kat_random.npz')) as zf:
        z = {k: zf[k] for k in zf.files}
    off = z['off']
    sets = [set(int(x) for x in z['oneps_val'][z['oneps_off'][s]:z['oneps_off'][s + 1]])
            for s in range(len(z['oneps_off']) - 1)]
    bad = 0
    for i in range(len(z['cls'])):
        cd = O.Candidate()
        cd.pos, cd.svread, cd.refread = int(z['pos'][i]), int(z['svread'][i]), int(z['refread'][i])
        a, b = off[i], off[i + 1]
        cd.marks = [(int(h), int(p), int(c)) if t else None
                    for t, h, p, c in zip(z['m_tagged'][a:b], z['m_hap'][a:b], z['m_ps'][a:b], z['m_pc'][a:b])]
        assert O.ps_class(cd) == int(z['cls'][i])
        got = O.decide(cd, int(z['cls'][i]), sets[int(z['oneps_set'][i])])
        bad += got != (int(z['pred'][i]), int(z['ps'][i]))
    assert bad == 0


@pytest.mark.parametrize('name,home,params', H.full_cases(), ids=[c[0] for c in H.full_cases()])
def test_full_cases_bytes(name, home, params):
    with open(os.path.join(home, 'phased_sv.vcf')) as f:
        want = f.read()
    got = O.sv_phasing_text(home, params['svlen_thres'], params['suppread_thres'])
    assert got == want


def test_seeded_cases_sha(tmp_path):
    plan = [p for p in H.seeded_plan() if p['kind'] != 'config2']
    assert len(plan) >= 100
    for p in plan:
        home = str(tmp_path / ('%s_%d_%s' % (p['kind'], p['seed'], p['dialect'])))
        H.build_case(home, p['kind'], p['seed'], p['dialect'], write_bam=False)
        assert H.inputs_digest(home) == p['inputs_sha256'], 'generator drifted: ' + str(p)
        got = O.sv_phasing_text(home, p['svlen_thres'], p['suppread_thres'])
        assert H.sha256_bytes(got.encode()) == p['output_sha256'], p
        assert sum(1 for l in got.splitlines() if not l.startswith('#')) == p['rows']
