# coding=utf-8
"""Pin the scalar C oracle (oracle/ef_oracle.c) and the host ingest (text/BAM -> SoA, rows -> text)
to the Python oracle / reference goldens. The C oracle stands in for the GPU HERE ONLY, as a checker
of the host logic; the product never calls it."""
import os
import shutil

import numpy as np
import pytest

from duet_amd import sv_phasing_fn as F
from duet_amd import write_file as W
from duet_amd import read_file as RF
from duet_amd import bamio, synth
from oracle import c_oracle
from oracle import ef_oracle as O
from tests import helpers as H


def host_path_with_checker(home, svlen_thres, suppread_thres):
    """Product host code end to end, with the C oracle in place of the HIP call."""
    vcf = home + '/sv_calling/variants.vcf'
    tokens = RF.read_file(vcf)
    head = W.header_text(tokens, RF.init_chrom_list(False, home), False)
    tab, soa = F.generate_callinfo(vcf, F.read_hap_bam(home + '/snp_phasing/', 4, False), False)
    rc, pred, ps = c_oracle.ef(soa, svlen_thres, suppread_thres)
    assert rc == 0
    rows = F.assemble_rows(tab, pred, ps, F.ps_classes(soa))
    return head + W.rows_text(rows), soa, pred, ps


def materialise_bams(home):
    """Fixture dirs keep only <x>.bam.sam; write the matching BAM files next to them."""
    d = os.path.join(home, 'snp_phasing')
    for n in sorted(os.listdir(d)):
        if n.endswith('.bam.sam'):
            with open(os.path.join(d, n)) as f:
                lines = f.read().split('\n')[:-1]
            bamio.write_bam_from_sam_lines(os.path.join(d, n[:-4]), [(n[:-8], 300000000)], lines)


@pytest.mark.parametrize('name,src,params', H.full_cases(), ids=[c[0] for c in H.full_cases()])
def test_full_cases_host_path(name, src, params, tmp_path):
    home = str(tmp_path / name)
    shutil.copytree(src, home)
    materialise_bams(home)
    with open(os.path.join(src, 'phased_sv.vcf')) as f:
        want = f.read()
    got, soa, pred, ps = host_path_with_checker(home, params['svlen_thres'], params['suppread_thres'])
    assert got == want
    # per-candidate trace against the Python oracle
    _, trace, callset = O.sv_phasing_text(home, params['svlen_thres'], params['suppread_thres'], want_trace=True)
    assert len(trace) == soa.n_cands
    for i, (kept, cls, opred, ops) in enumerate(trace):
        assert int(pred[i]) == (opred or 0), i
        assert int(ps[i]) == (ops if opred is not None else 0), i


def test_seeded_cases_host_path(tmp_path):
    n = 0
    for p in H.seeded_plan():
        if p['kind'] == 'config2' or (p['kind'] == 'fuzz' and p['seed'] % 4):
            continue
        home = str(tmp_path / ('%s_%d_%s' % (p['kind'], p['seed'], p['dialect'])))
        H.build_case(home, p['kind'], p['seed'], p['dialect'])
        got, _, _, _ = host_path_with_checker(home, p['svlen_thres'], p['suppread_thres'])
        assert H.sha256_bytes(got.encode()) == p['output_sha256'], p
        shutil.rmtree(home)
        n += 1
    assert n >= 40


def test_direct_soa_matches_text_path(tmp_path):
    """synth.soa_parts (used by bench.py to skip text) must describe the same problem as the files."""
    from duet_amd import engine
    contigs = H.case_contigs('chr21', 21)
    home = str(tmp_path / 'w')
    synth.write_workdir(home, contigs, dialect='cutesv', seed=21)
    _, soa_text, pred_t, ps_t = host_path_with_checker(home, 50, 2)
    # the text path lists all 24 default contigs; the direct path only the generated one
    soa_direct = engine.soa_from_synth(contigs)
    rc, pred_d, ps_d = c_oracle.ef(soa_direct, 50, 2)
    assert rc == 0
    assert soa_direct.n_cands == soa_text.n_cands and soa_direct.n_marks == soa_text.n_marks
    assert np.array_equal(pred_d, pred_t) and np.array_equal(ps_d, ps_t)


def test_config2_on_the_literal_8d_generator(tmp_path):
    """BASELINE configs[1] on SURVEY 8d's generator to the letter (20 % of the marks' names absent, pc = floor(Exp(600))):
    tests/golden/seeded_r5.json holds the unmodified reference's sha256 for it (make_golden_r5.py).  The regenerated inputs
    hash the same, and the C oracle's per-candidate results through the host formatter give the reference's bytes."""
    import json
    with open(os.path.join(H.GOLDEN, 'seeded_r5.json')) as f:
        p = json.load(f)[0]
    home = str(tmp_path / 'config2_8d')
    contigs = H.build_case(home, p['kind'], p['seed'], p['dialect'], write_sam=False)
    assert int(contigs[0].cand_off[-1]) == p['marks']
    from duet_amd import engine
    soa = engine.soa_from_synth(contigs)
    tagged = soa.read_tag != np.uint64(0xFFFFFFFFFFFFFFFF)
    absent = (soa.mark_read == engine.MARK_ABSENT).mean()
    # 20 % of the names have no SAM line at all, and a fifth of the others belong to reads without tags (the default generator:
    # 5 % + a fifth of the rest = 24 % -- which is the other way to read 8d's "20 % of names absent from the tagged subset")
    assert 0.35 < absent < 0.37
    pc = ((soa.read_tag[tagged] >> np.uint64(32)) & np.uint64(0x3FFFFFFF)).astype(np.int64)
    low = pc[pc <= 8100]
    assert 560 < low.mean() < 640 and (pc > 8100).mean() > 0.02
    got, soa2, _, _ = host_path_with_checker(home, p['svlen_thres'], p['suppread_thres'])
    assert soa2.n_marks == p['marks']
    assert sum(1 for l in got.splitlines() if not l.startswith('#')) == p['rows']
    assert H.sha256_bytes(got.encode()) == p['output_sha256']
