# coding=utf-8
"""-m gpu, round 2: the reference pins added by tests/golden/make_golden_r2.py through the HIP path, the N-GPU product
entry with the real kernels, and the small boundary checks (device selection, the integration stub's own _run)."""
import logging
import os
import shutil

import pytest

from duet_amd import _lib, engine, multi
from duet_amd.sv_phasing import sv_phasing
from tests import helpers as H
from tests.test_c_oracle import materialise_bams

pytestmark = pytest.mark.gpu


def run_product(home, svlen_thres, suppread_thres, all_ctgs=False, python_path=None, **kw):
    """sv_phasing through the native host path, the Python host path, or both (must agree); all end in the HIP kernels."""
    outs = []
    for force_py in ((False, True) if python_path is None else (python_path,)):
        old = os.environ.get('DUET_NATIVE_INGEST')
        os.environ['DUET_NATIVE_INGEST'] = '0' if force_py else '1'
        try:
            sv_phasing(home, svlen_thres, suppread_thres, 4, all_ctgs, **kw)
        finally:
            if old is None:
                del os.environ['DUET_NATIVE_INGEST']
            else:
                os.environ['DUET_NATIVE_INGEST'] = old
        with open(os.path.join(home, 'phased_sv.vcf'), 'rb') as f:
            outs.append(f.read())
    assert all(o == outs[0] for o in outs)
    return outs[0]


A_CASES = H.all_ctgs_cases()


@pytest.mark.parametrize('name,src,params', A_CASES, ids=[c[0] for c in A_CASES])
def test_all_ctgs_cases_bytes(name, src, params, tmp_path, monkeypatch):
    """-a / --include_all_ctgs (read_file.py:13-15, write_file.py:38-41) byte for byte against the reference."""
    H.install_tabix_shim(tmp_path, monkeypatch)
    home = str(tmp_path / name)
    shutil.copytree(src, home)
    os.remove(os.path.join(home, 'phased_sv.vcf'))
    materialise_bams(home)
    with open(os.path.join(src, 'phased_sv.vcf'), 'rb') as f:
        want = f.read()
    assert run_product(home, params['svlen_thres'], params['suppread_thres'], all_ctgs=True) == want


def test_all_ctgs_seeded_sha(tmp_path, monkeypatch):
    H.install_tabix_shim(tmp_path, monkeypatch)
    n = 0
    for p in H.seeded_r2_plan():
        if p['kind'] != 'fuzz_a':
            continue
        home = str(tmp_path / ('a_%d_%s' % (p['seed'], p['dialect'])))
        H.build_case(home, 'fuzz_a', p['seed'], p['dialect'], write_sam=False)
        got = run_product(home, p['svlen_thres'], p['suppread_thres'], all_ctgs=True, python_path=bool(n % 2))
        assert H.sha256_bytes(got) == p['output_sha256'], p
        shutil.rmtree(home)
        n += 1
    assert n == 120


def test_genome_small_svim_and_sniffles_sha(tmp_path):
    """Stand-ins for BASELINE configs[3] / [4] (the real ONT genomes are not obtainable here): 24 contigs in the SVIM and
    Sniffles dialects, through both host paths."""
    for p in H.seeded_r2_plan():
        if p['kind'] != 'genome_small':
            continue
        home = str(tmp_path / ('gs_' + p['dialect']))
        H.build_case(home, 'genome_small', p['seed'], p['dialect'], write_sam=False)
        got = run_product(home, p['svlen_thres'], p['suppread_thres'])
        assert sum(1 for l in got.split(b'\n') if l and not l.startswith(b'#')) == p['rows']
        assert H.sha256_bytes(got) == p['output_sha256'], p['dialect']
        shutil.rmtree(home)


def test_config3_text_sha(tmp_path):
    """BASELINE configs[2] as TEXT: 24 contigs, 2e7 marks, 2e6 candidates (564 MB caller VCF + 24 haplotagged BAMs) ->
    phased_sv.vcf, byte-identical to the one reference run recorded in seeded_r2.json (998,750 rows)."""
    p = [x for x in H.seeded_r2_plan() if x['kind'] == 'config3'][0]
    home = str(tmp_path / 'config3')
    from duet_amd import synth
    synth.write_workdir(home, synth.bench_genome(20000000, p['seed']), dialect=p['dialect'], seed=p['seed'], write_sam=False)
    got = run_product(home, p['svlen_thres'], p['suppread_thres'], python_path=False)
    assert H.sha256_bytes(got) == p['output_sha256']
    shutil.rmtree(home)


def _divzero_home(tmp_path):
    import sys
    sys.path.insert(0, os.path.join(H.GOLDEN))
    import make_golden_r2 as R
    home = str(tmp_path / 'dz')
    R.divzero_case(home)
    os.remove(home + '/snp_phasing/chr1.bam')
    materialise_bams(home)
    return home


@pytest.mark.parametrize('python_path', [False, True])
def test_division_by_zero_raises_and_leaves_the_header(tmp_path, python_path):
    """-r 0 with a kept candidate whose svread + refread == 0 and which is its contig's only seed source: upstream adds
    its seed (:198-203), then raises at :123, after the header has been written (sv_phasing.py:16)."""
    p = [x for x in H.seeded_r2_plan() if x['kind'] == 'divzero'][0]
    assert p['raised'] == 'ZeroDivisionError'
    home = _divzero_home(tmp_path)
    with pytest.raises(ZeroDivisionError):
        run_product(home, 50, 0, python_path=python_path)
    with open(home + '/phased_sv.vcf') as f:
        assert f.read() == p['file_left_behind']


def test_device_argument_reaches_the_context(tmp_path):
    """--device / sv_phasing(device=N): the context that runs E/F sits on device N (a box with one GPU cannot run on
    device 1 -- if the argument were ignored, this call would quietly succeed on device 0)."""
    import torch
    name, src, params = H.full_cases()[0]
    home = str(tmp_path / name)
    shutil.copytree(src, home)
    materialise_bams(home)
    sv_phasing(home, params['svlen_thres'], params['suppread_thres'], 4, False, device=0)
    assert engine.default_context(0).device_id == 0
    n = torch.cuda.device_count()
    with pytest.raises(_lib.DuetLibraryError):
        sv_phasing(home, params['svlen_thres'], params['suppread_thres'], 4, False, device=n)


def fresh_interpreter(code, env_extra=None, timeout=900):
    """Run `code` in a NEW python process (the sharded entries start their ranks from a parent that holds no GPU:
    duet_amd/launch.py refuses anything else, and this pytest process holds one).  -> CompletedProcess"""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env['PYTHONPATH'] = os.pathsep.join([repo] + ([env['PYTHONPATH']] if env.get('PYTHONPATH') else []))
    env.update(env_extra or {})
    return subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


@pytest.mark.parametrize('gpus', [2, 3])
def test_sharded_product_entry_with_the_real_kernels(tmp_path, gpus):
    """sv_phasing(..., gpus=N) from a fresh interpreter: one process per rank, contigs LPT-sharded on the caller VCF's line
    bytes, every rank reads its own contigs only, the three kernels per rank, ONE all-gather, every rank formats its rows, the
    parent assembles the file.  This box has one GPU, so the ranks share device 0 and the collective goes through gloo
    (DUET_ONE_GPU=1: plumbing mode).  cuteSV, Sniffles AND SVIM dialects."""
    for name, src, params in H.full_cases()[:4] + H.full_cases()[6:]:
        home = str(tmp_path / name)
        shutil.copytree(src, home)
        os.remove(os.path.join(home, 'phased_sv.vcf'))
        materialise_bams(home)
        r = fresh_interpreter('from duet_amd.sv_phasing import sv_phasing\nsv_phasing(%r, %d, %d, 4, False, gpus=%d)\n' % (
            home, params['svlen_thres'], params['suppread_thres'], gpus), {'DUET_ONE_GPU': '1'})
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        with open(os.path.join(src, 'phased_sv.vcf'), 'rb') as f:
            want = f.read()
        with open(os.path.join(home, 'phased_sv.vcf'), 'rb') as f:
            assert f.read() == want, name
        assert not [n for n in os.listdir(home) if 'part' in n]


def test_sharded_entry_refuses_a_gpu_holding_parent(tmp_path):
    name, src, params = H.full_cases()[0]
    home = str(tmp_path / name)
    shutil.copytree(src, home)
    materialise_bams(home)
    engine.default_context(0)
    with pytest.raises(RuntimeError, match='fresh interpreter'):
        sv_phasing(home, params['svlen_thres'], params['suppread_thres'], 4, False, gpus=2)


def test_sharded_division_by_zero(tmp_path):
    home = _divzero_home(tmp_path)
    r = fresh_interpreter('from duet_amd import multi\ntry:\n    multi.sv_phasing_sharded(%r, 50, 0, 4, False, 2)\n'
                          'except ZeroDivisionError:\n    print("ZERODIV")\n' % home, {'DUET_ONE_GPU': '1'})
    assert r.returncode == 0 and b'ZERODIV' in r.stdout, r.stderr.decode()[-2000:]
    p = [x for x in H.seeded_r2_plan() if x['kind'] == 'divzero'][0]
    with open(home + '/phased_sv.vcf') as f:
        assert f.read() == p['file_left_behind']


def test_log_lines_follow_upstream(tmp_path, caplog):
    """The native path logs what upstream logs, in upstream's order (sv_phasing.py:10-20, sv_phasing_fn.py:12,30-45)."""
    name, src, params = H.full_cases()[0]
    home = str(tmp_path / name)
    shutil.copytree(src, home)
    materialise_bams(home)
    with caplog.at_level(logging.INFO):
        sv_phasing(home, params['svlen_thres'], params['suppread_thres'], 4, False)
    msgs = [r.getMessage() for r in caplog.records]
    want_order = ['create output .vcf file', 'extract SNP signatures', 'extract SV signatures',
                  'integrate read weight information', 'calculate read weight statistics',
                  'predict SV haplotypes in the callset', 'write phased callset into .vcf file']
    at = [msgs.index(m) for m in want_order]
    assert at == sorted(at)
    sv_lines = msgs[msgs.index('extract SV signatures') + 1:msgs.index('integrate read weight information')]
    assert len(sv_lines) == 24 and all(l.startswith('  signatures extracted from ') or l.startswith('  no signature from ') for l in sv_lines)
    bam_lines = msgs[msgs.index('extract SNP signatures') + 1:msgs.index('extract SV signatures')]
    n_bams = sum(1 for n in os.listdir(home + '/snp_phasing') if n.endswith('.bam'))
    assert len(bam_lines) == n_bams


def test_integration_stub_with_its_own_run(tmp_path):
    """integration/ef_gpu.py exactly as INTEGRATION.md tells a maintainer to add it: its own ctypes _run, the real library."""
    from tests.test_integration_stub import load_stub, stub_rows_match_goldens
    os.environ['DUET_EF_LIB'] = _lib.LIB_PATH
    try:
        stub_rows_match_goldens(load_stub(real_run=True), tmp_path, H.full_cases())
    finally:
        del os.environ['DUET_EF_LIB']


def test_bench_self_spawns_and_runs_the_sharded_genome(tmp_path):
    """`python bench.py --gpus 2` started plainly: the parent spawns one process per rank before touching the GPU; the ranks
    run the LPT-sharded whole-genome problem with ONE all-gather per problem and rank 0 prints one JSON line whose merged
    result matches the C oracle.  One GPU here: DUET_BENCH_ONE_GPU=1 (ranks share device 0, gloo) -- plumbing, not a measurement."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env['DUET_BENCH_ONE_GPU'] = '1'
    out = subprocess.check_output([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                                   '--genome-marks', '400000', '--no-extra'], env=env, timeout=600).decode()
    line = [l for l in out.splitlines() if l.startswith('{')][-1]
    assert len(line) < 8000                          # (the driver could not parse round 4's 22.5 kB line)
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['steps'] == 3 and d['parity_vs_oracle'] is True
    assert d['gather']['collectives_per_problem'] == 1 and d['topology']['world_size'] == 2 and len(d['per_rank']['marks']) == 2
    assert sum(d['per_rank']['marks']) == d['config']['marks']
    # ... and the complete record beside it (per rank: contigs, kernel times; topology with device names)
    with open(os.path.join(repo, d['detail_file'])) as f:
        full = json.loads(f.readline())
    assert len(full['per_rank']) == 2 and sum(r['marks'] for r in full['per_rank']) == d['config']['marks']
    assert d['same_problem_on_1_gpu']['parity_vs_oracle'] is True
    assert 1.0 <= d['sharding']['lpt_imbalance_max_over_mean_marks'] < 1.2
