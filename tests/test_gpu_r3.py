# coding=utf-8
"""-m gpu, round 3: BASELINE configs[3] / [4] at their real SHAPE (24 contigs, 2e7 marks) in the dialects those
configurations produce, pinned to the unmodified reference by tests/golden/make_golden_r3.py; the sharded entries (step E/F and
the clustered SVIM mode) from a fresh interpreter with the real kernels."""
import json
import os
import shutil

import numpy as np
import pytest

from duet_amd import synth
from tests import helpers as H
from tests.test_gpu_r2 import fresh_interpreter, run_product

pytestmark = pytest.mark.gpu


def r3_plan():
    with open(os.path.join(H.GOLDEN, 'seeded_r3.json')) as f:
        return json.load(f)


@pytest.mark.parametrize('dialect', ['svim', 'sniffles'])
def test_config3_shape_in_the_callers_dialects(dialect, tmp_path):
    """configs[3] (`--sv_caller svim`: READS= / GT:DP:AD) and configs[4] (`--sv_caller sniffles`, min_support_read = 2:
    RNAMES= / GT:GQ:DR:DV, refread = GQ -- read_file.py:56-76) on the 24-contig, 2e7-mark genome: the whole product path
    (native ingest, E/F kernels, rows on the device) against the one reference run recorded in seeded_r3.json (998,750 rows),
    then the same work dir through gpus=2 (plumbing mode: both ranks on device 0, gloo)."""
    p = [x for x in r3_plan() if x['dialect'] == dialect][0]
    home = str(tmp_path / ('config3_' + dialect))
    synth.write_workdir(home, synth.bench_genome(20000000, p['seed']), dialect=dialect, seed=p['seed'], write_sam=False)
    got = run_product(home, p['svlen_thres'], p['suppread_thres'], python_path=False)
    assert sum(1 for l in got.split(b'\n') if l and not l.startswith(b'#')) == p['rows']
    assert H.sha256_bytes(got) == p['output_sha256']
    os.remove(home + '/phased_sv.vcf')
    r = fresh_interpreter('from duet_amd.sv_phasing import sv_phasing\nsv_phasing(%r, %d, %d, 8, False, gpus=2)\n' % (
        home, p['svlen_thres'], p['suppread_thres']), {'DUET_ONE_GPU': '1'}, timeout=1500)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    with open(home + '/phased_sv.vcf', 'rb') as f:
        assert H.sha256_bytes(f.read()) == p['output_sha256']
    shutil.rmtree(home)


@pytest.mark.parametrize('gpus', [2, 8])
def test_svim_gpu_mode_sharded_over_ranks(gpus, tmp_path):
    """`duet -b svim-gpu --gpus N` / svim_mode.sv_phasing_from_bams(..., gpus=N): every rank extracts its own contigs'
    signatures and runs the device pipeline (A0 -> E/F) on them; one all-gather of candidate records; rank 0 writes.
    Against the single-GPU run of this process (same kernels, whole genome at once) and its CPU twin."""
    from duet_amd import svim_mode
    from oracle import svim_oracle
    home = str(tmp_path / 'w')
    synth.write_svim_workdir(home, H.case_contigs('genome_small', 5), 5)
    svim_mode.sv_phasing_from_bams(home, 50, 2, 4, False, 0.9, 0)
    one = open(home + '/phased_sv.vcf').read()
    chroms = svim_mode.init_chrom_list(False, home)
    want = svim_oracle.phase_workdir(home, chroms, 50, 2, min_sv_size=50)
    assert one.endswith(svim_mode.rows_text(home, dict(want, chroms=chroms))) and one.count('Duet.') > 100
    os.remove(home + '/phased_sv.vcf')
    r = fresh_interpreter('from duet_amd import svim_mode\nsvim_mode.sv_phasing_from_bams(%r, 50, 2, 4, False, 0.9, 0, gpus=%d)\n' % (
        home, gpus), {'DUET_ONE_GPU': '1'})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert open(home + '/phased_sv.vcf').read() == one


def test_eight_ranks_with_the_real_kernels(tmp_path):
    """24 contigs over EIGHT ranks (all on device 0, gloo), SVIM dialect, -r 2, step E/F: what the first real 8-GPU run
    will do except for RCCL."""
    home = str(tmp_path / 'g')
    p = [x for x in H.seeded_r2_plan() if x['kind'] == 'genome_small' and x['dialect'] == 'svim'][0]
    H.build_case(home, 'genome_small', p['seed'], 'svim', write_sam=False)
    r = fresh_interpreter('from duet_amd.sv_phasing import sv_phasing\nsv_phasing(%r, %d, %d, 8, False, gpus=8)\n' % (
        home, p['svlen_thres'], p['suppread_thres']), {'DUET_ONE_GPU': '1'})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    with open(home + '/phased_sv.vcf', 'rb') as f:
        assert H.sha256_bytes(f.read()) == p['output_sha256']


def test_one_rank_over_rccl(tmp_path):
    """The collective backend a real multi-GPU run uses, on the one GPU of this box: DUET_FORCE_RANKS=1 takes
    `sv_phasing(..., gpus=1)` and the sharded SVIM mode through the one-process-per-GPU path with ONE rank and backend "nccl"
    (RCCL: communicator set-up on the device, all_gather_into_tensor on device memory beside the kernels' stream, barrier,
    tear-down) instead of the gloo plumbing mode of the other sharded tests."""
    home = str(tmp_path / 'g')
    p = [x for x in H.seeded_r2_plan() if x['kind'] == 'genome_small' and x['dialect'] == 'sniffles'][0]
    H.build_case(home, 'genome_small', p['seed'], 'sniffles', write_sam=False)
    r = fresh_interpreter('from duet_amd.sv_phasing import sv_phasing\nsv_phasing(%r, %d, %d, 4, False, gpus=1)\n' % (
        home, p['svlen_thres'], p['suppread_thres']), {'DUET_FORCE_RANKS': '1', 'NCCL_DEBUG': 'VERSION'})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b'RCCL version' in r.stdout + r.stderr            # (the library announces itself when its communicator is set up)
    with open(home + '/phased_sv.vcf', 'rb') as f:
        assert H.sha256_bytes(f.read()) == p['output_sha256']
    from duet_amd import svim_mode
    home2 = str(tmp_path / 'w')
    synth.write_svim_workdir(home2, H.case_contigs('genome_small', 5), 5)
    svim_mode.sv_phasing_from_bams(home2, 50, 2, 4, False, 0.9, 0)
    one = open(home2 + '/phased_sv.vcf').read()
    os.remove(home2 + '/phased_sv.vcf')
    r = fresh_interpreter('from duet_amd import svim_mode\nsvim_mode.sv_phasing_from_bams(%r, 50, 2, 4, False, 0.9, 0, gpus=1)\n' % home2,
                          {'DUET_FORCE_RANKS': '1'})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert open(home2 + '/phased_sv.vcf').read() == one


@pytest.mark.parametrize('collective', ['duet', 'torch'])
def test_bench_sharded_path_as_one_rank_over_rccl(collective):
    """bench.py's N > 1 code path -- LPT sharding, the asynchronous all-gather per problem on its own stream beside the kernels'
    raw stream, barriers, the reduced per-rank figures, the fused sharded extra -- as ONE rank over real RCCL
    (DUET_BENCH_RCCL_SELF=1; two ranks cannot share this box's GPU under RCCL): through the collective the product ships
    (duet_comm_* inside libduet_ef.so: the default) and through torch.distributed "nccl" (`--collective torch`)."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    env['DUET_BENCH_RCCL_SELF'] = '1'
    out = subprocess.check_output([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '1', '--steps', '5', '--warmup', '2',
                                   '--genome-marks', '2000000', '--collective', collective], env=env, timeout=900).decode()
    line = [l for l in out.splitlines() if l.startswith('{')][-1]
    assert len(line) < 8000
    d = json.loads(line)
    assert d['n_gpus'] == 1 and d['parity_vs_oracle'] is True
    assert d['topology']['backend'] == ('duet_comm' if collective == 'duet' else 'nccl'), d['topology']
    assert d['topology']['world_size'] == 1 and d['topology']['rccl_version']
    assert ('duet_comm_*' in d['collective']) == (collective == 'duet') and 'unavailable' not in d['collective']
    assert d['gather']['collectives_per_problem'] == 1
    assert d['extra']['fused_clustered_and_phased_sharded']['parity_rank0_vs_composed_oracles'] is True


def test_in_library_collective_one_rank():
    """duet_comm_* directly (round 4: the collective of the sharded product path lives in libduet_ef.so): RCCL loaded at run
    time, a communicator of ONE rank created from its own unique id on this box's GPU, the all-gather staged through device
    buffers -- what every rank of a real multi-GPU run does with world > 1 (two ranks cannot share one GPU under RCCL)."""
    r = fresh_interpreter(
        'import os\nos.environ["DUET_NO_TORCH"] = "1"\nimport numpy as np\nfrom duet_amd import _lib, comm\n'
        'ctx = _lib.Context(0)\nstar = comm.TcpStar(0, 1)\ng = comm.RcclGather(ctx, star)\n'
        'blk = (np.arange(100003) % 251).astype(np.uint8)\nout = g.allgather(blk)\n'
        'assert out.shape == (1, 100003) and np.array_equal(out[0], blk)\n'
        'out = g.allgather(blk[:17])\nassert np.array_equal(out[0], blk[:17])\ng.close()\nctx.close()\nprint("COMM OK")\n',
        {'NCCL_DEBUG': 'VERSION'})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b'COMM OK' in r.stdout and b'RCCL version' in r.stdout + r.stderr
