# coding=utf-8
"""The N-GPU product path (duet_amd/multi.py, duet_amd/launch.py) on CPU: world_size 2 over gloo.

What runs here is everything of `sv_phasing(..., gpus=2)` except the three HIP kernels: native ingest on every rank,
LPT contig sharding, ONE all-gather of the record blocks, merge into callset order, header first / rows appended.
The per-rank compute is played by the C oracle (tests only -- the product's rank entry, multi.rank_main, has no such
switch and fails without libduet_ef.so); the `-m gpu` twin in tests/test_gpu_multi.py runs the real kernels."""
import os
import shutil
import subprocess
import sys
import time

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from duet_amd import launch, multi
from oracle import c_oracle
from tests import helpers as H
from tests.test_c_oracle import materialise_bams


def oracle_compute(sub, svlen_thres, suppread_thres, n_max):
    rc, pred, ps = c_oracle.ef(sub, svlen_thres, suppread_thres)
    return multi.block_from_arrays(pred, ps, n_max, sub.n_contigs), (multi.RC_DIV_ZERO if rc == -5 else 0)


def _worker(rank, world, port, home, svlen_thres, suppread_thres, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        rc = multi.rank_body(home, svlen_thres, suppread_thres, 4, False, rank, world, oracle_compute, 'gloo')
        with open(os.path.join(out_dir, 'rc%d' % rank), 'w') as f:
            f.write(str(rc))
    finally:
        dist.destroy_process_group()


CASES = H.full_cases()


@pytest.mark.parametrize('name,src,params', CASES[::3], ids=[c[0] for c in CASES[::3]])
@pytest.mark.parametrize('world', [2, 3])
def test_sharded_ranks_write_the_golden_bytes(name, src, params, world, tmp_path):
    home = str(tmp_path / name)
    shutil.copytree(src, home)
    os.remove(os.path.join(home, 'phased_sv.vcf'))
    materialise_bams(home)
    mp.spawn(_worker, args=(world, launch.free_port(), home, params['svlen_thres'], params['suppread_thres'], str(tmp_path)),
             nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), 'rc%d' % r)).read() == '0'
    multi.assemble(home, False, world)                  # (what the parent does once every rank has exited with 0)
    with open(os.path.join(src, 'phased_sv.vcf'), 'rb') as f:
        want = f.read()
    with open(os.path.join(home, 'phased_sv.vcf'), 'rb') as f:
        assert f.read() == want
    assert not [n for n in os.listdir(home) if 'part' in n]


@pytest.mark.parametrize('dialect', ['svim', 'sniffles'])
def test_eight_ranks_24_contigs(dialect, tmp_path):
    """BASELINE configs[3] / [4] in shape: the 24 hg19 contigs over EIGHT ranks, SVIM (READS= / GT:DP:AD) and Sniffles
    (GT:GQ:DR:DV, refread = GQ) dialects, -r 2: every rank reads only its contigs' BAMs and records, one all-gather, every
    rank formats and numbers its own rows; the assembled file is the single-process oracle's, byte for byte."""
    from duet_amd import synth
    from oracle import ef_oracle
    home = str(tmp_path / 'g')
    contigs = synth.bench_genome(60000, 7)
    synth.write_workdir(home, contigs, dialect=dialect, seed=7, write_sam=True)
    want = ef_oracle.sv_phasing_text(home, 50, 2)
    mp.spawn(_worker, args=(8, launch.free_port(), home, 50, 2, str(tmp_path)), nprocs=8, join=True)
    for r in range(8):
        assert open(os.path.join(str(tmp_path), 'rc%d' % r)).read() == '0'
    multi.assemble(home, False, 8)
    got = open(os.path.join(home, 'phased_sv.vcf')).read()
    assert got == want
    assert got.count('\nchr') > 1000


def test_both_spellings_of_a_contig_number_across_ranks(tmp_path):
    """Records spelled chr1 AND 1 (read_file.py:30 accepts both; Q23): two CHROM texts of one contig, whose blocks sit apart
    in the file (text order: 1 < 2 < chr1 < chr2) with another rank's rows between them."""
    from duet_amd import bamio
    from oracle import ef_oracle
    home = str(tmp_path / 'w')
    os.makedirs(home + '/sv_calling')
    os.makedirs(home + '/snp_phasing')
    rec = '%s\t%d\tid\tN\t<DEL>\t.\tPASS\tPRECISE;SVTYPE=DEL;SVLEN=-80;END=180;RE=5;RNAMES=a,b,c,d;STRAND=+-\tGT:DR:DV:PL:GQ\t0/1:0:5:1,2,3:9'
    lines = ['##contig=<ID=chr1,length=1000000>', '##contig=<ID=chr2,length=1000000>']
    for i in range(40):
        lines.append(rec % (('chr1', '1', 'chr2', '2')[i % 4], 100 + 37 * i))
    with open(home + '/sv_calling/variants.vcf', 'w') as f:
        f.write('\n'.join(lines) + '\n')
    sam = ['%s\t0\t%s\t90\t60\t*\t*\t0\t0\t*\t*\tNM:i:1\tHP:i:%d\tPC:i:100\tPS:i:50' % (n, '%s', 1 + i % 2) for i, n in enumerate('abcd')]
    for c in ('chr1', 'chr2'):
        bamio.write_bam_from_sam_lines(home + '/snp_phasing/%s.bam' % c, [(c, 1000000)], [l % c for l in sam])
        with open(home + '/snp_phasing/%s.bam.sam' % c, 'w') as f:
            f.write('\n'.join(l % c for l in sam) + '\n')
    want = ef_oracle.sv_phasing_text(home, 50, 2)
    mp.spawn(_worker, args=(2, launch.free_port(), home, 50, 2, str(tmp_path)), nprocs=2, join=True)
    multi.assemble(home, False, 2)
    got = open(home + '/phased_sv.vcf').read()
    assert got == want and got.count('Duet.') == 40


def test_a_rank_that_never_finishes_is_killed(tmp_path):
    script = tmp_path / 'child.py'
    script.write_text('import time\ntime.sleep(600)\n')
    import time
    t0 = time.time()
    assert launch.spawn_ranks(2, [str(script)], timeout=2) == 124
    assert time.time() - t0 < 30


def test_spawn_ranks_refuses_a_parent_that_holds_a_gpu(monkeypatch):
    from duet_amd import _lib
    monkeypatch.setattr(_lib, 'CONTEXTS_CREATED', 1)
    with pytest.raises(RuntimeError, match='fresh interpreter'):
        launch.spawn_ranks(2, ['-c', 'pass'])


def test_division_by_zero_reaches_rank0_through_the_block(tmp_path):
    """svread + refread == 0 on ONE rank's contig (only reachable with -r 0): every rank still joins the collective,
    rank 0 reports it, and the file keeps its header only -- as upstream leaves it (sv_phasing.py:16, :123)."""
    from duet_amd import bamio
    home = str(tmp_path / 'w')
    os.makedirs(home + '/sv_calling')
    os.makedirs(home + '/snp_phasing')
    rec = '%s\t100\tid\tN\t<DEL>\t.\tPASS\tPRECISE;SVTYPE=DEL;SVLEN=-80;END=180;RE=%d;RNAMES=a,b;STRAND=+-\tGT:DR:DV:PL:GQ\t0/1:%d:5:1,2,3:9'
    with open(home + '/sv_calling/variants.vcf', 'w') as f:
        f.write('##contig=<ID=chr1,length=1000>\n' + rec % ('chr1', 5, 3) + '\n' + rec % ('chr2', 0, 0) + '\n')
    for c in ('chr1', 'chr2'):
        bamio.write_bam_from_sam_lines(home + '/snp_phasing/%s.bam' % c, [(c, 1000000)],
                                       ['a\t0\t%s\t90\t60\t*\t*\t0\t0\t*\t*\tNM:i:1\tHP:i:1\tPC:i:100\tPS:i:50' % c])
    mp.spawn(_worker, args=(2, launch.free_port(), home, 50, 0, str(tmp_path)), nprocs=2, join=True)
    assert open(os.path.join(str(tmp_path), 'rc0')).read() == str(multi.RC_DIV_ZERO)
    text = open(home + '/phased_sv.vcf').read()
    assert text.endswith('VALUE\n') and 'Duet.1' not in text


def test_spawn_ranks_sets_the_rendezvous_environment(tmp_path):
    script = tmp_path / 'child.py'
    script.write_text(
        'import os, sys\n'
        'import torch.distributed as d\n'
        'd.init_process_group("gloo")\n'
        'import torch\n'
        't = torch.tensor([int(os.environ["RANK"]) + 1])\n'
        'd.all_reduce(t)\n'
        'open(sys.argv[1] + os.environ["RANK"], "w").write(str(int(t)) + " " + os.environ["LOCAL_RANK"] + " " + os.environ["WORLD_SIZE"])\n'
        'd.destroy_process_group()\n')
    rc = launch.spawn_ranks(3, [str(script), str(tmp_path / 'out')], timeout=120)
    assert rc == 0
    for r in range(3):
        assert (tmp_path / ('out%d' % r)).read_text() == '6 %d 3' % r


def test_spawn_ranks_reports_the_first_failure_and_stops_the_rest(tmp_path):
    script = tmp_path / 'child.py'
    script.write_text('import os, sys, time\n'
                      'if os.environ["RANK"] == "1":\n    sys.exit(7)\n'
                      'time.sleep(60)\n')
    import time
    t0 = time.time()
    assert launch.spawn_ranks(2, [str(script)]) == 7
    assert time.time() - t0 < 30


def test_bench_parent_spawns_before_touching_the_gpu():
    """`python bench.py --gpus 2` started plainly must hand over to one process per GPU before importing torch;
    --launch-dry-run makes the children report their rank environment instead of running."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    out = subprocess.check_output([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--launch-dry-run'],
                                  env=env, timeout=120).decode()
    assert out.strip() == '{"launch_dry_run": true, "rank": 0, "world": 2, "torch_imported_by_parent": false}'


# ---------------------------------------------------------------------------------------------------------
# round 4: the ranks without torch -- duet_amd/comm.py's TCP star carries the blocks (what the one-GPU plumbing mode uses, and
# what hands RCCL's unique id around for the in-library collective on real multi-GPU nodes)
# ---------------------------------------------------------------------------------------------------------

def oracle_compute_np(sub, svlen_thres, suppread_thres, n_max):
    rc, pred, ps = c_oracle.ef(sub, svlen_thres, suppread_thres)
    return multi.block_np(pred, ps, n_max, sub.n_contigs), (multi.RC_DIV_ZERO if rc == -5 else 0)


def _worker_tcp(rank, world, port, home, svlen_thres, suppread_thres, out_dir):
    from duet_amd import comm
    star = comm.TcpStar(rank, world, '127.0.0.1', port, timeout=60)
    try:
        rc = multi.rank_body(home, svlen_thres, suppread_thres, 4, False, rank, world, oracle_compute_np, 'tcp',
                             gather=comm.HostGather(star))
        with open(os.path.join(out_dir, 'rc%d' % rank), 'w') as f:
            f.write(str(rc))
    finally:
        star.close()


@pytest.mark.parametrize('world', [2, 8])
def test_ranks_over_the_tcp_star(world, tmp_path):
    """24 contigs through `world` torch-free ranks: rendezvous, ONE host all-gather of the record blocks, per-rank rows."""
    from duet_amd import synth
    from oracle import ef_oracle
    home = str(tmp_path / 'g')
    synth.write_workdir(home, synth.bench_genome(50000, 11), dialect='cutesv', seed=11, write_sam=True)
    want = ef_oracle.sv_phasing_text(home, 50, 2)
    mp.spawn(_worker_tcp, args=(world, launch.free_port(), home, 50, 2, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), 'rc%d' % r)).read() == '0'
    multi.assemble(home, False, world)
    assert open(os.path.join(home, 'phased_sv.vcf')).read() == want


def test_tcp_star_primitives_and_timeout():
    import threading
    from duet_amd import comm
    port = launch.free_port()
    out = {}

    def run(rank):
        star = comm.TcpStar(rank, 3, '127.0.0.1', port, timeout=30)
        try:
            out[rank] = (star.bcast(b'id-of-rank-0' if rank == 0 else None), star.allgather(bytes([rank]) * (rank + 1)))
        finally:
            star.close()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    for r in range(3):
        assert out[r] == (b'id-of-rank-0', [bytes([0]), bytes([1, 1]), bytes([2, 2, 2])])
    # a rank that never arrives: rank 0 gives up after its timeout instead of hanging; so does a rank without a rank 0
    t0 = time.time()
    with pytest.raises(comm.CommError):
        comm.TcpStar(0, 2, '127.0.0.1', launch.free_port(), timeout=1.0)
    with pytest.raises(comm.CommError):
        comm.TcpStar(1, 2, '127.0.0.1', launch.free_port(), timeout=1.0)
    assert time.time() - t0 < 20
