# coding=utf-8
"""SVIM mode over several ranks (duet_amd/svim_mode.py: rank_body) on CPU: world_size 2 and 8 over gloo.

Everything of `duet -b svim-gpu --gpus N` except the device pipeline runs here: contigs assigned by BAM size, every rank
extracts the signatures of ITS contigs' BAMs only (native, CPU code), a 16-byte count exchange, ONE all-gather of fixed-size
candidate records, rank 0 orders and writes the rows.  The per-rank compute is played by the composed C oracles
(cluster rule -> adapter -> E/F) -- tests only; the product's rank entry creates a device context and fails without one."""
import os

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from duet_amd import engine, launch, svim_mode, synth
from oracle import c_oracle
from tests import helpers as H


def oracle_compute(got, svlen_thres, suppread_thres, max_dist, depth_bin):
    """The fused pipeline's contract (include/duet_ef.h: duet_svim_phase_device) on the CPU."""
    K = len(got['depth_off']) - 1
    cl = c_oracle.cluster(got['contig'], got['type'], got['pos'], got['span'], max_dist=max_dist)
    N = len(cl['cand_pos'])
    off = cl['cand_off'].astype(np.int64)
    support = np.diff(off)
    k = cl['cand_contig'].astype(np.int64)
    depth, depth_off = got['depth'], got['depth_off'].astype(np.int64)
    nb = np.diff(depth_off)[k]
    bins = np.minimum(cl['cand_pos'].astype(np.int64) // depth_bin, np.maximum(nb - 1, 0))
    d = np.where(nb > 0, depth[np.minimum(depth_off[k] + bins, max(len(depth) - 1, 0))] if len(depth) else 0, 0).astype(np.int64)
    soa = engine.EfSoA(cand_ctg_off=np.searchsorted(k, np.arange(K + 1)), read_tag=got['read_tag'],
                       cand_pos=cl['cand_pos'], cand_svlen=cl['cand_span'], cand_svread=support,
                       cand_refread=np.maximum(d - support, 0), cand_gt_ok=np.ones(N, dtype=np.uint8),
                       cand_off=off, mark_read=got['read'][cl['order']])
    rc, pred, ps = c_oracle.ef(soa, svlen_thres, suppread_thres)
    if rc:
        raise ZeroDivisionError('division by zero')
    return dict(cand_contig=cl['cand_contig'], cand_type=cl['cand_type'], cand_pos=cl['cand_pos'], cand_span=cl['cand_span'],
                support=support, pred=pred, ps=ps)


def _worker(rank, world, port, home, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        rc = svim_mode.rank_body(home, 50, 2, 4, False, 0.9, rank, world, oracle_compute)
        with open(os.path.join(out_dir, 'rc%d' % rank), 'w') as f:
            f.write(str(rc))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_sharded_svim_mode_writes_the_single_process_rows(world, tmp_path):
    home = str(tmp_path / 'w')
    synth.write_svim_workdir(home, H.case_contigs('genome_small', 5), 5)
    chroms = svim_mode.init_chrom_list(False, home)
    one = svim_mode.phase_from_bams(home, 50, 2, 2, compute=oracle_compute)
    want = svim_mode.header_text(home, chroms) + svim_mode.rows_text(home, one)
    assert int(np.count_nonzero(one['pred'])) > 100
    with open(home + '/phased_sv.vcf', 'w') as f:
        f.write(svim_mode.header_text(home, chroms))
    mp.spawn(_worker, args=(world, launch.free_port(), home, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), 'rc%d' % r)).read() == '0'
    assert open(home + '/phased_sv.vcf').read() == want


def test_contigs_are_assigned_by_bam_size(tmp_path):
    home = str(tmp_path / 'w')
    synth.write_svim_workdir(home, H.case_contigs('genome_small', 5), 5)
    chroms = svim_mode.init_chrom_list(False, home)
    w = svim_mode.bam_weights(home, chroms)
    assert len(w) == 24 and min(w) > 0 and w[0] > w[20]          # chr1's BAM is larger than chr21's
    from duet_amd import dist as D
    owned = D.lpt_assign(w, 8)
    assert sorted(k for o in owned for k in o) == list(range(24))
    loads = [sum(w[k] for k in o) for o in owned]
    assert max(loads) < 1.25 * (sum(w) / 8.0)


def _worker_tcp(rank, world, port, home, out_dir):
    """round 4: the torch-free rank -- counts as a control message over the TCP star, ONE gather of the records"""
    from duet_amd import comm
    star = comm.TcpStar(rank, world, '127.0.0.1', port, timeout=60)
    try:
        rc = svim_mode.rank_body(home, 50, 2, 4, False, 0.9, rank, world, oracle_compute, star=star, gather=comm.HostGather(star))
        with open(os.path.join(out_dir, 'rc%d' % rank), 'w') as f:
            f.write(str(rc))
    finally:
        star.close()


@pytest.mark.parametrize('world', [2, 8])
def test_sharded_svim_mode_over_the_tcp_star(world, tmp_path):
    home = str(tmp_path / 'w')
    synth.write_svim_workdir(home, H.case_contigs('genome_small', 5), 5)
    chroms = svim_mode.init_chrom_list(False, home)
    one = svim_mode.phase_from_bams(home, 50, 2, 2, compute=oracle_compute)
    want = svim_mode.header_text(home, chroms) + svim_mode.rows_text(home, one)
    with open(home + '/phased_sv.vcf', 'w') as f:
        f.write(svim_mode.header_text(home, chroms))
    mp.spawn(_worker_tcp, args=(world, launch.free_port(), home, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), 'rc%d' % r)).read() == '0'
    assert open(home + '/phased_sv.vcf').read() == want
