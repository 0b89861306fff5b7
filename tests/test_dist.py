# coding=utf-8
"""Contig sharding + the single all-gather, world_size 2 over gloo on CPU. The per-rank compute is
played by the C oracle here (tests only); on the GPU box bench.py runs the HIP kernels in its place."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from duet_amd import dist as D
from oracle import c_oracle
from tests import soa_fuzz


def test_lpt_assignment_balances_and_is_deterministic():
    w = [8, 7, 6, 5, 4, 3, 2, 1, 1, 1]
    owned = D.lpt_assign(w, 3)
    assert sorted(k for o in owned for k in o) == list(range(len(w)))
    loads = [sum(w[k] for k in o) for o in owned]
    assert max(loads) - min(loads) <= 2
    assert owned == D.lpt_assign(w, 3)
    assert D.lpt_assign([5], 4) == [[0], [], [], []]


def test_shard_and_merge_equals_whole():
    soa = soa_fuzz.random_soa(5, n_contigs=7, empty_contig_rate=4)
    rc, want_pred, want_ps = c_oracle.ef(soa, 50, 2)
    owned = D.lpt_assign(D.contig_mark_counts(soa), 3)
    per_rank = []
    for o in owned:
        sub = D.shard_soa(soa, o)
        rc, p, s = c_oracle.ef(sub, 50, 2)
        assert rc == 0
        per_rank.append((p, s))
    pred, ps = D.merge_results(soa, owned, per_rank)
    assert np.array_equal(pred, want_pred) and np.array_equal(ps, want_ps)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, seed, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        soa = soa_fuzz.random_soa(seed, n_contigs=6, empty_contig_rate=5)
        owned = D.lpt_assign(D.contig_mark_counts(soa), world)
        sizes = D.shard_sizes(soa, owned)
        n_max = max(max(sizes), 1)
        sub = D.shard_soa(soa, owned[rank])
        rc, pred, ps = c_oracle.ef(sub, 50, 2)          # stand-in for the HIP kernels (CPU test only)
        assert rc == 0
        block = np.zeros(D.record_bytes(n_max), dtype=np.uint8)
        block[:4 * sub.n_cands] = ps.view(np.uint8)
        block[4 * n_max:4 * n_max + sub.n_cands] = pred
        gathered = D.allgather_records(torch.from_numpy(block), world).numpy()      # the ONE collective
        per_rank = [D.unpack_block(gathered[r], n_max, sizes[r]) for r in range(world)]
        got_pred, got_ps = D.merge_results(soa, owned, per_rank)
        rc, want_pred, want_ps = c_oracle.ef(soa, 50, 2)
        ok = np.array_equal(got_pred, want_pred) and np.array_equal(got_ps, want_ps)
        with open(os.path.join(out_dir, 'rank%d' % rank), 'w') as f:
            f.write('ok' if ok else 'mismatch')
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('seed', [3, 4])
def test_two_ranks_gloo(tmp_path, seed):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, seed, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        with open(str(tmp_path / ('rank%d' % r))) as f:
            assert f.read() == 'ok'


def _grouped_worker(rank, world, port, n_jobs, out_dir):
    """bench.py's N > 1 exchange: jobs write fixed-size record blocks, `group` of them per asynchronous all-gather,
    two groups rotating.  Every job's block must arrive on every rank, also for a last, partly filled group."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        rb, G = 48, 4
        storage = torch.zeros(2 * G * rb, dtype=torch.uint8)
        gg = D.GroupedGather(storage, rb, world, G, dist, always=world == 1)
        ok = True
        seen = 0
        for job in range(n_jobs):
            slot = gg.next_slot()
            storage[slot * rb:(slot + 1) * rb] = (7 * job + 31 * rank + torch.arange(rb)) % 251      # the "results" of this job
            gg.job_enqueued()
            if gg.filled == 0 and gg.last is not None:                      # a collective was just issued
                g, k = gg.last
                gg.pending[g].wait()
                got = gg.gathered[g][:world * k * rb].view(world, k, rb)
                for r in range(world):
                    for i in range(k):
                        want = (7 * (job - k + 1 + i) + 31 * r + torch.arange(rb)) % 251
                        ok = ok and bool(torch.equal(got[r, i].to(torch.int64), want))
                seen += k
        gg.drain()
        last = gg.last_job_blocks()
        for r in range(world):
            want = (7 * (n_jobs - 1) + 31 * r + torch.arange(rb)) % 251
            ok = ok and bool(torch.equal(last[r].to(torch.int64), want))
        ok = ok and seen == (n_jobs // G) * G
        with open(os.path.join(out_dir, 'rank%d' % rank), 'w') as f:
            f.write('ok' if ok else 'mismatch')
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_jobs', [3, 8, 21])
def test_grouped_gather_two_ranks_gloo(tmp_path, n_jobs):
    port = _free_port()
    mp.spawn(_grouped_worker, args=(2, port, n_jobs, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        with open(str(tmp_path / ('rank%d' % r))) as f:
            assert f.read() == 'ok'


def test_grouped_gather_one_rank_group_still_communicates(tmp_path):
    """always=True: a process group of ONE rank goes through the collectives all the same (bench.py's DUET_BENCH_RCCL_SELF check
    of the real backend on a one-GPU box; gloo here)."""
    port = _free_port()
    mp.spawn(_grouped_worker, args=(1, port, 9, str(tmp_path)), nprocs=1, join=True)
    with open(str(tmp_path / 'rank0')) as f:
        assert f.read() == 'ok'


def test_grouped_gather_single_rank_rotates_slots():
    storage = torch.zeros(2 * 16, dtype=torch.uint8)
    gg = D.GroupedGather(storage, 16, 1, 8, None)
    slots = []
    for _ in range(5):
        slots.append(gg.next_slot())
        gg.job_enqueued()
    gg.drain()
    assert slots == [0, 1, 0, 1, 0] and gg.last_job_blocks() is None
