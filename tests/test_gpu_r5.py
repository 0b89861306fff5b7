# coding=utf-8
"""-m gpu, round 5: the contig-sharded path's collective inside the library -- device-resident (duet_comm_ef_allgather: the
kernels write into the rank's record block, a small kernel appends the trailer, ncclAllGather on the kernels' stream) and
bounded (DUET_RDZV_TIMEOUT: a rank that waits for one that never comes gives up and leaves non-zero)."""
import os
import time

import numpy as np
import pytest

from tests.test_gpu_r2 import fresh_interpreter

pytestmark = pytest.mark.gpu

ONE_RANK = r'''
import os
os.environ["DUET_NO_TORCH"] = "1"
import numpy as np
from duet_amd import _lib, comm, dist
from oracle import c_oracle
from tests import soa_fuzz
ctx = _lib.Context(0)
star = comm.TcpStar(0, 1)
g = comm.RcclGather(ctx, star)
for seed, kw in ((1, dict(n_contigs=5)), (2, dict(n_contigs=3, cands_per_contig=(3000, 9000), reads_per_contig=(500, 900))),
                 (3, dict(n_contigs=2, allow_divzero=True, empty_contig_rate=0, no_seed_contig_rate=0)), (4, dict(n_contigs=2500, cands_per_contig=(0, 9), reads_per_contig=(4, 20)))):
    soa = soa_fuzz.random_soa(7000 + seed, **kw)
    supp = 0 if seed == 3 else 2
    rc, want_pred, want_ps = c_oracle.ef(soa, 50, supp)
    C, K = soa.n_cands, soa.n_contigs
    rng = np.random.RandomState(seed)
    # a slot per candidate as the ingest hands it over: 2 * contig + spelling
    contig = np.searchsorted(soa.cand_ctg_off[1:], np.arange(C), side="right")
    slots = (2 * contig + rng.randint(0, 2, C)).astype(np.uint32)
    n_max = C + 37
    out = g.ef_allgather(soa, 50, supp, slots, 2 * K, n_max)
    rb = dist.record_bytes(n_max)
    assert out.shape == (1, comm.block_bytes(n_max, 2 * K)) and comm.block_bytes(n_max, 2 * K) == rb + 16 + 16 * K
    status = int(out[0, rb:rb + 4].view(np.uint32)[0])
    pred, ps = dist.unpack_block(out[0], n_max, C)
    if rc == -5:
        assert status == 5, status
        continue
    assert rc == 0 and status == 0
    assert np.array_equal(pred, want_pred) and np.array_equal(ps, want_ps)
    assert not out[0, 4 * C:4 * n_max].any() and not out[0, 4 * n_max + C:rb].any()          # the other ranks' room stays zero
    kept = out[0, rb + 16:].view(np.uint64)
    assert np.array_equal(kept, np.bincount(slots[want_pred != 0], minlength=2 * K).astype(np.uint64))
# a rank without candidates (more ranks than contigs) still takes part
empty = soa_fuzz.random_soa(1, n_contigs=1, empty_contig_rate=1)
assert empty.n_cands == 0
out = g.ef_allgather(empty, 50, 2, np.zeros(0, np.uint32), 2, 100)
assert out.shape == (1, comm.block_bytes(100, 2)) and not out.any()
# round 6: what RCCL itself says about the communicator, and the rank-stamped pattern through it
info = g.info()
assert info["rank"] == 0 and info["world"] == 1 and info["rccl_ranks"] == 1 and info["rccl_rank"] == 0 and info["rccl_device"] == 0, info
g.selftest(4096)
# a rank whose own part fails (a null array with a non-zero count) still contributes a block -- status word with the top bit -- and
# then reports its own error (ADVICE round 5: it used to be dereferenced before any check)
bad = soa_fuzz.random_soa(7001, n_contigs=2)
prob, keep = _lib.problem_from_arrays(bad, 50, 2)
prob.cand_pos = None
import ctypes
nm = bad.n_cands + 5
outb = np.empty((1, comm.block_bytes(nm, 0)), dtype=np.uint8)
rc = ctx.lib.duet_comm_ef_allgather(g.handle, ctypes.byref(prob), None, 0, nm, outb.ctypes.data_as(ctypes.c_void_p))
assert rc == _lib.DUET_ERR_INVALID and "null array" in ctx.last_error(), (rc, ctx.last_error())
rbb = dist.record_bytes(nm)
st = int(outb[0, rbb:rbb + 4].view(np.uint32)[0])
assert st == (0x80000000 | (-_lib.DUET_ERR_INVALID)) and not outb[0, :rbb].any(), hex(st)
g.close()
ctx.close()
print("EF GATHER OK")
'''


def test_device_resident_rank_path_one_rank():
    """duet_comm_ef_allgather as ONE rank over real RCCL: results written into the record block on the device, the trailer
    (status word, rows kept per CHROM-text slot) by the trailer kernel, gathered, copied back once -- against the C oracle and a
    numpy count; thousands of slots (-a style contig universes) take the global-atomics path of the trailer kernel."""
    r = fresh_interpreter(ONE_RANK, {}, timeout=600)
    assert r.returncode == 0, (r.stdout.decode()[-1500:], r.stderr.decode()[-3000:])
    assert b'EF GATHER OK' in r.stdout


MISSING_RANK = r'''
import os, sys, time
os.environ["DUET_NO_TORCH"] = "1"
from duet_amd import _lib, comm
ctx = _lib.Context(0)
star = comm.TcpStar(0, 1, timeout=3.0)        # (a one-rank star: only the unique id's hand-over, nobody to wait for)
star.world = 2                                # ... but RCCL is told there are two ranks: ncclCommInitRank waits for rank 1
t0 = time.time()
try:
    comm.RcclGather(ctx, star)
except comm.CommTimeout as e:
    print("TIMEOUT after %.1f s: %s" % (time.time() - t0, e))
    sys.stdout.flush()
    os._exit(7)                               # as duet_amd/multi.py leaves: a helper thread still sits inside RCCL
print("NO TIMEOUT")
os._exit(0)
'''


def test_missing_rank_times_out_on_the_rccl_path():
    """DUET_RDZV_TIMEOUT bounds ncclCommInitRank (a helper thread inside the library, waited for with a deadline): rank 0 of a
    two-rank communicator whose rank 1 never shows up gives up after 3 s and the process leaves with a non-zero code -- it
    used to sit in ctypes until the launcher's DUET_RANK_TIMEOUT (3600 s)."""
    t0 = time.time()
    r = fresh_interpreter(MISSING_RANK, {}, timeout=120)
    dt = time.time() - t0
    assert r.returncode == 7, (r.returncode, r.stdout.decode()[-1000:], r.stderr.decode()[-2000:])
    assert b'TIMEOUT after' in r.stdout and b'did not finish within' in r.stdout
    assert dt < 60, dt                        # (interpreter start and HIP initialisation included; the wait itself is 3 s)


def test_two_ranks_over_real_rccl(tmp_path):
    """`duet --gpus 2` with the in-library collective over real RCCL -- when this box has two devices (the driver's GPU box has
    one: skipped there; an 8-GPU node runs it)."""
    import subprocess
    import sys
    n = int(subprocess.check_output([sys.executable, '-c', 'import torch; print(torch.cuda.device_count())']).decode().strip() or 0)
    if n < 2:
        pytest.skip('one visible device: two RCCL ranks cannot share it')
    from duet_amd import synth
    from tests import helpers as H
    home = str(tmp_path / 'w')
    synth.write_workdir(home, H.case_contigs('genome_small', 5), dialect='cutesv', seed=5, write_sam=False)
    r = fresh_interpreter('from duet_amd.sv_phasing import sv_phasing\nsv_phasing(%r, 50, 2, 4, False)\n' % home, {}, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    one = open(home + '/phased_sv.vcf', 'rb').read()
    os.remove(home + '/phased_sv.vcf')
    r = fresh_interpreter('from duet_amd.sv_phasing import sv_phasing\nsv_phasing(%r, 50, 2, 4, False, gpus=2)\n' % home,
                          {'DUET_RDZV_TIMEOUT': '120'}, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert open(home + '/phased_sv.vcf', 'rb').read() == one and one.count(b'Duet.') > 100


def test_svim_host_entry_refuses_contig_ids_beyond_the_depth_description():
    """duet_svim_phase_host has every array on the host: a mark whose contig id is not below n_contigs (= the depth
    description's contig count) would index the depth offsets and the E/F plan outside their K + 1 entries on the device; a
    depth_off that is not non-decreasing likewise.  Refused with DUET_ERR_INVALID before anything is uploaded."""
    from duet_amd import _lib, engine, synth
    from tests import helpers as H
    contigs = H.case_contigs('genome_small', 5)
    soa = engine.soa_from_synth(contigs)
    marks = synth.raw_marks(contigs, 1, reads_of=soa)
    depth, depth_off = synth.depth_bins(contigs, 1000, 1)
    ctx = _lib.Context(0)
    try:
        good = ctx.svim_host(marks, soa.read_tag, depth, depth_off, 1000, 50, 2)
        assert len(good['pred']) > 100
        bad = dict(marks)
        bad['contig'] = marks['contig'].copy()
        bad['contig'][len(bad['contig']) // 2] = len(depth_off) - 1            # one id too far
        with pytest.raises(_lib.DuetLibraryError, match='contig id'):
            ctx.svim_host(bad, soa.read_tag, depth, depth_off, 1000, 50, 2)
        off = depth_off.copy()
        off[2] = off[3] + 1
        with pytest.raises(_lib.DuetLibraryError, match='non-decreasing'):
            ctx.svim_host(marks, soa.read_tag, depth, off, 1000, 50, 2)
        again = ctx.svim_host(marks, soa.read_tag, depth, depth_off, 1000, 50, 2)     # the context is still good
        assert np.array_equal(again['pred'], good['pred']) and np.array_equal(again['ps'], good['ps'])
    finally:
        ctx.close()


def test_config2_on_the_literal_8d_generator_text_sha(tmp_path):
    """BASELINE configs[1] on SURVEY 8d's generator to the letter (20 % of the marks' names absent, pc = floor(Exp(600)):
    duet_amd.synth.bench_contig(literal_8d=True)) through the whole product path -- native ingest and Python host path, the HIP
    kernels, the rows on the device -- byte-identical to the unmodified reference's phased_sv.vcf (tests/golden/seeded_r5.json
    from make_golden_r5.py).  bench.py's extra.config2_literal_8d_generator times the same problem."""
    import json
    from tests import helpers as H
    from tests.test_gpu_parity import run_product
    with open(os.path.join(H.GOLDEN, 'seeded_r5.json')) as f:
        p = json.load(f)[0]
    home = str(tmp_path / 'config2_8d')
    H.build_case(home, p['kind'], p['seed'], p['dialect'], write_sam=False)
    got = run_product(home, p['svlen_thres'], p['suppread_thres'])
    assert sum(1 for l in got.splitlines() if not l.startswith('#')) == p['rows']
    assert H.sha256_bytes(got.encode()) == p['output_sha256']
