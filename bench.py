#!/usr/bin/env python3
# coding=utf-8
"""bench.py -- throughput of Duet's step E/F hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (ef_classify -> ef_seed_sort -> ef_finalize, plus the single
all-gather of the per-candidate records when N > 1) over one batch of synthetic input that is already
resident in HBM.  Workload: BASELINE.json configs[1] -- one contig, ~1.0M SV support-read marks,
200k reads, 100k candidates (duet_amd.synth.bench_contig, seed 1).  With N > 1 every rank owns one
such contig (contig sharding, weak scaling) and the results are reassembled with one
all_gather_into_tensor over RCCL.  Rank 0 prints ONE JSON line.

`roofline` prices the dominant kernel (ef_classify): algorithmic bytes per launch
(12 B/mark + 18 B/candidate in + 5 B/candidate out + 8 B/read) / mean duration from HIP
events recorded on the launch stream during the timed region / 8 TB/s.
`cpu_baseline` times oracle/ef_oracle.c (scalar C port, 1 core) on the same arrays, rank 0, N=1.
"""

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np   # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def classify_bytes(soa):
    """Algorithmic HBM bytes of one ef_classify launch: mark_read 4 B + gathered tag 8 B per mark;
    cand_off 4 + svlen 4 + svread 4 + refread 4 + gt_ok 1 in and pred 1 + ps 4 out per candidate;
    the read-tag table counted once (8 B/read)."""
    return 12 * soa.n_marks + 22 * soa.n_cands + 8 * soa.n_reads


def pmc_traffic(soa):
    """HBM bytes per ef_classify launch from the committed rocprofv3 --pmc passes of this same command
    (profiles/r01_pmc_traffic.json: FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE exact), or None
    when no committed counter run matches the workload."""
    path = os.path.join(REPO, 'profiles', 'r01_pmc_traffic.json')
    try:
        with open(path) as f:
            d = json.load(f)
        if '%d marks' % soa.n_marks in d.get('workload', ''):
            return d['traffic_bytes_per_launch']
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_leg():
    """The CPU leg of this script -- with python_baseline() below (the pure-Python restatement, timed only) the ONLY
    place where bench.py touches oracle/.  The C restatement is timed as
    `cpu_baseline` and supplies the expected results every parity flag compares the GPU output with; it is the checker
    beside the measured path, never on it (the GPU path is libduet_ef.so through duet_amd/_lib.py and fails loudly
    without it)."""
    from oracle import c_oracle
    c_oracle.load()
    return c_oracle


def cpu_baseline(soa, budget_s=10.0):
    c_oracle = cpu_leg()
    t0 = time.perf_counter()
    c_oracle.ef(soa, 50, 2)
    one = time.perf_counter() - t0
    reps = max(1, min(2000, int(budget_s / max(one, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(reps):
        c_oracle.ef(soa, 50, 2)
    dt = (time.perf_counter() - t0) / reps
    return {'value': soa.n_marks / dt, 'unit': 'marks/s', 'cores': 1, 'kind': 'port',
            'sample': '%d passes of oracle/ef_oracle.c (scalar C restatement, gcc -O2) over the same %d-mark / '
                      '%d-candidate arrays; filter+class+seeds+vote+decision only, no text I/O' % (
                          reps, soa.n_marks, soa.n_cands),
            'ms_per_pass': dt * 1e3}


def python_baseline(contigs, n_cands=4000):
    """The Python oracle (closest in kind to upstream's own code) on a bounded slice, rows/s only as a
    side note: it needs the text inputs, so a small work dir is written to a temp dir."""
    import shutil
    import tempfile
    from duet_amd import synth
    from oracle import ef_oracle
    small = synth.bench_contig('1', 2 * n_cands, n_cands, 1)
    home = tempfile.mkdtemp(prefix='duet_bench_')
    try:
        synth.write_workdir(home, [small], write_bam=False)
        t0 = time.perf_counter()
        ef_oracle.sv_phasing_text(home, 50, 2)
        dt = time.perf_counter() - t0
    finally:
        shutil.rmtree(home, ignore_errors=True)
    marks = int(small.cand_off[-1])
    return {'value': marks / dt, 'unit': 'marks/s', 'cores': 1, 'kind': 'port',
            'sample': 'oracle/ef_oracle.py (pure Python, text VCF+SAM -> phased_sv.vcf text) on %d marks' % marks}


GATHER_GROUP = 8          # jobs per all-gather at N > 1 (two groups rotate, so a group's collective overlaps the next group's kernels)


def timed_steps(ctx, dp, steps, warmup, world, torch, dist_mod):
    """W warm-up + K timed steps.  With world > 1 every job's per-candidate records are all-gathered; the results of
    GATHER_GROUP consecutive jobs sit side by side in one buffer and go out in ONE asynchronous collective (its own
    RCCL stream) that overlaps the kernels of the following jobs -- a 0.5 MB-per-rank all-gather per 24 us job would be
    bound by the collective's latency, not by the work.  Every job is gathered completely before the clock stops."""
    from duet_amd.dist import GroupedGather
    stream = torch.cuda.current_stream().cuda_stream
    n_slots = len(dp.out_blocks)
    gg = GroupedGather(dp.out_storage, dp.out_blocks[0].numel(), world, GATHER_GROUP, dist_mod)

    def one():
        dp.run(ctx, stream, gg.next_slot())
        gg.job_enqueued()

    drain = gg.drain

    for _ in range(warmup):
        one()
    drain()
    ctx.check(stream)
    ctx.set_profiling(3)                 # HIP start/stop events on ef_classify's own dispatch, every 8th step
    ctx.profile_collect()
    if world > 1:
        dist_mod.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist_mod.barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile_collect()
    # after the timed region: every kernel bracketed by its own dispatch events (serialises the stream, so it
    # is kept out of `value`); these are the durations rocprofv3 --kernel-trace reports
    ctx.set_profiling(2)
    keep_slot = gg.slot
    spare = (keep_slot + 1) % n_slots if n_slots > 1 else 0
    for _ in range(min(steps, 50)):
        dp.run(ctx, stream, spare)
    torch.cuda.synchronize()
    iso = ctx.profile_collect()
    ctx.set_profiling(0)
    ctx.check(stream)
    last = gg.last_job_blocks()
    return dt, prof, iso, last, keep_slot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the larger single-GPU roofline points')
    ap.add_argument('--large', action='store_true', help='also run the 2e8-mark single-GPU point')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d '
                     '--master-addr 127.0.0.1 --master-port 29500 bench.py --gpus %d' % (args.gpus, args.gpus))
        sys.exit('--gpus %d does not match WORLD_SIZE %d' % (args.gpus, world))

    import torch
    from duet_amd import _lib, dist, engine, synth
    from duet_amd.devmem import DeviceProblem

    # DUET_BENCH_ONE_GPU=1 (plumbing test on a 1-GPU box only): every rank uses device 0 and the collective
    # goes through gloo instead of RCCL; never set for a measurement.
    one_gpu = os.environ.get('DUET_BENCH_ONE_GPU') == '1'
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist_mod = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if one_gpu:
            dist_mod.init_process_group('gloo')
        else:
            dist_mod.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    ctx = _lib.Context(local_rank)

    # ---- workload: one config-2 contig per rank ------------------------------------------------------
    label = synth.DEFAULT_CONTIGS[rank % len(synth.DEFAULT_CONTIGS)]
    contig = synth.bench_contig('1', 200000, 100000, 1 + rank, spelled='chr' + label)
    soa = engine.soa_from_synth([contig])
    n_max = soa.n_cands                      # every rank has exactly 100000 candidates
    dp = DeviceProblem(soa, 50, 2, device='cuda:%d' % local_rank, n_cands_max=n_max, n_out=2 * GATHER_GROUP if world > 1 else 2)

    with torch.cuda.stream(torch.cuda.Stream()):             # a stream of its own, not the legacy default stream
        dt, prof, iso, gathered, last_slot = timed_steps(ctx, dp, args.steps, args.warmup, world, torch, dist_mod)

    # correctness of what was timed (rank-local, against the C oracle) -- outside the timed region
    pred, ps = dp.results(last_slot)
    c_oracle = cpu_leg()
    rc, want_pred, want_ps = c_oracle.ef(soa, 50, 2)
    parity = bool(rc == 0 and np.array_equal(pred, want_pred) and np.array_equal(ps, want_ps))
    if world > 1:
        # this rank's slice of the gathered block must be what it computed
        mine = gathered[rank].cpu().numpy()
        gp, gs = dist.unpack_block(mine, n_max, soa.n_cands)
        parity = parity and bool(np.array_equal(gp, want_pred) and np.array_equal(gs, want_ps))

    marks_local = soa.n_marks
    if world > 1:
        t = torch.tensor([dt, float(marks_local), float(parity)], dtype=torch.float64, device='cuda')
        tmax = t.clone()
        dist_mod.all_reduce(tmax, op=dist_mod.ReduceOp.MAX)
        tsum = t.clone()
        dist_mod.all_reduce(tsum, op=dist_mod.ReduceOp.SUM)
        dt = float(tmax[0])
        marks_total = float(tsum[1])
        parity = bool(tsum[2] == world)
    else:
        marks_total = float(marks_local)

    if rank == 0:
        kname = _lib.KERNEL_NAMES[0]
        k_ms = float(prof.kernel_ms[0])
        abytes = classify_bytes(soa)
        achieved = abytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        out = {
            'metric': 'SV support-read marks clustered+phased /sec; bit-exact phased_sv.vcf vs ref',
            'value': marks_total * args.steps / dt, 'unit': 'marks/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u32/u64 integer + f64 threshold compares',
            'data': 'synthetic',
            'plumbing_test_one_gpu': one_gpu,
            'config': {'workload': 'BASELINE configs[1]: synthetic 1 contig per GPU, %d SV marks / %d candidates / '
                                   '%d tagged reads per contig, resident in HBM; step = classify+seed_sort+finalize%s'
                                   % (soa.n_marks, soa.n_cands, soa.n_reads,
                                      ' + all_gather_into_tensor of the 5 B/candidate records, %d jobs per collective (async, overlapping the next jobs)' % GATHER_GROUP if world > 1 else ''),
                       'marks_per_gpu': soa.n_marks, 'candidates_per_gpu': soa.n_cands, 'reads_per_gpu': soa.n_reads,
                       'parallelism': 'contig-sharded x%d' % world, 'svlen_thres': 50, 'suppread_thres': 2},
            'parity_vs_oracle': parity,
            'kernels_us_isolated': {n: round(float(iso.kernel_ms[i]) * 1e3, 2) for i, n in enumerate(_lib.KERNEL_NAMES)},
            'roofline': {'kernel': kname, 'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': pmc_traffic(soa),
                         'algorithmic_bytes_per_launch': abytes, 'launch_ms': k_ms,
                         'launches_timed': int(prof.n_profiled_runs),
                         'note': 'config 2 is 14.6 MB per launch (~1.8 us at peak): launch-latency bound; '
                                 'see extra.* for the bandwidth-bound sizes'},
        }
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(soa)
            out['cpu_baseline_python'] = python_baseline([contig])
        if world == 1 and not args.no_extra:
            out['extra'] = extra_points(ctx, torch, engine, synth, DeviceProblem, args.large)
            out['extra']['A0_clustering_config2_marks'] = cluster_point(ctx, torch, synth, [contig])
            out['extra']['fused_clustered_and_phased_config2'] = fused_point(ctx, torch, engine, synth, [contig])
            out['extra']['fused_clustered_and_phased_2e7_marks'] = fused_point(ctx, torch, engine, synth,
                                                                               synth.bench_genome(20000000, 3), runs=5)
            out['extra']['three_timed_regions_config2'] = abi_and_e2e(ctx, soa, contig, float(iso.total_ms))
            out['extra']['concurrent_jobs_config2'] = concurrent_jobs(torch, _lib, DeviceProblem, soa, args.steps)
        print(json.dumps(out))
        sys.stdout.flush()

    if world > 1:
        dist_mod.barrier()
        dist_mod.destroy_process_group()
    ctx.close()


def concurrent_jobs(torch, _lib, DeviceProblem, soa, steps, n_streams=4):
    """Throughput when independent jobs (here: the same config-2 problem) are kept in flight on several HIP
    streams, each with its own context/workspace and result block: a single job leaves most of the chip idle
    (one-workgroup seed sort, launch gaps), so jobs overlap.  Reported beside `value`, which stays the
    one-stream, one-job-at-a-time figure."""
    ctxs = [_lib.Context(0) for _ in range(n_streams)]
    dps = [DeviceProblem(soa, 50, 2) for _ in range(n_streams)]
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    for i in range(n_streams * 3):
        dps[i % n_streams].run(ctxs[i % n_streams], streams[i % n_streams].cuda_stream)
    torch.cuda.synchronize()
    n = max(steps, 100)
    t0 = time.perf_counter()
    for i in range(n):
        k = i % n_streams
        dps[k].run(ctxs[k], streams[k].cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    c_oracle = cpu_leg()
    rc, wp, ws = c_oracle.ef(soa, 50, 2)
    ok = True
    for k in range(n_streams):
        ctxs[k].check(streams[k].cuda_stream)
        pred, ps = dps[k].results()
        ok = ok and bool(np.array_equal(pred, wp) and np.array_equal(ps, ws))
    for c in ctxs:
        c.close()
    return {'streams': n_streams, 'jobs': n, 'us_per_job': dt * 1e6, 'marks_per_s': soa.n_marks / dt, 'parity_vs_oracle': ok}


def abi_and_e2e(ctx, soa, contig, kernels_ms):
    """SURVEY 8d asks for three timed regions: kernels only, the C-ABI call on host arrays (H2D + kernels + D2H),
    and end to end (caller VCF + haplotagged BAM on disk -> phased_sv.vcf on disk, native host path, -t 4)."""
    import shutil
    import tempfile
    from duet_amd import synth
    from duet_amd.sv_phasing import sv_phasing
    ctx.run_host(soa, 50, 2)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        ctx.run_host(soa, 50, 2)
    t_abi = (time.perf_counter() - t0) / reps
    home = tempfile.mkdtemp(prefix='duet_e2e_')
    try:
        synth.write_workdir(home, [contig], dialect='cutesv', seed=1, write_sam=False)
        sv_phasing(home, 50, 2, 4, False)
        t0 = time.perf_counter()
        for _ in range(3):
            sv_phasing(home, 50, 2, 4, False)
        t_e2e = (time.perf_counter() - t0) / 3
        size = os.path.getsize(os.path.join(home, 'phased_sv.vcf'))
        # the last step alone: (pred, ps) -> text of the rows, on the device vs on the host (one thread)
        from duet_amd.native import NativeIngest
        from duet_amd.read_file import init_chrom_list
        from duet_amd.devmem import DeviceProblem, device_rows
        ing = NativeIngest.load(os.path.join(home, 'sv_calling', 'variants.vcf'), home + '/snp_phasing/',
                                init_chrom_list(False, home), 4)
        rows = ing.rows()
        dpx = DeviceProblem(ing.soa, 50, 2)
        stx = dpx.run(ctx)
        ctx.check(stx)
        pr_, ps_ = dpx.results()
        device_rows(ctx, dpx, rows, stream=stx)
        t0 = time.perf_counter()
        for _ in range(5):
            body, n_rows = device_rows(ctx, dpx, rows, stream=stx)
        t_rows_dev = (time.perf_counter() - t0) / 5
        t0 = time.perf_counter()
        for _ in range(3):
            host_text = ing.emit(pr_, ps_, False)
        t_rows_host = (time.perf_counter() - t0) / 3
        rows_ok = bool(ing.header(False) + body == host_text)
        ing.close()
    finally:
        shutil.rmtree(home, ignore_errors=True)
    M = soa.n_marks
    return {'t_kernels_ms': kernels_ms, 'marks_per_s_kernels': M / (kernels_ms * 1e-3) if kernels_ms else None,
            't_abi_ms': t_abi * 1e3, 'marks_per_s_abi': M / t_abi,
            't_e2e_ms': t_e2e * 1e3, 'marks_per_s_e2e': M / t_e2e, 'phased_sv_vcf_bytes': size,
            'rows': {'n_rows': int(n_rows), 't_device_ms_incl_upload_and_download': t_rows_dev * 1e3,
                     't_host_ms_1_thread': t_rows_host * 1e3, 'identical': rows_ok},
            'note': 't_abi = duet_ef_run_host on pageable host arrays (PCIe both ways); t_e2e = duet_amd.sv_phasing.sv_phasing '
                    'with the native ingest (libduet_ingest.so), 4 host threads, files in the page cache'}


def cluster_point(ctx, torch, synth, contigs):
    """Stage A0 (span-position clustering, SVIM-mode) on the raw marks behind the same workload: jittered
    (pos, span) per support read, shuffled; sort + partition + average linkage + emit, resident in HBM.
    Reported beside the E/F number, not inside `value`: the reference computes A0 in an external binary."""
    from duet_amd.devmem import DeviceCluster
    c_oracle = cpu_leg()
    marks = synth.raw_marks(contigs, 1)
    dc = DeviceCluster(marks)
    for _ in range(3):
        dc.run(ctx)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        dc.run(ctx)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    want = c_oracle.cluster(marks['contig'], marks['type'], marks['pos'], marks['span'])
    cpu = time.perf_counter() - t0
    M = len(marks['pos'])
    return {'marks': M, 'candidates_found': dc.n_cands(), 'candidates_oracle': int(len(want['cand_off']) - 1),
            'ms_per_run': dt * 1e3, 'marks_per_s': M / dt, 'algorithmic_bytes_18_per_mark': 18 * M,
            'GBs_vs_B_A0': 18 * M / dt / 1e9, 'cpu_oracle_ms_1core': cpu * 1e3,
            'note': 'latency-bound at this size: ~30 small launches; the few 65..100-mark partitions are on the critical path'}


def fused_point(ctx, torch, engine, synth, contigs, runs=20):
    """The metric read literally -- marks clustered AND phased: duet_svim_phase_device on raw shuffled marks
    (A0 sort + linkage + emit, adapter, E/F) resident in HBM, checked against the two C oracles composed."""
    from duet_amd.devmem import DeviceSvim
    c_oracle = cpu_leg()
    soa = engine.soa_from_synth(contigs)
    marks = synth.raw_marks(contigs, 1, reads_of=soa)
    depth, depth_off = synth.depth_bins(contigs, 1000, 1)
    ds = DeviceSvim(marks, soa.read_tag, depth, depth_off, 1000, 50, 2)
    for _ in range(3):
        ds.run_fused(ctx, wait=False)
    torch.cuda.synchronize()
    n = runs
    t0 = time.perf_counter()
    for _ in range(n):
        ds.run_fused(ctx, wait=False)                   # nothing waits: E/F is planned on the device
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    got_async = ds.fetch()
    t0 = time.perf_counter()
    for _ in range(n):
        ds.run_fused(ctx, wait=True)                    # the variant that hands the candidate count back
    torch.cuda.synchronize()
    dt_wait = (time.perf_counter() - t0) / n
    got = ds.fetch()
    cl = c_oracle.cluster(marks['contig'], marks['type'], marks['pos'], marks['span'])
    N = len(cl['cand_pos'])
    support = np.diff(cl['cand_off'].astype(np.int64))
    k = cl['cand_contig'].astype(np.int64)
    nb = np.diff(depth_off)[k]
    bins = np.minimum(cl['cand_pos'].astype(np.int64) // 1000, np.maximum(nb - 1, 0))
    d = depth[depth_off[k] + bins].astype(np.int64)
    ref = engine.EfSoA(cand_ctg_off=np.searchsorted(k, np.arange(len(contigs) + 1)), read_tag=soa.read_tag,
                       cand_pos=cl['cand_pos'], cand_svlen=cl['cand_span'], cand_svread=support,
                       cand_refread=np.maximum(d - support, 0), cand_gt_ok=np.ones(N, dtype=np.uint8),
                       cand_off=cl['cand_off'], mark_read=marks['read'][cl['order']])
    rc, wp, ws = c_oracle.ef(ref, 50, 2)
    ok = bool(rc == 0 and ds.n_found == N and np.array_equal(got['pred'], wp) and np.array_equal(got['ps'], ws)
              and np.array_equal(got['cand_off'], cl['cand_off']) and np.array_equal(got['order'], cl['order'])
              and np.array_equal(got_async['pred'], wp) and np.array_equal(got_async['ps'], ws))
    M = len(marks['pos'])
    return {'marks': M, 'candidates_found': int(ds.n_found), 'phased': int((got['pred'] != 0).sum()),
            'ms_per_run': dt * 1e3, 'marks_per_s': M / dt, 'ms_per_run_with_count_returned': dt_wait * 1e3,
            'parity_vs_composed_oracles': ok,
            'note': 'asynchronous call (E/F planned on the device); the second figure is the variant with one host round trip'}


def extra_points(ctx, torch, engine, synth, DeviceProblem, large):
    """Single-GPU roofline at sizes where the path is bandwidth-bound rather than launch-bound
    (SURVEY.md section 8d): config 3's 2e7 marks on one GPU, optionally 2e8."""
    pts = {}
    sizes = [('config3_1gpu_2e7_marks', 20000000, 24)]
    if large:
        sizes.append(('1gpu_2e8_marks', 200000000, 24))
    for name, marks, _k in sizes:
        contigs = synth.bench_genome(marks, 3)
        soa = engine.soa_from_synth(contigs)
        del contigs
        dp = DeviceProblem(soa, 50, 2)
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            dp.run(ctx, stream)
        ctx.check(stream)
        ctx.set_profiling(2)
        ctx.profile_collect()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            dp.run(ctx, stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        prof = ctx.profile_collect()
        ctx.set_profiling(0)
        c_oracle = cpu_leg()
        pred, ps = dp.results()
        rc, want_pred, want_ps = c_oracle.ef(soa, 50, 2)
        ok = bool(rc == 0 and np.array_equal(pred, want_pred) and np.array_equal(ps, want_ps))
        ab = classify_bytes(soa)
        kms = float(prof.kernel_ms[0])
        pts[name] = {'parity_vs_oracle': ok, 'marks': soa.n_marks, 'candidates': soa.n_cands, 'reads': soa.n_reads,
                     'contigs': soa.n_contigs, 'ms_per_step': dt * 1e3, 'marks_per_s': soa.n_marks / dt,
                     'kernels_ms': {k: float(prof.kernel_ms[i]) for i, k in enumerate(('ef_classify', 'ef_seed_sort', 'ef_finalize'))},
                     'classify_GBs': ab / (kms * 1e-3) / 1e9 if kms > 0 else 0.0,
                     'classify_frac_of_8TBs': ab / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS if kms > 0 else 0.0,
                     'pipeline_GBs_B_EF': soa.algorithmic_bytes() / (float(prof.total_ms) * 1e-3) / 1e9
                     if prof.total_ms > 0 else 0.0}
        del dp
        torch.cuda.empty_cache()
    return pts


if __name__ == '__main__':
    main()
