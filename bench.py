#!/usr/bin/env python3
# coding=utf-8
"""bench.py -- throughput of Duet's step E/F hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]          # started plainly: spawns one process per GPU itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM.

N = 1   BASELINE.json configs[1]: one contig, ~1.0 M SV support-read marks, 200 k reads, 100 k candidates
        (duet_amd.synth.bench_contig, seed 1); step = ef_classify -> ef_finalize_own (two launches up to 2048 tiles of 256 candidates and 64 contigs; beyond: ef_classify -> ef_seed_sort -> ef_finalize).  `value` is the phasing
        (E/F) rate; `value_clustered_and_phased` is the metric read literally: the same marks, raw and shuffled, clustered
        (stage A0) AND phased in one device pipeline, with its own roofline block.  `roofline` prices the dominant kernel
        of the step (ef_classify) at config 2, where it is launch-latency bound; `roofline_bandwidth_bound` is the same
        kernel on 2e8 marks (3.2 GB), the only size that is unambiguously HBM traffic.  `scaling_point_1gpu` is configs[2]
        (what --gpus N shards) on this one GPU: the point an N > 1 line's `value` continues -- this line's own `value` is a
        different workload.  `summed_kernels_frac_B_EF`: B_EF over the SUM of the three kernels' times, of 8 TB/s, at 1e6 /
        2e7 / 2e8 marks (BASELINE.md section 3's definition).  `roofline_clustered_and_phased.traffic`: FETCH_SIZE x 2 +
        WRITE_SIZE of every kernel of the fused pipeline per run, from the committed collection profiles/*fused_traffic*.json.
N > 1   BASELINE.json configs[2]: the synthetic whole genome (24 contigs, 2e7 marks) as ONE problem, contigs assigned to
        ranks longest-processing-time-first, each rank runs the three kernels on its shard, exactly ONE all-gather (RCCL over
        xGMI) per problem reassembles the records -- through the collective the product ships, duet_comm_* inside libduet_ef.so
        (`--collective torch`: torch.distributed "nccl"; the line's `collective` says which); "scaling": "strong".  Rank 0 also
        times the same problem on its GPU alone (`same_problem_on_1_gpu`), so the line carries its own 1-GPU reference
        (`value_vs_1gpu_same_problem` = value / that).
Rank 0 prints ONE JSON line.

Algorithmic bytes: ef_classify 12 B/mark + 22 B/candidate + 8 B/read per launch; the whole
E/F step B_EF = 12 M + 27 C + 8 R; stage A0 18 B/mark.  Kernel durations come from HIP events on the kernel's own
dispatch (duet_ctx_set_profiling), recorded inside the timed region.
`cpu_baseline` times oracle/ef_oracle.c (scalar C port, 1 core) on the same arrays, rank 0, N = 1.
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


def pmc_traffic_entry(soa):
    """The committed rocprofv3 --pmc passes for this workload (profiles/*pmc_traffic*.json, the newest round that has it:
    FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE exact) -> (HBM bytes per ef_classify launch, where they come
    from), or (None, None) when no committed counter run matches the workload.  Counters cannot be collected inside this
    run (a --pmc pass is a separate profiler run), so the line says where the number was measured."""
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, 'profiles', '*pmc_traffic*.json')), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            for e in (d if isinstance(d, list) else [d]):
                if '%d marks' % soa.n_marks in e.get('workload', ''):
                    src = {'file': 'profiles/' + os.path.basename(path), 'collected': e.get('collected'),
                           'counter_files': e.get('files'), 'kernel_trace_avg_us': e.get('kernel_trace_avg_us'),
                           'note': 'committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same workload, NOT measured '
                                   'in this run; kernel-trace average of that collection: %s us' % e.get('kernel_trace_avg_us')}
                    return e['traffic_bytes_per_launch'], src
        except (OSError, ValueError, KeyError):
            pass
    return None, None


def pmc_traffic(soa):
    return pmc_traffic_entry(soa)[0]


def fused_traffic(marks):
    """Counter traffic of ONE run of the fused clustered + phased pipeline (every kernel's FETCH_SIZE doubled + WRITE_SIZE,
    summed) from the committed collection profiles/*fused_traffic*.json -> (bytes per run, per-kernel table, source) or
    (None, None, None).  Like pmc_traffic_entry: a --pmc pass is a separate profiler run, not part of this one."""
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, 'profiles', '*fused_traffic*.json')), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            for e in d:
                if abs(int(e['marks']) - int(marks)) <= max(2000, int(marks) // 500):
                    return e['traffic_bytes_per_run'], e['per_kernel'], {'file': 'profiles/' + os.path.basename(path),
                                                                         'collected': e.get('collected'), 'command': e.get('command')}
        except (OSError, ValueError, KeyError):
            pass
    return None, None, None


def roofline_block(kernel, soa, launch_ms, launches, source, note=None, workload=None):
    """roofline for ef_classify on `soa`: achieved = algorithmic bytes / launch time; real_hbm_frac = committed counter
    bytes / the SAME launch time / 8 TB/s (the 2x gap between the two is the convention: SURVEY 8d charges 8 B per gathered
    tag, the distinct tag table is fetched once)."""
    ab = classify_bytes(soa)
    traffic, tsrc = pmc_traffic_entry(soa)
    # the launch time must be reproducible from profiles/ -- the line carries the in-run figure (HIP events
    # on the kernel's dispatch) AND the committed kernel-trace average of the same workload, and `achieved` / `frac` use the SLOWER
    # of the two (launch_ms); the in-run figure stays beside it as launch_ms_in_run
    in_run_ms = launch_ms
    prof_ms = (float(tsrc['kernel_trace_avg_us']) * 1e-3) if (tsrc and tsrc.get('kernel_trace_avg_us')) else None
    if prof_ms and prof_ms > launch_ms:
        launch_ms = prof_ms
    gbs = ab / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    out = {'kernel': kernel, 'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
           'traffic': traffic, 'traffic_source': tsrc,
           'real_hbm_frac': (traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and launch_ms > 0) else None,
           'algorithmic_bytes_per_launch': ab, 'launch_ms': launch_ms, 'launch_ms_in_run': in_run_ms, 'launch_ms_profiles': prof_ms,
           'frac_in_run': (ab / (in_run_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if in_run_ms > 0 else None,
           'launches_timed': int(launches), 'launch_ms_source': source}
    if note:
        out['note'] = note
    if workload:
        out['workload'] = workload
    return out


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


GATHER_GROUP = 8          # extra.weak_grouped only: jobs per all-gather (two groups rotate)


def host_info():
    """CPU model and core count of the box the CPU legs run on (north_star: 'core count stated')."""
    model = None
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count()
    return {'host_cpu_model': model, 'host_cores': os.cpu_count(), 'host_cores_usable': usable}


def cpu_baseline_ncore(soa, one_core_marks_per_s, budget_s=6.0, max_threads=64):
    """The same scalar C restatement on T host threads at once (each thread passes over the same read-only arrays
    into its own outputs; ctypes releases the GIL) -- the N-core figure SURVEY 8d asks for beside the 1-core one."""
    import threading
    c_oracle = cpu_leg()
    info = host_info()
    T = max(1, min(int(info['host_cores_usable'] or 1), max_threads))
    reps = max(1, int(budget_s * one_core_marks_per_s / max(soa.n_marks, 1)))

    def work():
        for _ in range(reps):
            c_oracle.ef(soa, 50, 2)

    ths = [threading.Thread(target=work) for _ in range(T)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    return {'value': soa.n_marks * reps * T / dt, 'unit': 'marks/s', 'cores': T, 'kind': 'port',
            'sample': '%d threads x %d passes of oracle/ef_oracle.c over the same arrays (throughput of independent '
                      'passes; the restatement itself is scalar)' % (T, reps)}


def cpu_calibration():
    """profiles/cpu_calibration.json: upstream's Python step E/F against the C restatement, both timed in the
    development container on the config-2 inputs (tools/calibrate_cpu.py)."""
    try:
        with open(os.path.join(REPO, 'profiles', 'cpu_calibration.json')) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def step_kernels_profiled(ctx, dp, stream, torch, n=64):
    """After a timed region: every kernel bracketed by its own dispatch events (serialises the stream, so it is kept out
    of `value`); these are the durations rocprofv3 --kernel-trace reports."""
    ctx.set_profiling(2)
    spare = len(dp.out_blocks) - 1
    for _ in range(n):
        dp.run(ctx, stream, spare)
    torch.cuda.synchronize()
    iso = ctx.profile_collect()
    ctx.set_profiling(0)
    ctx.check(stream)
    return iso


class TorchColl(object):
    """Everything through torch.distributed: backend "nccl" (= RCCL; `--collective torch`) or "gloo" (DUET_BENCH_ONE_GPU=1,
    the one-GPU plumbing mode).  Control plane and data path alike."""

    def __init__(self, torch, dist_mod, backend, local_rank, group=None):
        self.torch, self.dist, self.backend, self.group = torch, dist_mod, backend, group
        self.name = 'torch.distributed "%s"' % backend
        self.dev = 'cpu' if backend == 'gloo' else 'cuda'
        self.world, self.rank = dist_mod.get_world_size(), dist_mod.get_rank()

    def barrier(self):
        self.dist.barrier(group=self.group)

    def reduce_max_sum(self, values):
        t = self.torch.tensor(values, dtype=self.torch.float64, device=self.dev)
        tmax, tsum = t.clone(), t.clone()
        self.dist.all_reduce(tmax, op=self.dist.ReduceOp.MAX, group=self.group)
        self.dist.all_reduce(tsum, op=self.dist.ReduceOp.SUM, group=self.group)
        return [float(x) for x in tmax], [float(x) for x in tsum]

    def all_gather_object(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj, group=self.group)
        return out

    def all_gather_into_tensor(self, dst, src, async_op=False):
        if self.backend == 'gloo' and src.device.type != 'cpu':
            # plumbing mode: gloo gathers host tensors; staged (synchronously) through the host
            h = src.cpu()
            out = self.torch.empty(dst.numel(), dtype=dst.dtype)
            self.dist.all_gather_into_tensor(out, h, group=self.group)
            dst.copy_(out)
            return _Done()
        w = self.dist.all_gather_into_tensor(dst, src, async_op=async_op, group=self.group)
        return w if async_op else _Done()

    def version(self):
        try:
            return '.'.join(str(v) for v in self.torch.cuda.nccl.version())
        except Exception:
            return None

    def close(self):
        pass


class _Done(object):
    def wait(self):
        pass


class _StreamWork(object):
    """What an asynchronous collective hands back: wait() makes the CURRENT stream wait for it (as torch's work objects do)."""

    def __init__(self, torch, ev):
        self.torch, self.ev = torch, ev

    def wait(self):
        self.torch.cuda.current_stream().wait_event(self.ev)


class DuetColl(object):
    """What `duet --gpus N` ships (default for N > 1): the DATA PATH's all-gather is the collective inside libduet_ef.so
    (duet_comm_*: ncclCommInitRank + ncclAllGather over xGMI, RCCL loaded by the library) on a side stream beside the kernels'
    stream; the bench's CONTROL plane -- barriers around the timed region, max-over-ranks of a few floats, the topology census,
    the hand-over of RCCL's unique id -- goes through torch.distributed "gloo" on the host (the product's ranks use a TCP star
    for that: duet_amd/comm.py)."""

    def __init__(self, torch, dist_mod, ctx, timeout=300.0):
        from duet_amd import comm
        self.torch, self.dist, self.ctx = torch, dist_mod, ctx
        self.ctl = TorchColl(torch, dist_mod, 'gloo', 0)
        self.world, self.rank = self.ctl.world, self.ctl.rank
        self.backend = 'duet_comm'
        self.name = 'duet_comm_* in libduet_ef.so (RCCL ncclAllGather); control plane: torch.distributed "gloo"'
        dist = dist_mod

        class _IdCarrier(object):                       # what comm.RcclGather needs of a star: rank, world, timeout, bcast
            rank, world = self.rank, self.world

            def __init__(self, timeout):
                self.timeout = timeout

            def bcast(self, data=None):
                box = [data]
                dist.broadcast_object_list(box, src=0)
                return box[0]

        self.gather = comm.RcclGather(ctx, _IdCarrier(float(timeout)))
        self.stream = torch.cuda.Stream()
        # before anything is timed: every rank gathers a rank-stamped pattern through the communicator and checks every slot
        # (duet_comm_selftest), and RCCL itself is asked how many ranks it connected (duet_comm_info -> the line's rccl_ranks_seen)
        self.selftest = None
        if os.environ.get('DUET_COMM_SELFTEST', '1') != '0':
            self.gather.selftest(4096)
            self.selftest = 'ok: 4096 rank-stamped words per rank, every slot checked on every rank'
        self.info = self.gather.info()

    def barrier(self):
        self.ctl.barrier()

    def reduce_max_sum(self, values):
        return self.ctl.reduce_max_sum(values)

    def all_gather_object(self, obj):
        return self.ctl.all_gather_object(obj)

    def all_gather_into_tensor(self, dst, src, async_op=False):
        torch = self.torch
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        self.stream.wait_event(ready)
        self.gather.allgather_device(src.data_ptr(), src.numel() * src.element_size(), dst.data_ptr(), self.stream.cuda_stream)
        done = torch.cuda.Event()
        done.record(self.stream)
        w = _StreamWork(torch, done)
        if not async_op:
            w.wait()
        return w

    def version(self):
        v = int(self.ctx.lib.duet_comm_rccl_version(self.ctx.handle))
        return '%d.%d.%d' % (v // 10000, v // 100 % 100, v % 100) if v > 0 else None

    def close(self):
        self.gather.close()


PREHEAT_MS = 40.0


def preheat(ctx, dp, stream, torch):
    """Untimed, before the W warm-up steps: the same step for ~40 ms of device time.  The timed region of the default run is 200 steps of
    18 us -- 4 ms -- and starts after a minute of host-side input generation with the device idle: in one of round 6's three collections
    that region ran 13 % slower than in the others (20.4 against 18.0 us per step; the same binary's kernel trace and every later
    measurement of the same process agreed with the faster figure) -- the device's clocks had not come up yet.  Returns the steps it ran."""
    n = 0
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < PREHEAT_MS:
        for _ in range(100):
            dp.run(ctx, stream, 0)
        torch.cuda.synchronize()
        n += 100
    return n


def timed_steps(ctx, dp, steps, warmup, world, torch, coll, group):
    """W warm-up + K timed steps of (ef_classify -> [ef_seed_sort ->] ef_finalize [-> all-gather]; two launches for small shards, see duet_ef.hip: ef_finalize_own).  With world > 1 the
    record blocks of `group` consecutive jobs go out in ONE asynchronous all_gather_into_tensor on RCCL's stream, which
    overlaps the kernels of the following job(s); group = 1 is one collective per problem (the sharded configs[2] run).
    Every job is gathered completely before the clock stops."""
    from duet_amd.dist import GroupedGather
    stream = torch.cuda.current_stream().cuda_stream
    gg = GroupedGather(dp.out_storage, dp.out_blocks[0].numel(), world, group, coll, always=coll is not None)

    def one():
        dp.run(ctx, stream, gg.next_slot())
        gg.job_enqueued()

    for _ in range(warmup):
        one()
    gg.drain()
    ctx.check(stream)
    ctx.set_profiling(3)                 # HIP start/stop events on ef_classify's own dispatch, every 8th step
    ctx.profile_collect()
    if coll is not None:
        coll.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    gg.drain()
    torch.cuda.synchronize()
    if coll is not None:
        coll.barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile_collect()
    ctx.set_profiling(0)
    ctx.check(stream)
    return dt, prof, gg


def topology(torch, coll, rank, world, local_rank, one_gpu):
    """What proves N ranks on N devices: per rank its device index, name, PCI bus id and the XCD/CU count; the
    collective path and RCCL's version."""
    props = torch.cuda.get_device_properties(local_rank)
    mine = {'rank': rank, 'device': local_rank, 'name': props.name, 'cus': props.multi_processor_count,
            'hbm_GiB': round(props.total_memory / 2 ** 30, 1),
            'pci_bus_id': getattr(props, 'pci_bus_id', None), 'pid': os.getpid()}
    info = getattr(coll, 'info', None)
    if isinstance(info, dict):
        mine['rccl'] = info                                   # what ncclCommCount / ncclCommUserRank / ncclCommCuDevice say on this rank
    allr = coll.all_gather_object(mine)
    seen = [r['rccl'].get('rccl_ranks') for r in allr if isinstance(r.get('rccl'), dict)]
    return {'backend': coll.backend, 'collective': coll.name, 'world_size': coll.world, 'rccl_version': coll.version(),
            # the smallest communicator size any rank's RCCL reports (ncclCommCount through duet_comm_info): == world_size when RCCL
            # itself connected every rank; null on the torch.distributed paths
            'rccl_ranks_seen': (min(seen) if seen and len(seen) == len(allr) else None),
            'comm_selftest': getattr(coll, 'selftest', None),
            'one_gpu_plumbing_mode': one_gpu, 'ranks': allr,
            'distinct_devices': len(set((r['device'], r['pci_bus_id']) for r in allr))}


def sharded_run(args, ctx, torch, coll, rank, world, local_rank, one_gpu):
    """N > 1: BASELINE configs[2] -- the synthetic whole genome (24 contigs, 2e7 marks) as ONE problem, contigs assigned
    to ranks longest-processing-time-first on mark counts (duet_amd/dist.py), each rank runs the three kernels on its
    shard, exactly ONE all_gather_into_tensor per problem reassembles the (pred, ps) records on every rank.  Strong
    scaling: the problem is the same at every N."""
    from duet_amd import dist, engine, synth
    from duet_amd.devmem import DeviceProblem
    contigs = synth.bench_genome(args.genome_marks, 3)
    soa = engine.soa_from_synth(contigs)
    del contigs
    weights = dist.contig_mark_counts(soa)
    owned = dist.lpt_assign(weights, world)
    sizes = dist.shard_sizes(soa, owned)
    n_max = max(max(sizes), 1)
    sub = dist.shard_soa(soa, owned[rank])
    dev = 'cuda:%d' % local_rank
    dp = DeviceProblem(sub, 50, 2, device=dev, n_cands_max=n_max, n_out=4)
    stream_obj = torch.cuda.Stream()
    with torch.cuda.stream(stream_obj):
        stream = torch.cuda.current_stream().cuda_stream
        preheat(ctx, dp, stream, torch)
        dt, prof, gg = timed_steps(ctx, dp, args.steps, args.warmup, world, torch, coll, 1)
        gathered = gg.last_job_blocks()
        # --- outside the timed region: the pieces on their own ---------------------------------------------------
        iso = step_kernels_profiled(ctx, dp, stream, torch, n=min(args.steps, 50))
        n = max(10, min(args.steps, 50))
        coll.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            dp.run(ctx, stream, 3)
        torch.cuda.synchronize()
        t_kern = (time.perf_counter() - t0) / n
        src = dp.out_blocks[3]
        dst = torch.empty(world * src.numel(), dtype=torch.uint8, device=src.device)
        for _ in range(3):
            coll.all_gather_into_tensor(dst, src)
        torch.cuda.synchronize()
        coll.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            coll.all_gather_into_tensor(dst, src)
        torch.cuda.synchronize()
        t_gather = (time.perf_counter() - t0) / n
    kms = float(prof.kernel_ms[0])
    ab = classify_bytes(sub)
    mine = {'rank': rank, 'contigs': [int(k) for k in owned[rank]], 'marks': sub.n_marks, 'candidates': sub.n_cands,
            'reads': sub.n_reads, 'ef_classify_ms': kms, 'ef_classify_launches_timed': int(prof.n_profiled_runs),
            'ef_classify_algorithmic_bytes': ab,
            'ef_classify_GBs': ab / (kms * 1e-3) / 1e9 if kms > 0 else 0.0,
            'ef_classify_frac_of_8TBs': ab / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS if kms > 0 else 0.0,
            'kernels_us_isolated': {k: round(float(iso.kernel_ms[i]) * 1e3, 2) for i, k in enumerate(('ef_classify', 'ef_seed_sort', 'ef_finalize'))},
            'kernels_only_ms_per_step': t_kern * 1e3, 'gather_only_us': t_gather * 1e6}
    per_rank = coll.all_gather_object(mine)
    (dt,), _ = coll.reduce_max_sum([dt])
    topo = topology(torch, coll, rank, world, local_rank, one_gpu)

    out = None
    same = None
    if rank == 0:
        # the merged call set of the LAST timed problem against the C oracle on the unsharded problem
        c_oracle = cpu_leg()
        g = gathered.cpu().numpy()
        got = dist.merge_results(soa, owned, [dist.unpack_block(g[r], n_max, sizes[r]) for r in range(world)])
        rc, want_pred, want_ps = c_oracle.ef(soa, 50, 2)
        parity = bool(rc == 0 and np.array_equal(got[0], want_pred) and np.array_equal(got[1], want_ps))
        # the SAME problem on this rank's GPU alone (the other ranks wait at the barrier below): the 1-GPU point of the
        # strong-scaling curve, measured in the same run
        del dp
        torch.cuda.empty_cache()
        dpf = DeviceProblem(soa, 50, 2, device=dev)
        with torch.cuda.stream(stream_obj):
            for _ in range(3):
                dpf.run(ctx, stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                dpf.run(ctx, stream)
            torch.cuda.synchronize()
            t1 = (time.perf_counter() - t0) / 20
            ctx.check(stream)
        p1, s1 = dpf.results()
        same = {'ms_per_step': t1 * 1e3, 'marks_per_s': soa.n_marks / t1,
                'parity_vs_oracle': bool(np.array_equal(p1, want_pred) and np.array_equal(s1, want_ps)),
                'note': 'no collective: one GPU holds the whole call set'}
        del dpf
        torch.cuda.empty_cache()
        slow = max(per_rank, key=lambda r: r['ef_classify_ms'])
        loads = [r['marks'] for r in per_rank]
        out = {
            'metric': METRIC, 'value': soa.n_marks * args.steps / dt, 'unit': 'marks/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': DTYPE, 'data': 'synthetic',
            'value_is': VALUE_IS.replace('value_clustered_and_phased', 'extra.fused_clustered_and_phased_sharded.marks_per_s'),
            'plumbing_test_one_gpu': one_gpu,
            'config': {'workload': 'BASELINE configs[2]: synthetic whole genome chr1-22,X,Y, %d SV marks / %d candidates / %d '
                                   'tagged reads in %d contigs, resident in HBM, contigs LPT-sharded over %d GPUs; step = '
                                   'classify[+seed_sort]+finalize per rank + ONE all-gather of the 5 B/candidate '
                                   'records per problem (asynchronous, overlapping the next problem\'s kernels)'
                                   % (soa.n_marks, soa.n_cands, soa.n_reads, soa.n_contigs, world),
                       'marks': soa.n_marks, 'candidates': soa.n_cands, 'reads': soa.n_reads, 'contigs': soa.n_contigs,
                       'parallelism': 'contig-sharded x%d (LPT on mark counts)' % world, 'svlen_thres': 50,
                       'suppread_thres': 2},
            'parity_vs_oracle': parity,
            'collective': coll.name,
            'roofline': {'kernel': 'ef_classify', 'bound': 'hbm', 'achieved': slow['ef_classify_GBs'], 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': slow['ef_classify_frac_of_8TBs'], 'traffic': None,
                         'algorithmic_bytes_per_launch': slow['ef_classify_algorithmic_bytes'],
                         'launch_ms': slow['ef_classify_ms'], 'launches_timed': slow['ef_classify_launches_timed'],
                         'note': 'the rank whose ef_classify launch is longest (rank %d); every rank is in per_rank' % slow['rank']},
            'sharding': {'lpt_imbalance_max_over_mean_marks': max(loads) / (sum(loads) / float(world)),
                         'candidates_per_rank': sizes, 'record_bytes_per_rank': dist.record_bytes(n_max)},
            'gather': {'collectives_per_problem': 1, 'bytes_contributed_per_rank': dist.record_bytes(n_max),
                       'us_isolated_max_over_ranks': max(r['gather_only_us'] for r in per_rank),
                       'kernels_only_ms_per_step_max_over_ranks': max(r['kernels_only_ms_per_step'] for r in per_rank)},
            'per_rank': per_rank,
            'topology': topo,
            'same_problem_on_1_gpu': same,
            # (what a scaling curve should be read against: the SAME call set on one GPU of this run -- the N = 1 line of
            # bench.py is BASELINE configs[1], a different workload; its `scaling_point_1gpu` is this problem too)
            'value_vs_1gpu_same_problem': (soa.n_marks * args.steps / dt) / same['marks_per_s'],
        }
    coll.barrier()
    return out


def fused_sharded_run(args, ctx, torch, coll, rank, world, local_rank):
    """extra (N > 1): the CLUSTERED pipeline sharded like BASELINE configs[3] runs it (`--sv_caller svim`, 8 GPUs): the raw,
    shuffled marks of the whole-genome problem go to the rank that owns their contig (stage A0's partitions never cross a
    contig), every rank runs duet_svim_phase_device (A0 + E/F) on its marks, and ONE all_gather_into_tensor of fixed-size
    candidate records (ps, pos, span, pred: 13 B, slots sized once from the warm-up's counts) reassembles the call set.
    Strong scaling: the same 2e7 marks at every N."""
    from duet_amd import dist, engine, synth
    from duet_amd.devmem import DeviceSvim
    err = None
    try:
        contigs = synth.bench_genome(args.genome_marks, 3)
        soa = engine.soa_from_synth(contigs)
        owned = dist.lpt_assign(dist.contig_mark_counts(soa), world)
        marks = synth.raw_marks(contigs, 1, reads_of=soa)
        depth, depth_off = synth.depth_bins(contigs, 1000, 1)
        del contigs
        sel = np.isin(marks['contig'], np.array(owned[rank], dtype=marks['contig'].dtype))
        sub = {k: np.ascontiguousarray(v[sel]) for k, v in marks.items()}
        M_all, M_mine = len(marks['pos']), int(sel.sum())
        del marks
        ds = DeviceSvim(sub, soa.read_tag, depth, depth_off, 1000, 50, 2, device='cuda:%d' % local_rank)
        stream = torch.cuda.current_stream().cuda_stream
        ds.run_fused(ctx, stream, wait=True)
        n_mine = int(ds.n_found)
    except Exception as e:
        err = '%s: %s' % (type(e).__name__, e)
    # every rank learns whether every rank got this far BEFORE the first collective of the timed part: a rank that failed
    # on its own must not leave the others waiting in an all-gather
    states = coll.all_gather_object({'err': err, 'n': None if err else n_mine})
    if any(x['err'] for x in states):
        return {'error': err or 'another rank failed while setting up: %s' % [x['err'] for x in states if x['err']][0]}
    counts = [int(x['n']) for x in states]
    n_max = max(max(counts), 1)
    block = torch.zeros(13 * n_max, dtype=torch.uint8, device='cuda')
    gathered = torch.empty(world * 13 * n_max, dtype=torch.uint8, device='cuda')
    ps_b = ds.out_ps.view(torch.uint8)
    pos_b, span_b, pred_b = ds.keep['out_cand_pos'], ds.keep['out_cand_span'], ds.out_pred

    def one():
        ds.run_fused(ctx, stream, wait=False)
        n = min(n_mine, n_max)
        block[0:4 * n].copy_(ps_b[:4 * n])
        block[4 * n_max:4 * n_max + 4 * n].copy_(pos_b[:4 * n])
        block[8 * n_max:8 * n_max + 4 * n].copy_(span_b[:4 * n])
        block[12 * n_max:12 * n_max + n].copy_(pred_b[:n])
        coll.all_gather_into_tensor(gathered, block)

    for _ in range(max(2, args.warmup // 4)):
        one()
    coll.barrier()
    torch.cuda.synchronize()
    steps = max(3, min(args.steps, 20))
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    coll.barrier()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ds.run_fused(ctx, stream, wait=False)
    torch.cuda.synchronize()
    t_alone = (time.perf_counter() - t0) / steps
    ctx.check(stream)
    parity = None
    if rank == 0:
        # rank 0's shard against the composed C oracles (cluster rule -> adapter -> E/F) on the same marks
        c_oracle = cpu_leg()
        got = ds.fetch()
        cl = c_oracle.cluster(sub['contig'], sub['type'], sub['pos'], sub['span'])
        N = len(cl['cand_pos'])
        support = np.diff(cl['cand_off'].astype(np.int64))
        k = cl['cand_contig'].astype(np.int64)
        nb = np.diff(depth_off)[k]
        bins = np.minimum(cl['cand_pos'].astype(np.int64) // 1000, np.maximum(nb - 1, 0))
        d = depth[depth_off[k] + bins].astype(np.int64)
        ref = engine.EfSoA(cand_ctg_off=np.searchsorted(k, np.arange(soa.n_contigs + 1)), read_tag=soa.read_tag,
                           cand_pos=cl['cand_pos'], cand_svlen=cl['cand_span'], cand_svread=support,
                           cand_refread=np.maximum(d - support, 0), cand_gt_ok=np.ones(N, dtype=np.uint8),
                           cand_off=cl['cand_off'], mark_read=sub['read'][cl['order']])
        rc, wp, ws = c_oracle.ef(ref, 50, 2)
        g0 = gathered[:13 * n_max].cpu().numpy()
        parity = bool(rc == 0 and n_mine == N and np.array_equal(got['pred'], wp) and np.array_equal(got['ps'], ws)
                      and np.array_equal(g0[:4 * N].view(np.uint32), ws) and np.array_equal(g0[12 * n_max:12 * n_max + N], wp))
    (dt, t_alone_max), (_, _) = coll.reduce_max_sum([dt, t_alone])
    cand = counts
    return {'scaling': 'strong', 'workload': 'BASELINE configs[3] in shape: %d raw shuffled SV marks of 24 contigs, contig-sharded over %d '
                                             'GPUs; step = duet_svim_phase_device (A0 + E/F) per rank + ONE all-gather of 13 B '
                                             'candidate records' % (M_all, world),
            'marks': M_all, 'marks_rank0': M_mine, 'candidates_per_rank': cand, 'steps': steps,
            'marks_per_s': M_all * steps / dt, 'ms_per_step': dt / steps * 1e3,
            'pipeline_only_ms_per_step_max_over_ranks': t_alone_max * 1e3, 'record_bytes_per_rank': 13 * n_max,
            'parity_rank0_vs_composed_oracles': parity,
            'note': 'clustering rule: parity unpinned (own rule, oracle/cluster_oracle.c); E/F pinned'}


def weak_grouped_run(args, ctx, torch, coll, rank, world, local_rank):
    """extra (N > 1): round 1's weak-scaling variant -- one config-2 contig per rank, the record blocks of GATHER_GROUP
    consecutive jobs in one asynchronous collective."""
    from duet_amd import dist, engine, synth
    from duet_amd.devmem import DeviceProblem
    label = synth.DEFAULT_CONTIGS[rank % len(synth.DEFAULT_CONTIGS)]
    contig = synth.bench_contig('1', 200000, 100000, 1 + rank, spelled='chr' + label)
    soa = engine.soa_from_synth([contig])
    n_max = soa.n_cands
    dp = DeviceProblem(soa, 50, 2, device='cuda:%d' % local_rank, n_cands_max=n_max, n_out=2 * GATHER_GROUP)
    with torch.cuda.stream(torch.cuda.Stream()):
        dt, prof, gg = timed_steps(ctx, dp, args.steps, args.warmup, world, torch, coll, GATHER_GROUP)
        gathered = gg.last_job_blocks()
    c_oracle = cpu_leg()
    rc, want_pred, want_ps = c_oracle.ef(soa, 50, 2)
    gp, gs = dist.unpack_block(gathered[rank].cpu().numpy(), n_max, soa.n_cands)
    ok = float(rc == 0 and np.array_equal(gp, want_pred) and np.array_equal(gs, want_ps))
    (dt, _, _), (_, marks, oks) = coll.reduce_max_sum([dt, float(soa.n_marks), ok])
    return {'scaling': 'weak', 'workload': 'one BASELINE configs[1] contig per rank, %d jobs per all-gather' % GATHER_GROUP,
            'marks_per_s': marks * args.steps / dt, 'ms_per_step': dt / args.steps * 1e3, 'parity_vs_oracle': bool(oks == world)}


LINE_BUDGET = 8000        # bytes of the ONE printed JSON line (round 4's 22.5 kB line was not parsed by the driver; 11.9 kB was)

_ROOF_KEYS = ('kernel', 'kernels', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'real_hbm_frac', 'launch_ms_in_run', 'launch_ms_profiles', 'frac_in_run',
              'algorithmic_bytes_per_launch', 'algorithmic_bytes_per_run', 'launch_ms', 'run_ms', 'launches_timed', 'workload')


def _num(x):
    """Floats to 6 significant digits (the line is a record, not an archive; the detail file keeps everything)."""
    if isinstance(x, float):
        return float('%.6g' % x)
    if isinstance(x, dict):
        return {k: _num(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v) for v in x]
    return x


def _slim_roofline(r):
    if not isinstance(r, dict):
        return r
    out = {k: r[k] for k in _ROOF_KEYS if k in r}
    if isinstance(out.get('kernels'), str):
        out['kernels'] = out['kernels'][:60]
    src = r.get('traffic_source')
    if isinstance(src, dict) and src.get('file'):
        out['traffic_file'] = src['file']
    return out


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(full, detail_path=None):
    """The ONE line rank 0 prints: the contract's keys, `roofline`, `cpu_baseline` and a handful of numbers per extra
    point -- no per-kernel tables, no prose.  Everything else (`full`) goes to the detail file and to stderr.  Pure
    function of `full` (tests/test_bench_line.py builds it from canned numbers)."""
    top = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
           'dtype', 'data', 'parity_vs_oracle', 'plumbing_test_one_gpu', 'kernels_us_isolated', 'value_clustered_and_phased',
           'ms_per_step_clustered_and_phased', 'parity_clustered_and_phased_vs_composed_oracles', 'value_vs_1gpu_same_problem',
           'collective', 'value_is')
    out = {k: full[k] for k in top if k in full}
    cfg = full.get('config', {})
    out['config'] = _pick(cfg, ('workload', 'marks', 'candidates', 'reads', 'contigs', 'marks_per_gpu', 'candidates_per_gpu',
                                'reads_per_gpu', 'parallelism', 'svlen_thres', 'suppread_thres', 'generator', 'device_preheat_steps'))
    if isinstance(out['config'].get('workload'), str):
        out['config']['workload'] = out['config']['workload'][:200]
    for k in ('roofline', 'roofline_clustered_and_phased', 'roofline_bandwidth_bound'):
        if k in full:
            out[k] = _slim_roofline(full[k])
    if 'cpu_baseline' in full:
        cb = full['cpu_baseline']
        out['cpu_baseline'] = _pick(cb, ('value', 'unit', 'cores', 'kind', 'ms_per_pass', 'host_cpu_model', 'host_cores'))
        out['cpu_baseline']['sample'] = str(cb.get('sample', ''))[:160]
        cal = cb.get('calibration')
        if isinstance(cal, dict):
            out['cpu_baseline']['reference_python_estimate_marks_per_s'] = cal.get('reference_python_ef_marks_per_s_here_estimate')
    for k in ('cpu_baseline_ncore', 'cpu_baseline_python'):
        if k in full:
            out[k] = _pick(full[k], ('value', 'unit', 'cores', 'kind'))
    if 'scaling_point_1gpu' in full:
        out['scaling_point_1gpu'] = _pick(full['scaling_point_1gpu'], ('value', 'unit', 'ms_per_step', 'marks'))
    if 'summed_kernels_frac_B_EF' in full:
        out['summed_kernels_frac_B_EF'] = {k: v.get('frac_of_8TBs') for k, v in full['summed_kernels_frac_B_EF'].items()}
    # N > 1
    if 'sharding' in full:
        out['sharding'] = _pick(full['sharding'], ('lpt_imbalance_max_over_mean_marks', 'record_bytes_per_rank'))
    if 'gather' in full:
        out['gather'] = full['gather']
    if 'per_rank' in full:
        pr = full['per_rank']
        out['per_rank'] = {'marks': [r.get('marks') for r in pr], 'ef_classify_us': [round(r.get('ef_classify_ms', 0) * 1e3, 2) for r in pr],
                           'kernels_only_us': [round(r.get('kernels_only_ms_per_step', 0) * 1e3, 2) for r in pr]}
    if 'topology' in full:
        tp = full['topology']
        out['topology'] = _pick(tp, ('backend', 'world_size', 'rccl_version', 'rccl_ranks_seen', 'comm_selftest', 'one_gpu_plumbing_mode', 'distinct_devices'))
        out['rccl_ranks_seen'] = tp.get('rccl_ranks_seen')
        out['topology']['devices'] = ['%s@%s' % (r.get('device'), r.get('pci_bus_id')) for r in tp.get('ranks', [])]
    if 'same_problem_on_1_gpu' in full and full['same_problem_on_1_gpu']:
        out['same_problem_on_1_gpu'] = _pick(full['same_problem_on_1_gpu'], ('ms_per_step', 'marks_per_s', 'parity_vs_oracle'))
    ex = full.get('extra')
    if isinstance(ex, dict):
        keep = ('marks', 'candidates', 'candidates_found', 'phased', 'ms_per_step', 'ms_per_run', 'marks_per_s', 'parity_vs_oracle',
                'parity_vs_composed_oracles', 'parity_rank0_vs_composed_oracles', 'ms_per_run_with_count_returned', 'kernels_ms',
                'kernels_us', 'us_per_job', 'streams', 't_kernels_ms', 't_abi_ms', 't_e2e_ms', 'scaling', 'error',
                'pipeline_only_ms_per_step_max_over_ranks', 'ef_classify_us', 'lane_efficiency', 'degree_mean', 'degree_max',
                'sha256_matches_reference', 'ms_per_step_e_f', 't_e2e_ms_t4', 'marks_per_s_t4')
        sx = {}
        for name, pt in ex.items():
            if not isinstance(pt, dict):
                continue
            e = _pick(pt, keep)
            rf = pt.get('roofline')
            if isinstance(rf, dict):
                e['frac'] = rf.get('frac')
                e['traffic'] = rf.get('traffic')
                e['real_hbm_frac'] = rf.get('real_hbm_frac')
            sx[name] = e
        out['extra'] = sx
    if detail_path:
        out['detail_file'] = detail_path
    out = _num(out)
    # the hard bound: shed the optional blocks, largest first, until the line fits
    for k in ('extra', 'per_rank', 'topology', 'kernels_us_isolated', 'summed_kernels_frac_B_EF', 'cpu_baseline_python',
              'roofline_bandwidth_bound', 'roofline_clustered_and_phased'):
        if len(json.dumps(out)) <= LINE_BUDGET:
            break
        out.pop(k, None)
    return out


def write_detail(full, world):
    """The complete record (per-kernel traffic tables, notes, every extra point): gpurun_out/ when it exists (it is merged
    back from the GPU box), else the temp dir.  NOT echoed on stderr: the driver's record is a bounded tail of stdout + stderr
    and the printed line must stay inside it.  Returns the path or None."""
    import tempfile
    text = json.dumps(full)
    for d in (os.path.join(REPO, 'gpurun_out'), tempfile.gettempdir()):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, 'bench_detail_n%d.json' % world)
            with open(path, 'w') as f:
                f.write(text + '\n')
            return os.path.relpath(path, REPO) if path.startswith(REPO) else path
        except OSError:
            continue
    return None


METRIC = 'SV support-read marks clustered+phased /sec; bit-exact phased_sv.vcf vs ref'
VALUE_IS = 'phased_only (step E/F on pre-clustered candidates); the literal clustered+phased figure is value_clustered_and_phased'
DTYPE = 'u32/u64 integer + f64 threshold compares'


def single_gpu_run(args, ctx, torch):
    """N = 1: BASELINE configs[1] -- one contig, ~1.0 M marks / 100 k candidates / 200 k reads (160 k tagged)."""
    from duet_amd import _lib, engine, synth
    from duet_amd.devmem import DeviceProblem
    contig = synth.bench_contig('1', 200000, 100000, 1, spelled='chr1')
    soa = engine.soa_from_synth([contig])
    dp = DeviceProblem(soa, 50, 2, n_out=3)
    with torch.cuda.stream(torch.cuda.Stream()):             # a stream of its own, not the legacy default stream
        stream = torch.cuda.current_stream().cuda_stream
        preheat_steps = preheat(ctx, dp, stream, torch)
        dt, prof, gg = timed_steps(ctx, dp, args.steps, args.warmup, 1, torch, None, 1)
        last_slot = gg.slot
        iso = step_kernels_profiled(ctx, dp, stream, torch, n=max(min(args.steps, 100), 50))
    pred, ps = dp.results(last_slot)
    c_oracle = cpu_leg()
    rc, want_pred, want_ps = c_oracle.ef(soa, 50, 2)
    parity = bool(rc == 0 and np.array_equal(pred, want_pred) and np.array_equal(ps, want_ps))

    kname = _lib.KERNEL_NAMES[0]
    # HIP events on ef_classify's own dispatch: inside the timed region every 8th step carries a pair; with fewer than 16
    # such launches (a short --steps) the serialised loop that follows the region (>= 50 launches, same kernel, same
    # stream, same arguments) supplies the duration instead
    k_ms, k_n, k_src = float(prof.kernel_ms[0]), int(prof.n_profiled_runs), 'events on every 8th step inside the timed region'
    if k_n < 16:
        k_ms, k_n, k_src = float(iso.kernel_ms[0]), int(iso.n_profiled_runs), \
            'events on every launch of the serialised loop right after the timed region (the region held only %d sampled launches)' % k_n
    out = {
        'metric': METRIC, 'value': soa.n_marks * args.steps / dt, 'unit': 'marks/s', 'n_gpus': 1, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE, 'data': 'synthetic',
        # `value` is the PHASING step (E/F) on candidates that are already clustered; the metric read
        # literally -- the same marks raw, clustered (stage A0) and phased in one device pipeline -- is value_clustered_and_phased
        'value_is': VALUE_IS,
        'config': {'workload': 'BASELINE configs[1]: synthetic 1 contig, %d SV marks / %d candidates / %d tagged reads, '
                               'resident in HBM; step = ef_classify + ef_finalize_own, two launches (phasing, E/F; ef_seed_sort 0: not launched at this size); the clustered+phased '
                               'figure for the same marks is value_clustered_and_phased'
                               % (soa.n_marks, soa.n_cands, soa.n_reads),
                   'marks_per_gpu': soa.n_marks, 'candidates_per_gpu': soa.n_cands, 'reads_per_gpu': soa.n_reads,
                   'parallelism': 'contig-sharded x1', 'svlen_thres': 50, 'suppread_thres': 2,
                   'device_preheat_steps': preheat_steps,      # untimed, in front of the W warm-up steps (bench.py: preheat)
                   'generator_vs_SURVEY_8d': 'marks with no SAM line: 5 % (8d: 20 % of names absent); pc ~ geometric(1/2)*400 + '
                                             'U[0,400), mean ~600, capped at 8100 + 3 % U[8101,20000] (8d: exp mean 600); '
                                             'the reference sha256 in tests/golden/seeded.json pins exactly these inputs'},
        'parity_vs_oracle': parity,
        'kernels_us_isolated': {n: round(float(iso.kernel_ms[i]) * 1e3, 2) for i, n in enumerate(_lib.KERNEL_NAMES)},
        'roofline': roofline_block(kname, soa, k_ms, k_n, k_src,
                                   note='config 2 is 15.5 MB per launch (~1.9 us at peak): launch-latency bound; '
                                        'roofline_bandwidth_bound below is the same kernel at 2e8 marks'),
    }
    if not args.no_extra:
        # the metric read literally: raw marks clustered (A0) AND phased (E/F) in one device pipeline
        fused = fused_point(ctx, torch, engine, synth, [contig])
        out['value_clustered_and_phased'] = fused['marks_per_s']
        out['ms_per_step_clustered_and_phased'] = fused['ms_per_run']
        out['roofline_clustered_and_phased'] = fused['roofline']
        out['parity_clustered_and_phased_vs_composed_oracles'] = fused['parity_vs_composed_oracles']
    if not args.no_cpu_baseline:
        cb = cpu_baseline(soa)
        cb.update(host_info())
        cal = cpu_calibration()
        if cal:
            cb['calibration'] = {'reference_python_over_port': cal['ratio_reference_over_port'],
                                 'measured': cal['where'], 'reference_python_ef_marks_per_s_here_estimate':
                                 cb['value'] / cal['ratio_reference_over_port'],
                                 'note': 'BASELINE.md section 4: upstream\'s Python cannot travel to the GPU box; estimate = '
                                         'port on this host / (reference time / port time, both measured in the dev container, '
                                         'same inputs, same region: filter..decision, sv_phasing_fn.py:189-228)'}
        out['cpu_baseline'] = cb
        out['cpu_baseline_ncore'] = cpu_baseline_ncore(soa, cb['value'])
        out['cpu_baseline_python'] = python_baseline([contig])
    if not args.no_extra:
        ex = extra_points(ctx, torch, engine, synth, DeviceProblem, not args.no_large)
        out['extra'] = ex
        big = ex.get('1gpu_2e8_marks') or ex.get('config3_1gpu_2e7_marks')
        out['roofline_bandwidth_bound'] = big['roofline']
        out['same_problem_as_multi_gpu_runs'] = {'workload': 'BASELINE configs[2] on one GPU (what --gpus N shards)',
                                                 'marks_per_s': ex['config3_1gpu_2e7_marks']['marks_per_s'],
                                                 'ms_per_step': ex['config3_1gpu_2e7_marks']['ms_per_step']}
        # the 1-GPU point of the strong-scaling curve that `--gpus N` lines continue (their `value` is this workload's; this
        # line's `value` is configs[1]'s): compare value(N) with scaling_point_1gpu.value, not with this line's value
        out['scaling_point_1gpu'] = {'workload': 'BASELINE configs[2]: 24 contigs, %d marks, one GPU, no collective' % ex['config3_1gpu_2e7_marks']['marks'],
                                     'value': ex['config3_1gpu_2e7_marks']['marks_per_s'], 'unit': 'marks/s',
                                     'ms_per_step': ex['config3_1gpu_2e7_marks']['ms_per_step']}
        # BASELINE.md section 3's definition: B_EF = 12 M + 27 C + 8 R over the SUM of the three kernels' times, of 8 TB/s
        sk = {'config2': {'B_EF': soa.algorithmic_bytes(), 'kernels_ms_sum': float(sum(iso.kernel_ms[i] for i in range(3)))}}
        for name, key in (('2e7_marks', 'config3_1gpu_2e7_marks'), ('2e8_marks', '1gpu_2e8_marks')):
            if key in ex:
                sk[name] = {'B_EF': 12 * ex[key]['marks'] + 27 * ex[key]['candidates'] + 8 * ex[key]['reads'],
                            'kernels_ms_sum': float(sum(ex[key]['kernels_ms'].values()))}
        for v in sk.values():
            v['frac_of_8TBs'] = v['B_EF'] / (v['kernels_ms_sum'] * 1e-3) / 1e9 / HBM_PEAK_GBS if v['kernels_ms_sum'] > 0 else None
        out['summed_kernels_frac_B_EF'] = sk
        ex['A0_clustering_config2_marks'] = cluster_point(ctx, torch, synth, [contig])
        ex['fused_clustered_and_phased_config2'] = fused
        tailed = {}
        ex['fused_clustered_and_phased_2e7_marks'] = fused_point(ctx, torch, engine, synth,
                                                                  synth.bench_genome(20000000, 3), runs=5, tailed=tailed)
        ex['ef_tailed_sizes_2e7'] = tailed
        ex['fused_clustered_and_phased_2e7_marks_scan_order'] = fused_point(ctx, torch, engine, synth,
                                                                             synth.bench_genome(20000000, 3), runs=5, scan_order=True)
        ex['config2_literal_8d_generator'] = literal_8d_point(ctx, torch, engine, synth, DeviceProblem, args.steps)
        ex['three_timed_regions_config2'] = abi_and_e2e(ctx, soa, contig, float(iso.total_ms))
        if not args.no_large:
            try:
                ex['three_timed_regions_config3'] = e2e_config3(ctx, torch, ex['config3_1gpu_2e7_marks']['ms_per_step'])
            except Exception as e:                       # (disk space for the 564 MB text, ...): said, not hidden
                ex['three_timed_regions_config3'] = {'error': '%s: %s' % (type(e).__name__, e)}
        ex['concurrent_jobs_config2'] = concurrent_jobs(torch, _lib, DeviceProblem, soa, args.steps)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the larger single-GPU roofline points')
    ap.add_argument('--no-large', action='store_true', help='skip the 2e8-mark single-GPU point (about a minute of host-side generation)')
    ap.add_argument('--large', action='store_true', help='(default now; kept for round-1 command lines)')
    ap.add_argument('--genome-marks', type=int, default=20000000, help='N > 1: marks of the sharded whole-genome problem')
    ap.add_argument('--collective', choices=('duet', 'torch'), default='duet',
                    help='N > 1: the all-gather of the data path -- "duet": duet_comm_* inside libduet_ef.so, what `duet --gpus N` ships '
                         '(default); "torch": torch.distributed backend "nccl" (rounds 1-4)')
    ap.add_argument('--launch-dry-run', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()

    from duet_amd import launch
    if args.gpus > 1 and not launch.under_launcher():
        # started plainly: hand over to one fresh process per GPU BEFORE anything here has touched torch or HIP
        sys.exit(launch.spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:],
                                    extra_env={'DUET_BENCH_PARENT_TORCH': '1' if 'torch' in sys.modules else '0'}))

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        sys.exit('--gpus %d does not match WORLD_SIZE %d' % (args.gpus, world))
    if args.launch_dry_run:
        if rank == 0:
            print(json.dumps({'launch_dry_run': True, 'rank': rank, 'world': world,
                              'torch_imported_by_parent': os.environ.get('DUET_BENCH_PARENT_TORCH') == '1'}))
        return

    # rank 0 prints ONE JSON line: from here on file descriptor 1 is stderr for everything else in the process (a
    # collective library greeting its peers, a runtime warning), and the line goes out through the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from duet_amd import _lib

    # DUET_BENCH_ONE_GPU=1 (plumbing test on a 1-GPU box only): every rank uses device 0 and the collective
    # goes through gloo instead of RCCL; never set for a measurement.
    one_gpu = os.environ.get('DUET_BENCH_ONE_GPU') == '1'
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist_mod, coll, fallback = None, None, None
    # DUET_BENCH_RCCL_SELF=1 (a check on a 1-GPU box, never a measurement): the N > 1 code path as ONE rank over real RCCL --
    # communicator set-up, the asynchronous all-gather on its own stream beside the kernels' raw stream, barriers, reductions --
    # on the sharded configs[2] problem
    rccl_self = world == 1 and os.environ.get('DUET_BENCH_RCCL_SELF') == '1'
    ctx = _lib.Context(local_rank)
    if world > 1 or rccl_self:
        import datetime
        import torch.distributed as dist_mod
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rccl_self:
            from duet_amd import launch as _launch
            os.environ.setdefault('MASTER_PORT', str(_launch.free_port()))
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        limit = float(os.environ.get('DUET_RDZV_TIMEOUT', '300'))
        if one_gpu:
            dist_mod.init_process_group('gloo', timeout=datetime.timedelta(seconds=limit))
            coll = TorchColl(torch, dist_mod, 'gloo', local_rank)
        elif args.collective == 'torch':
            dist_mod.init_process_group('nccl', device_id=torch.device('cuda', local_rank), timeout=datetime.timedelta(seconds=limit))
            coll = TorchColl(torch, dist_mod, 'nccl', local_rank)
        else:
            # default: the collective the product's ranks use (duet_comm_* inside libduet_ef.so); gloo carries the control plane
            dist_mod.init_process_group('gloo', timeout=datetime.timedelta(seconds=limit))
            err = None
            try:
                coll = DuetColl(torch, dist_mod, ctx, timeout=limit)
            except Exception as e:                       # (RCCL missing, communicator set-up failed or timed out)
                err = '%s: %s' % (type(e).__name__, e)
            errs = [None] * world
            dist_mod.all_gather_object(errs, err)
            if any(errs):
                # every rank agrees to step down to torch.distributed "nccl" -- said on the line, never silently
                fallback = [e for e in errs if e][0]
                if coll is not None:
                    coll.close()
                grp = dist_mod.new_group(backend='nccl', timeout=datetime.timedelta(seconds=limit))
                coll = TorchColl(torch, dist_mod, 'nccl', local_rank, group=grp)
                coll.name += ' (duet_comm_* unavailable: %s)' % fallback[:200]

    if world == 1 and not rccl_self:
        out = single_gpu_run(args, ctx, torch)
    else:
        out = sharded_run(args, ctx, torch, coll, rank, world, local_rank, one_gpu)
        if not args.no_extra:
            weak = weak_grouped_run(args, ctx, torch, coll, rank, world, local_rank)
            fused = fused_sharded_run(args, ctx, torch, coll, rank, world, local_rank)
            if rank == 0:
                out['extra'] = {'weak_grouped_config2_per_rank': weak, 'fused_clustered_and_phased_sharded': fused}
    if rank == 0:
        path = write_detail(out, world)
        os.write(json_fd, (json.dumps(compact_line(out, path)) + '\n').encode())
    os.close(json_fd)

    if coll is not None:
        coll.barrier()
        coll.close()
    if dist_mod is not None:
        dist_mod.destroy_process_group()
    ctx.close()
    if fallback and 'did not finish within' in fallback:
        os._exit(0)                                      # a helper thread may still sit inside RCCL's set-up: no interpreter shutdown


def ef_point(ctx, torch, DeviceProblem, soa, steps):
    """One E/F problem resident in HBM: wall time per step of `steps` back-to-back steps, then every kernel's own duration
    (HIP events on its dispatch), parity against the C oracle."""
    dp = DeviceProblem(soa, 50, 2)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        dp.run(ctx, stream)
    ctx.check(stream)
    torch.cuda.synchronize()
    n = max(20, min(int(steps), 200))
    t0 = time.perf_counter()
    for _ in range(n):
        dp.run(ctx, stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    ctx.set_profiling(2)
    ctx.profile_collect()
    for _ in range(20):
        dp.run(ctx, stream)
    torch.cuda.synchronize()
    prof = ctx.profile_collect()
    ctx.set_profiling(0)
    ctx.check(stream)
    pred, ps = dp.results()
    rc, want_pred, want_ps = cpu_leg().ef(soa, 50, 2)
    ok = bool(rc == 0 and np.array_equal(pred, want_pred) and np.array_equal(ps, want_ps))
    del dp
    torch.cuda.empty_cache()
    return {'marks': soa.n_marks, 'candidates': soa.n_cands, 'reads': soa.n_reads, 'ms_per_step': dt * 1e3, 'marks_per_s': soa.n_marks / dt,
            'parity_vs_oracle': ok,
            'kernels_us': {k: round(float(prof.kernel_ms[i]) * 1e3, 2) for i, k in enumerate(('ef_classify', 'ef_seed_sort', 'ef_finalize'))}}


def literal_8d_point(ctx, torch, engine, synth, DeviceProblem, steps):
    """BASELINE configs[1] on SURVEY 8d's generator TO THE LETTER (20 % of the marks' names without a SAM line, pc = floor(Exp(600));
    the headline's default generator: 5 % and a geometric PC) -- the reference's sha256 for these inputs is
    tests/golden/seeded_r5.json (tests/test_gpu_r5.py runs the text path against it)."""
    soa = engine.soa_from_synth([synth.bench_contig('1', 200000, 100000, 1, spelled='chr1', literal_8d=True)])
    out = ef_point(ctx, torch, DeviceProblem, soa, steps)
    out['marks_absent_share'] = float((soa.mark_read == engine.MARK_ABSENT).mean())
    out['reference_sha256_pin'] = 'tests/golden/seeded_r5.json'
    return out


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


def e2e_config3(ctx, torch, kernels_ms=None):
    """BASELINE configs[2] end to end on ONE GPU: the 24-contig, 2e7-mark genome as a 564 MB caller VCF + 24
    haplotagged BAMs on disk -> phased_sv.vcf on disk, native host path, rows formatted on the device, -t 8 and -t 4; the output's
    sha256 against the ONE run of the unmodified reference recorded in tests/golden/seeded_r2.json (272 s there)."""
    import hashlib
    import shutil
    import tempfile
    from duet_amd import synth
    from duet_amd.sv_phasing import sv_phasing
    want = None
    try:
        with open(os.path.join(REPO, 'tests', 'golden', 'seeded_r2.json')) as f:
            want = [x for x in json.load(f) if x.get('kind') == 'config3'][0]
    except (OSError, ValueError, IndexError):
        pass
    seed = want['seed'] if want else 3
    home = tempfile.mkdtemp(prefix='duet_e2e3_')
    try:
        t0 = time.perf_counter()
        contigs = synth.bench_genome(20000000, seed)
        marks = int(sum(len(c.mark_name_id) for c in contigs))
        synth.write_workdir(home, contigs, dialect='cutesv', seed=seed, write_sam=False)
        del contigs
        t_gen = time.perf_counter() - t0
        out = {}
        sv_phasing(home, 50, 2, 8, False)                   # (warm: page cache, the context's buffers)
        with open(os.path.join(home, 'phased_sv.vcf'), 'rb') as f:
            sha = hashlib.sha256(f.read()).hexdigest()
        for T in (8, 4):
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter()
                sv_phasing(home, 50, 2, T, False)
                best = min(best, time.perf_counter() - t0)
            out['t_e2e_ms_t%d' % T] = best * 1e3
        size = os.path.getsize(os.path.join(home, 'phased_sv.vcf'))
        vcf_bytes = os.path.getsize(os.path.join(home, 'sv_calling', 'variants.vcf'))
    finally:
        shutil.rmtree(home, ignore_errors=True)
    M = marks or 20000000
    out.update({'workload': 'BASELINE configs[2] as text: 24 contigs, %d marks, caller VCF %d bytes + 24 BAMs -> phased_sv.vcf %d bytes' % (M, vcf_bytes, size),
                'marks': M, 't_e2e_ms': out['t_e2e_ms_t8'], 'marks_per_s': M / (out['t_e2e_ms_t8'] * 1e-3),
                'marks_per_s_t4': M / (out['t_e2e_ms_t4'] * 1e-3), 't_kernels_ms': kernels_ms,
                'sha256_matches_reference': bool(want and sha == want['output_sha256']), 'inputs_generated_in_s': t_gen,
                'reference_python_seconds_dev_container': want.get('reference_seconds') if want else None})
    return out


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
        # ... with -t as a hard bound on the ingest's workers (default: the VCF's first half runs beside the BAM loop, each
        # with `-t` workers -- up to 2 x 4 alive at once)
        os.environ['DUET_INGEST_STRICT_THREADS'] = '1'
        try:
            sv_phasing(home, 50, 2, 4, False)
            t0 = time.perf_counter()
            for _ in range(3):
                sv_phasing(home, 50, 2, 4, False)
            t_e2e_strict = (time.perf_counter() - t0) / 3
        finally:
            del os.environ['DUET_INGEST_STRICT_THREADS']
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
            't_e2e_ms_strict_threads': t_e2e_strict * 1e3, 'marks_per_s_e2e_strict_threads': M / t_e2e_strict,
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
            'note': 'latency-bound at this size: ~20 small launches; the partitions with the most groups left are on the critical path'}


def fused_point(ctx, torch, engine, synth, contigs, runs=20, scan_order=False, tailed=None):
    """The metric read literally -- marks clustered AND phased: duet_svim_phase_device on raw shuffled marks
    (A0 sort + linkage + emit, adapter, E/F) resident in HBM, checked against the two C oracles composed."""
    from duet_amd.devmem import DeviceSvim
    c_oracle = cpu_leg()
    soa = engine.soa_from_synth(contigs)
    marks = synth.raw_marks(contigs, 1, reads_of=soa, scan_order=scan_order)
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
    n_found = int(ds.n_found)                           # (what the DEVICE found, not the oracle's count: ADVICE round 5)
    if tailed is not None:
        # step E/F ALONE on exactly the candidates stage A0 found (sizes with a tail: a real caller's shape; the bench's own E/F
        # problems have sizes U{2..18}) -- the honest figure for ef_classify beside the uniform-size roofline points
        from duet_amd.devmem import DeviceProblem
        deg = np.diff(ref.cand_off.astype(np.int64))
        pad = (-len(deg)) % 64
        mx = np.concatenate([deg, np.zeros(pad, dtype=deg.dtype)]).reshape(-1, 64).max(axis=1)
        del ds
        torch.cuda.empty_cache()
        t = ef_point(ctx, torch, DeviceProblem, ref, 20)
        t.update({'degree_mean': float(deg.mean()), 'degree_p99': int(np.percentile(deg, 99)), 'degree_max': int(deg.max()),
                  'lane_efficiency': float(deg.sum() / (64.0 * mx.sum())), 'ef_classify_us': t['kernels_us']['ef_classify'],
                  'ef_classify_GBs_algorithmic': classify_bytes(ref) / (t['kernels_us']['ef_classify'] * 1e-6) / 1e9})
        tailed.update(t)
    M = len(marks['pos'])
    b_a0, b_ef = 18 * M, 12 * M + 27 * n_found + 8 * soa.n_reads
    gbs = (b_a0 + b_ef) / dt / 1e9
    traffic, per_kernel, tsrc = fused_traffic(M)
    return {'marks': M, 'candidates_found': n_found, 'phased': int((got['pred'] != 0).sum()),
            'ms_per_run': dt * 1e3, 'marks_per_s': M / dt, 'ms_per_run_with_count_returned': dt_wait * 1e3,
            'parity_vs_composed_oracles': ok,
            'roofline': {'kernels': 'duet_svim_phase_device: A0 (sort, partitions, linkage, emit) + E/F, ~25 launches',
                         'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_source': tsrc, 'traffic_per_kernel_bytes': per_kernel,
                         'real_hbm_frac': (traffic / dt / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         'algorithmic_bytes_per_run': b_a0 + b_ef, 'B_A0_18_per_mark': b_a0, 'B_EF': b_ef,
                         'run_ms': dt * 1e3,
                         'note': 'whole pipeline: (B_A0 + B_EF) / wall time per run, back-to-back runs resident in HBM; the '
                                 'sort passes and the pair distances of the agglomeration are not credited (SURVEY 8d)'},
            'mark_order': ('as a scan of coordinate-sorted BAMs emits them (contig by contig, read by read)' if scan_order
                           else 'shuffled over the whole genome (worst case for the gathers through the sort permutation)'),
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
                     'roofline': roofline_block('ef_classify', soa, kms, int(prof.n_profiled_runs),
                                                'events on every launch of %d back-to-back steps' % n, workload=name),
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
