#!/usr/bin/env python3
"""GPU-side: ef_classify's threshold between its two walks (one lane per candidate, mark after mark / the whole wavefront, 64 marks
per step), swept in ONE process over the problems that matter: stage A0's own candidates (sizes with a tail: 12 marks on average,
35 at the 99th percentile, 85 at most at 2e7 marks) and the bench's E/F problems (sizes U{2..18}).  Kernel times from HIP events on
every kernel's own dispatch (duet_ctx_set_profiling(2)), the durations rocprofv3 --kernel-trace reports.

    python3 tools/sweep_heavy.py [big] [steps=20] [T=off,16,24,32,48,all] [dbg=0,0x400000,...]     dbg: DUET_DBG_EF_* bits, each with every T
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from duet_amd import _lib, engine, synth
from duet_amd.devmem import DeviceSvim, DeviceProblem

big = 'big' in sys.argv[1:]
steps = ([int(a[6:]) for a in sys.argv[1:] if a.startswith('steps=')] or [20])[0]
Ts = ([a[2:] for a in sys.argv[1:] if a.startswith('T=')] or ['off,16,24,32,48,all'])[0].split(',')
dbgs = [int(x, 0) for x in ([a[4:] for a in sys.argv[1:] if a.startswith('dbg=')] or ['0'])[0].split(',')]     # DUET_DBG_EF_* bits


def a0_problem(contigs):
    soa0 = engine.soa_from_synth(contigs)
    marks = synth.raw_marks(contigs, 1, reads_of=soa0)
    depth, depth_off = synth.depth_bins(contigs, 1000, 1)
    K = len(contigs)
    ctx = _lib.Context(0)
    ds = DeviceSvim(marks, soa0.read_tag, depth, depth_off, 1000, 50, 2)
    ds.run_fused(ctx, wait=True)
    got = ds.fetch()
    N = ds.n_found
    off = got['cand_off'].astype(np.int64)
    support = np.diff(off)
    k = got['cand_contig'].astype(np.int64)
    nb = np.diff(depth_off.astype(np.int64))[k]
    bins = np.minimum(got['cand_pos'].astype(np.int64) // 1000, np.maximum(nb - 1, 0))
    d = np.where(nb > 0, depth[np.minimum(depth_off[k].astype(np.int64) + bins, len(depth) - 1)], 0).astype(np.int64)
    soa = engine.EfSoA(cand_ctg_off=np.searchsorted(k, np.arange(K + 1)), read_tag=soa0.read_tag, cand_pos=got['cand_pos'],
                       cand_svlen=got['cand_span'], cand_svread=support, cand_refread=np.maximum(d - support, 0),
                       cand_gt_ok=np.ones(N, dtype=np.uint8), cand_off=off, mark_read=marks['read'][got['order']])
    del ds
    ctx.close()
    return soa0, soa


contigs = synth.bench_genome(20000000, 3) if big else [synth.bench_contig('1', 200000, 100000, 1)]
bench_soa, a0_soa = a0_problem(contigs)
del contigs
problems = [('stage A0 candidates', a0_soa), ('bench E/F problem', bench_soa)]
for name, soa in problems:
    deg = np.diff(soa.cand_off.astype(np.int64))
    print('%s: %d marks / %d candidates; marks per candidate mean %.1f p99 %d max %d; share above 16/24/32/48: %.2f/%.2f/%.2f/%.2f %%' % (
        name, soa.n_marks, soa.n_cands, deg.mean(), np.percentile(deg, 99), deg.max(),
        100.0 * (deg > 16).mean(), 100.0 * (deg > 24).mean(), 100.0 * (deg > 32).mean(), 100.0 * (deg > 48).mean()))
    ref = None
    for T, dbg in [(T, d) for d in dbgs for T in Ts]:
        os.environ['DUET_EF_HEAVY_T'] = {'off': str(0xFFFFFFFF), 'all': '0'}.get(T, T)
        ctx = _lib.Context(0)
        ctx.set_debug(dbg)
        dp = DeviceProblem(soa, 50, 2)
        with torch.cuda.stream(torch.cuda.Stream()):
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(3):
                dp.run(ctx, st)
            ctx.check(st)
            ctx.set_profiling(2)
            ctx.profile_collect()
            for _ in range(steps):
                dp.run(ctx, st)
            torch.cuda.synchronize()
            prof = ctx.profile_collect()
            ctx.set_profiling(0)
            ctx.check(st)
        pred, ps = dp.results()
        if ref is None:
            ref = (pred.copy(), ps.copy())
        same = bool(np.array_equal(pred, ref[0]) and np.array_equal(ps, ref[1]))
        print('  T=%-4s dbg=%#8x ef_classify %8.2f us   ef_seed_sort %7.2f   ef_finalize %7.2f   same results as the first: %s' % (
            T, dbg, prof.kernel_ms[0] * 1e3, prof.kernel_ms[1] * 1e3, prof.kernel_ms[2] * 1e3, same))
        del dp
        ctx.close()
        torch.cuda.empty_cache()
