#!/usr/bin/env python3
"""GPU-side: step E/F as two launches (ef_classify + ef_finalize_own: every finalize tile builds its contig's seed set itself) against the
three launches of rounds 1-5, over problem sizes -- where the two launches stop paying sets kOwnMaxTiles (duet_ef.hip).

    python3 tools/own_sweep.py [steps=200]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from duet_amd import _lib, engine, synth
from duet_amd.devmem import DeviceProblem

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
OWN_OFF, OWN_ALL = 0x800000, 0x1000000
ctx = _lib.Context(0)
cases = [('config2 1 contig 1.0e6', [synth.bench_contig('1', 200000, 100000, 1, spelled='chr1')])]
for m in (250000, 1000000, 2000000, 4000000, 8000000, 20000000):
    cases.append(('genome 24 contigs %.1e' % m, synth.bench_genome(m, 3)))
for name, contigs in cases:
    soa = engine.soa_from_synth(contigs)
    del contigs
    dp = DeviceProblem(soa, 50, 2)
    line = '%-28s %8d cand %5d tiles:' % (name, soa.n_cands, (soa.n_cands + 255) // 256)
    with torch.cuda.stream(torch.cuda.Stream()):
        st = torch.cuda.current_stream().cuda_stream
        ref = None
        for label, dbg in (('three', OWN_OFF), ('two', OWN_ALL)):
            ctx.set_debug(dbg)
            for _ in range(5):
                dp.run(ctx, st)
            ctx.check(st)
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                for _ in range(steps):
                    dp.run(ctx, st)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / steps)
            ctx.set_profiling(2)
            for _ in range(50):
                dp.run(ctx, st)
            torch.cuda.synchronize()
            iso = ctx.profile_collect()
            ctx.set_profiling(0)
            pred, ps = dp.results(0)
            if ref is None:
                ref = (pred.copy(), ps.copy())
            same = bool(np.array_equal(pred, ref[0]) and np.array_equal(ps, ref[1]))
            line += '  %s %.2f us (kernels %s)%s' % (label, best * 1e6, ' / '.join('%.1f' % (x * 1e3) for x in iso.kernel_ms), '' if same else ' RESULTS DIFFER')
        ctx.set_debug(0)
    print(line, flush=True)
    del dp
    torch.cuda.empty_cache()
