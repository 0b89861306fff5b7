#!/usr/bin/env python3
"""GPU-side diagnostic: time the three kernels under ablation flags (duet_ctx_set_debug) at two sizes."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from duet_amd import _lib, engine, synth
from duet_amd.devmem import DeviceProblem

ctx = _lib.Context(0)
lib = _lib.load()
lib.duet_ctx_set_debug.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
sizes = {'cfg2': engine.soa_from_synth([synth.bench_contig('1', 200000, 100000, 1)]),
         'cfg3': engine.soa_from_synth(synth.bench_genome(20000000, 3))}
flags = [int(x, 0) for x in (sys.argv[1:] or ['0', '1', '2', '4'])]
for name, soa in sizes.items():
    dp = DeviceProblem(soa, 50, 2)
    stream = torch.cuda.current_stream().cuda_stream
    for fl in flags:
        lib.duet_ctx_set_debug(ctx.handle, fl)
        for _ in range(5):
            dp.run(ctx, stream)
        torch.cuda.synchronize()
        res = []
        for mode in (0, 1, 2):
            ctx.set_profiling(mode); ctx.profile_collect()
            t0 = time.perf_counter()
            for _ in range(50):
                dp.run(ctx, stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 50
            st = ctx.profile_collect(); ctx.set_profiling(0)
            res.append((dt, st))
        print(name, 'dbg=%d' % fl, 'step_us(no events / classify events / all events)=%.1f %.1f %.1f' % tuple(r[0] * 1e6 for r in res),
              'k1_us(mode1)=%.1f' % (res[1][1].kernel_ms[0] * 1e3), 'k_us(mode2)=[%.1f %.1f %.1f]' % tuple(1e3 * x for x in res[2][1].kernel_ms))
    lib.duet_ctx_set_debug(ctx.handle, 0)
