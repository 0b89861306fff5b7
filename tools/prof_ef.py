#!/usr/bin/env python3
"""GPU-side: step E/F alone (ef_classify -> ef_seed_sort -> ef_finalize) on one synthetic problem of a given size, resident in
HBM, a fixed number of steps -- the command rocprofv3 wraps for the per-size kernel statistics and the FETCH_SIZE / WRITE_SIZE
counter passes under profiles/.

    python3 tools/prof_ef.py <marks> [steps=20] [deg=D] [cold]     marks <= 1.1e6: BASELINE configs[1] (one contig); else the 24-contig genome
                                                            deg=D: every candidate with exactly D marks (the walk's lanes all
                                                            loop D times: what a perfectly load-balanced walk would cost)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from duet_amd import _lib, engine, synth
from duet_amd.devmem import DeviceProblem

marks = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
steps = int(sys.argv[2]) if len(sys.argv) > 2 and '=' not in sys.argv[2] else 20
deg = [int(a[4:]) for a in sys.argv[1:] if a.startswith('deg=')]
cold = 'cold' in sys.argv[1:]      # cold: 1 GiB streamed through the chip between two steps -- nothing of the problem is left in
                                   # L2 / the 256 MiB Infinity Cache, as when stage A0 has run in front (the fused pipeline)
if marks <= 1100000:
    contigs = [synth.bench_contig('1', 200000, 100000, 1, spelled='chr1')]
else:
    contigs = synth.bench_genome(marks, 3, mean_deg=deg[0], deg_lo=deg[0], deg_hi=deg[0]) if deg else synth.bench_genome(marks, 3)
soa = engine.soa_from_synth(contigs)
del contigs
ctx = _lib.Context(0)
for a in sys.argv[1:]:
    if a.startswith('dbg='):
        ctx.set_debug(int(a[4:], 0))      # DUET_DBG_EF_* bits of include/duet_ef.h, e.g. dbg=0x80
dp = DeviceProblem(soa, 50, 2)
with torch.cuda.stream(torch.cuda.Stream()):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        dp.run(ctx, st)
    torch.cuda.synchronize()
    flush = torch.empty(1 << 28, dtype=torch.int32, device='cuda') if cold else None
    t0 = time.perf_counter()
    for _ in range(steps):
        if cold:
            flush.add_(1)
        dp.run(ctx, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ctx.check(st)
print('E/F: %d marks / %d candidates / %d tagged reads, %d contigs: %.4f ms per step, classify bytes %d, B_EF %d' % (
    soa.n_marks, soa.n_cands, soa.n_reads, soa.n_contigs, dt * 1e3, 12 * soa.n_marks + 22 * soa.n_cands + 8 * soa.n_reads,
    soa.algorithmic_bytes()))
