#!/usr/bin/env python3
"""GPU-side diagnostic: the fused SVIM-mode pipeline a few times (for rocprofv3 --kernel-trace)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from duet_amd import _lib, engine, synth
from duet_amd.devmem import DeviceSvim
wait = 'wait' in sys.argv[1:]
big = 'big' in sys.argv[1:]          # 2e7 marks over 24 contigs instead of config 2's 1e6
ctx = _lib.Context(0)
for a in sys.argv[1:]:
    if a.startswith('dbg='):
        ctx.set_debug(int(a[4:], 0))      # DUET_DBG_* bits of include/duet_ef.h, e.g. dbg=0x200
n_marks = [int(float(a[6:])) for a in sys.argv[1:] if a.startswith('marks=')]      # marks=N: the 24-contig genome at another size
contigs = synth.bench_genome(n_marks[0], 3) if n_marks else (synth.bench_genome(20000000, 3) if big else [synth.bench_contig('1', 200000, 100000, 1)])
soa = engine.soa_from_synth(contigs)
marks = synth.raw_marks(contigs, 1, reads_of=soa, scan_order='scan' in sys.argv[1:])
depth, depth_off = synth.depth_bins(contigs, 1000, 1)
ds = DeviceSvim(marks, soa.read_tag, depth, depth_off, 1000, 50, 2)
for _ in range(3):
    ds.run_fused(ctx, wait=wait)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    ds.run_fused(ctx, wait=wait)
torch.cuda.synchronize()
print('fused ms/run', (time.perf_counter() - t0) / 10 * 1e3, 'wait' if wait else 'async')
