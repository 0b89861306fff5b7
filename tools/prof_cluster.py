#!/usr/bin/env python3
"""GPU-side diagnostic: run stage A0 a few times (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from duet_amd import _lib, synth
from duet_amd.devmem import DeviceCluster
n_marks = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ctx = _lib.Context(0)
if n_marks <= 1100000:
    contigs = [synth.bench_contig('1', n_marks // 5, n_marks // 10, 1)]
else:
    contigs = synth.bench_genome(n_marks, 3)
marks = synth.raw_marks(contigs, 1)
dc = DeviceCluster(marks)
for _ in range(3):
    dc.run(ctx)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    dc.run(ctx)
torch.cuda.synchronize()
print('marks', len(marks['pos']), 'cands', dc.n_cands(), 'ms/run', (time.perf_counter() - t0) / 10 * 1e3)
