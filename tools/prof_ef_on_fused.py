#!/usr/bin/env python3
"""GPU-side diagnostic: step E/F ALONE on the problem the fused pipeline hands it -- the candidates stage A0 finds in the raw
marks (type-major inside a contig), with the host-planned launch of duet_ef_run_device -- in the back-to-back loop of
tools/prof_ef.py.  Separates what the candidates' order and shape cost ef_classify / ef_seed_sort from what the pipeline's
device-planned launch and cold caches cost (profiles/history/r04_ef_cold_caches.txt).

    python3 tools/prof_ef_on_fused.py [big] [steps=20] [cold] [sorted]     sorted: the same candidates in (contig, position) order
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from duet_amd import _lib, engine, synth
from duet_amd.devmem import DeviceSvim, DeviceProblem

big = 'big' in sys.argv[1:]
cold = 'cold' in sys.argv[1:]
resort = 'sorted' in sys.argv[1:]
steps = [int(a[6:]) for a in sys.argv[1:] if a.startswith('steps=')]
steps = steps[0] if steps else 20
contigs = synth.bench_genome(20000000, 3) if big else [synth.bench_contig('1', 200000, 100000, 1)]
soa0 = engine.soa_from_synth(contigs)
marks = synth.raw_marks(contigs, 1, reads_of=soa0)
depth, depth_off = synth.depth_bins(contigs, 1000, 1)
K = len(contigs)
del contigs
ctx = _lib.Context(0)
ds = DeviceSvim(marks, soa0.read_tag, depth, depth_off, 1000, 50, 2)
ds.run_fused(ctx, wait=True)
got = ds.fetch()
N = ds.n_found
# the adapter rules of include/duet_ef.h (duet_svim_phase_device) in numpy, as tests/test_gpu_fused.py has them
off = got['cand_off'].astype(np.int64)
support = np.diff(off)
k = got['cand_contig'].astype(np.int64)
nb = np.diff(depth_off.astype(np.int64))[k]
bins = np.minimum(got['cand_pos'].astype(np.int64) // 1000, np.maximum(nb - 1, 0))
d = np.where(nb > 0, depth[np.minimum(depth_off[k].astype(np.int64) + bins, len(depth) - 1)], 0).astype(np.int64)
mark_read = marks['read'][got['order']]
cand_pos, cand_span = got['cand_pos'], got['cand_span']
refread = np.maximum(d - support, 0)
if resort:
    perm = np.lexsort((cand_pos, k))                        # stable: (contig, position)
    starts = off[:-1][perm]
    lens = support[perm]
    new_off = np.concatenate([[0], np.cumsum(lens)])
    idx = np.repeat(starts - new_off[:-1], lens) + np.arange(int(new_off[-1]))
    mark_read = mark_read[idx]
    cand_pos, cand_span, support, refread, k = cand_pos[perm], cand_span[perm], lens, refread[perm], k[perm]
    off = new_off
soa = engine.EfSoA(cand_ctg_off=np.searchsorted(k, np.arange(K + 1)), read_tag=soa0.read_tag, cand_pos=cand_pos, cand_svlen=cand_span,
                   cand_svread=support, cand_refread=refread, cand_gt_ok=np.ones(N, dtype=np.uint8), cand_off=off, mark_read=mark_read)
del ds
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
print('E/F on stage A0\'s candidates%s: %d marks / %d candidates, %d contigs: %.4f ms per step%s' % (
    ' in position order' if resort else ' (type-major)', soa.n_marks, soa.n_cands, soa.n_contigs, dt * 1e3, ' (cold caches)' if cold else ''))
