#!/usr/bin/env python3
"""GPU-side diagnostic: per-block phase timeline of the three kernels (libduet_ef_stamps.so, -DDUET_STAMPS).
Not part of the product; run its numbers as SHARES, the stamped build is slower."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from duet_amd import _lib, engine, synth
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libduet_ef_stamps.so')
from duet_amd.devmem import DeviceProblem

ctx = _lib.Context(0)
if '3k' in sys.argv:
    ctx.set_debug(0x800000)        # DUET_DBG_EF_OWN_OFF: the three launches of rounds 1-5
lib = _lib.load()
lib.duet_dbg_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
which = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
if which == 'fused':
    # the E/F problem that stage A0 hands over in the fused pipeline (candidates ordered by type, then centre)
    from oracle import c_oracle
    contigs = [synth.bench_contig('1', 200000, 100000, 1)]
    base = engine.soa_from_synth(contigs)
    marks = synth.raw_marks(contigs, 1, reads_of=base)
    depth, depth_off = synth.depth_bins(contigs, 1000, 1)
    cl = c_oracle.cluster(marks['contig'], marks['type'], marks['pos'], marks['span'])
    N = len(cl['cand_pos'])
    support = np.diff(cl['cand_off'].astype(np.int64))
    k = cl['cand_contig'].astype(np.int64)
    d = depth[depth_off[k] + np.minimum(cl['cand_pos'].astype(np.int64) // 1000, np.diff(depth_off)[k] - 1)].astype(np.int64)
    soa = engine.EfSoA(cand_ctg_off=np.searchsorted(k, np.arange(2)), read_tag=base.read_tag, cand_pos=cl['cand_pos'],
                       cand_svlen=cl['cand_span'], cand_svread=support, cand_refread=np.maximum(d - support, 0),
                       cand_gt_ok=np.ones(N, dtype=np.uint8), cand_off=cl['cand_off'], mark_read=marks['read'][cl['order']])
else:
    soa = engine.soa_from_synth([synth.bench_contig('1', 200000, 100000, 1)]) if which == 'cfg2' else \
        engine.soa_from_synth(synth.bench_genome(20000000, 3))
dp = DeviceProblem(soa, 50, 2)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    dp.run(ctx, stream)
torch.cuda.synchronize()
lib.duet_dbg_stamps(ctx.handle, 1, None)
dp.run(ctx, stream)
torch.cuda.synchronize()
buf = np.zeros(6 * 65536 * 8, dtype=np.uint64)          # (areas 3..5: stage A0's chains, tools/stamps_cl.py)
lib.duet_dbg_stamps(ctx.handle, 0, buf.ctypes.data)
st = buf[:3 * 65536 * 8].reshape(3, 65536, 8).astype(np.int64)
t0 = st[0][st[0][:, 0] > 0][:, 0].min()
names = [['start', 'offsets', 'staged', 'consumed', 'loop_end', 'decided', 'end'],
         ['start', 'gathered', 'sorted', 'end', 'recs', 'counted', 'written', 'runs'],
         ['start', 'meta', 'staged'] if '3k' in sys.argv else ['start', 'cleared', 'inserted', 'swept', 'sorted64', 'ranked', 'end']]
for k in range(3):
    blk = st[k][st[k][:, 0] > 0]
    nb = len(blk)
    print('kernel', k, 'blocks stamped', nb)
    rel = (blk - t0) * 10.0 / 1000.0   # 100 MHz ticks -> us
    for i, nm in enumerate(names[k]):
        col = rel[:, i]
        col = col[blk[:, i] > 0]
        if len(col):
            print('  %-9s first %.2f  median %.2f  p90 %.2f  last %.2f us' % (nm, col.min(), np.median(col), np.percentile(col, 90), col.max()))
    if k == 1:
        order = [0, 4, 5, 6, 1, 7, 2, 3]          # stamp indices in program order
        for b in range(nb):
            print('  contig block %d:' % b, ' '.join('%s@%.2f' % (names[1][i], rel[b, i]) for i in order if blk[b, i] > 0))
        continue
    d = np.diff(rel[:, :len(names[k])], axis=1)
    print('  phase durations (median us):', ' '.join('%s=%.2f' % (names[k][i + 1], np.median(d[:, i])) for i in range(d.shape[1])))
if st[1][1][0] > 0:                         # (a diagnostic build may leave (n << 32 | distinct) of contig 0's seed list there)
    print('seed list of contig 0: n %d distinct %d' % (int(st[1][1][0]) >> 32, int(st[1][1][0]) & 0xFFFFFFFF))
