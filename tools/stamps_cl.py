#!/usr/bin/env python3
"""GPU-side diagnostic: where the agglomeration chains of the fused pipeline spend their time (libduet_ef_stamps.so, -DDUET_STAMPS:
a log of (tag, wall clock) per workgroup of cl_tight_big / cl_fast_all / cl_link_one).  Prints, per kernel, the workgroup that ends
last -- the kernel is as slow as its slowest chain -- with the time between consecutive tags, and the totals per tag over all
workgroups.  Not part of the product; the stamped build is a little slower.

    python3 tools/stamps_cl.py [big]
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from duet_amd import _lib, engine, synth
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libduet_ef_stamps.so')
from duet_amd.devmem import DeviceSvim

TAGS = {0x10: 'fast: unit', 0x11: 'fast: rows in LDS', 0x12: 'fast: threshold graph done', 0x13: 'fast: linkage done', 0x14: 'fast: emitted',
        0x20: 'link: start', 0x21: 'link: triangle filled', 0x22: 'link: tight groups merged', 0x23: 'link: round', 0x24: 'link: rounds over',
        0x30: 'tight: unit', 0x31: 'tight: rows in LDS', 0x32: 'tight: pair tests done', 0x33: 'tight: cliques at the threshold',
        0x34: 'tight: half / quarter levels done', 0x35: 'tight: their cliques', 0x36: 'tight: groups set up', 0x37: 'tight: cross-group sums done',
        0x38: 'tight: round', 0x39: 'tight: rounds over', 0x3A: 'tight: before emit', 0x3B: 'tight: emitted',
        0x40: 'wide: unit', 0x41: 'wide: rows in LDS', 0x42: 'wide: pair pass done', 0x43: 'wide: cliques known', 0x44: 'wide: tight groups contracted',
        0x45: 'wide: round', 0x46: 'wide: rounds over', 0x47: 'wide: emitted', 0x48: 'wide: r scanned', 0x49: 'wide: r lanes met', 0x4A: 'wide: r nn known',
        0x4B: 'wide: r merges known', 0x4C: 'wide: r contracted'}
big = 'big' in sys.argv[1:]
contigs = synth.bench_genome(20000000, 3) if big else [synth.bench_contig('1', 200000, 100000, 1)]
soa0 = engine.soa_from_synth(contigs)
marks = synth.raw_marks(contigs, 1, reads_of=soa0)
depth, depth_off = synth.depth_bins(contigs, 1000, 1)
del contigs
ctx = _lib.Context(0)
lib = _lib.load()
lib.duet_dbg_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
ds = DeviceSvim(marks, soa0.read_tag, depth, depth_off, 1000, 50, 2)
for _ in range(3):
    ds.run_fused(ctx, wait=True)
lib.duet_dbg_stamps(ctx.handle, 1, None)
ds.run_fused(ctx, wait=True)
buf = np.zeros(6 * 65536 * 8, dtype=np.uint64)
lib.duet_dbg_stamps(ctx.handle, 0, buf.ctypes.data)
for area, name in ((3, 'cl_wide_big / cl_tight_big'), (4, 'cl_fast_all (1.0 M marks) / cl_tight_one<64> (large inputs)'), (5, 'cl_wide_list / cl_link_one + cl_tight_one<32> (large inputs)')):
    log = buf[area * 65536 * 8:(area + 1) * 65536 * 8].reshape(8192, 64)[:, 2:]
    used = (log != 0).sum(axis=1)
    blocks = np.nonzero(used)[0]
    if len(blocks) == 0:
        print(name, ': no workgroup logged'); continue
    tag = (log >> np.uint64(56)).astype(np.int64)
    tim = (log & np.uint64(0x00FFFFFFFFFFFFFF)).astype(np.int64)
    t0 = min(int(tim[b, 0]) for b in blocks)
    ends = {int(b): int(tim[b, used[b] - 1]) for b in blocks}
    last = max(ends, key=ends.get)
    print('%s: %d workgroups logged (of at most 8192), first stamp to last stamp %.2f us; full logs (62 entries) in %d of them' %
          (name, len(blocks), (max(ends.values()) - t0) / 100.0, int((used == 62).sum())))
    # per-tag totals: the time from a stamp to the next one is charged to the FIRST stamp's tag
    tot, cnt = {}, {}
    for b in blocks:
        for i in range(used[b] - 1):
            tg = int(tag[b, i])
            tot[tg] = tot.get(tg, 0) + int(tim[b, i + 1] - tim[b, i]); cnt[tg] = cnt.get(tg, 0) + 1
    print('  time after each tag until the next stamp, summed over all logged workgroups (us; count):')
    for tg in sorted(tot):
        print('    %-36s %10.1f  %7d   mean %.2f' % (TAGS.get(tg, hex(tg)), tot[tg] / 100.0, cnt[tg], tot[tg] / 100.0 / cnt[tg]))
    for which, b in (('the workgroup that ends last', last),):
        print('  %s: block %d, %d stamps, starts %.2f us after the kernel\'s first stamp, ends at %.2f' %
              (which, b, used[b], (int(tim[b, 0]) - t0) / 100.0, (ends[b] - t0) / 100.0))
        prev = int(tim[b, 0]); rounds = []
        for i in range(used[b]):
            tg, t = int(tag[b, i]), int(tim[b, i])
            if tg in (0x23, 0x38, 0x45):
                rounds.append((t - t0) / 100.0)
                continue
            if rounds:
                d = np.diff(np.array(rounds + [(t - t0) / 100.0]))
                print('      %d rounds from %.2f: %s us each' % (len(rounds), rounds[0], ' '.join('%.2f' % x for x in d)))
                rounds = []
            print('    %8.2f  %s' % ((t - t0) / 100.0, TAGS.get(tg, hex(tg))))
        if rounds:
            print('      %d rounds from %.2f (log full)' % (len(rounds), rounds[0]))
