#!/usr/bin/env python3
"""GPU-side one-off stress: many random E/F and A0 problems against the C oracles (bit-exact)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from duet_amd import _lib, synth
from oracle import c_oracle
from tests import soa_fuzz
from tests.test_gpu_cluster import random_marks, sv_like_marks, FIELDS

n_ef = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_cl = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = _lib.Context(0)
rng = synth.SplitMix(20261002)
t0 = time.time()
bad = 0
for i in range(n_ef):
    k = dict(n_contigs=1 + rng.one(6), cands_per_contig=(0, 50 + rng.one(3000)), reads_per_contig=(2, 10 + rng.one(1500)),
             n_ps=(1, 1 + rng.one(12)), deg=(1, 1 + rng.one(40)), big_deg=(5000 if rng.one(25) == 0 else 0),
             empty_contig_rate=rng.one(6), no_seed_contig_rate=rng.one(6), ps_spread=50 + rng.one(100000),
             sorted_pos=bool(rng.one(2)), absent_rate=rng.one(8))
    k['cands_per_contig'] = (min(k['cands_per_contig']), max(k['cands_per_contig']))
    soa = soa_fuzz.random_soa(5000 + i, **k)
    if soa.n_cands == 0:
        continue
    sl, sr = [(50, 2), (0, 0), (52, 4), (45, 1)][rng.one(4)]
    rc, wp, ws = c_oracle.ef(soa, sl, sr)
    if rc:
        continue
    # (E/F paths in turn: default = two launches, every finalize tile with its own seed set; that with a seed set of 8 entries -- the
    # array-free walk; three launches; ef_finalize with two / four tiles per workgroup; the seed sort without its hash set)
    ctx.set_debug([0, 0x3000000, 0x800000, 0x20, 0x80, 0x40][i % 6])
    p, s = ctx.run_host(soa, sl, sr)
    ctx.set_debug(0)
    if not (np.array_equal(p, wp) and np.array_equal(s, ws)):
        bad += 1
        print('E/F MISMATCH seed', 5000 + i, k)
print('E/F: %d problems, %d mismatches, %.1f s' % (n_ef, bad, time.time() - t0))
t0 = time.time()
badc = 0
for i in range(n_cl):
    M = 1 + rng.one(20000)
    if i % 2:
        marks = sv_like_marks(9000 + i, 1 + rng.one(2500))
        M = len(marks['pos'])
    else:
        marks = random_marks(9000 + i, M, clumps=1 + rng.one(60), contigs=1 + rng.one(4), types=1 + rng.one(4),
                             spread=1 + rng.one(1500), span_lo=1 + rng.one(100), span_hi=200 + rng.one(5000))
    if i % 3 == 2:                                   # genome-sized coordinates: the low key bits are sorted locally
        marks = dict(marks, pos=(marks['pos'].astype(np.uint64) + 150000000 + 1000 * (i % 977)).astype(np.uint32))
    kw = dict(max_dist=[0.1, 0.3, 0.9, 1.5, 0.5, 0.7][rng.one(6)], part_gap=[10, 1000, 5000][rng.one(3)], part_max=[7, 100, 128][rng.one(3)])
    want = c_oracle.cluster(marks['contig'], marks['type'], marks['pos'], marks['span'], **kw)
    # (launch structures and paths in turn: default, everything through the exact linkage, large-input structure, no box test)
    ctx.set_debug([0, 0x100, 0x200, 0x800, 0xA00, 0x300, 0x1000, 0x1800, 0x1A00, 0x2000, 0x3800, 0x4000, 0x8000, 0x10000, 0x10200, 0x4200, 0x40000, 0x20000, 0x20200, 0x200, 0x40100, 0x40800, 0x44000, 0x60000, 0x4000000, 0x4040000, 0x8000000, 0x8000800, 0x10000000, 0x10000800, 0x14000000, 0x10040000][i % 32])
    got = ctx.cluster_host(marks['contig'], marks['type'], marks['pos'], marks['span'], **kw)
    ctx.set_debug(0)
    if any(got[f].shape != want[f].shape or not np.array_equal(got[f], want[f]) for f in FIELDS):
        badc += 1
        print('A0 MISMATCH seed', 9000 + i, M, kw)
print('A0: %d problems, %d mismatches, %.1f s' % (n_cl, badc, time.time() - t0))
sys.exit(1 if bad or badc else 0)
