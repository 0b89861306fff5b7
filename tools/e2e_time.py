#!/usr/bin/env python3
"""GPU-side diagnostic: end to end (caller VCF + haplotagged BAM on disk -> phased_sv.vcf) of config 2 per host thread count,
with the native ingest's own lap times (DUET_INGEST_TIMING)."""
import os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['DUET_INGEST_TIMING'] = '1'
from duet_amd import synth
from duet_amd.sv_phasing import sv_phasing

home = tempfile.mkdtemp(prefix='duet_e2e_')
try:
    c = synth.bench_contig('1', 200000, 100000, 1)
    synth.write_workdir(home, [c], dialect='cutesv', seed=1, write_sam=False)
    sv_phasing(home, 50, 2, 4, False)
    for T in (1, 2, 4, 8, 16, 32):
        best = 1e9
        for _ in range(3):
            sys.stderr.write('--- threads %d\n' % T)
            t0 = time.perf_counter()
            sv_phasing(home, 50, 2, T, False)
            best = min(best, time.perf_counter() - t0)
        print('threads %2d: %.1f ms end to end' % (T, best * 1e3))
finally:
    shutil.rmtree(home, ignore_errors=True)
