#!/usr/bin/env python3
"""GPU-side diagnostic: end-to-end time of sv_phasing() on the config-2 work dir, rows on the device vs on the host."""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from duet_amd import synth
from duet_amd.sv_phasing import sv_phasing
home = tempfile.mkdtemp(prefix='duet_e2e_')
try:
    synth.write_workdir(home, [synth.bench_contig('1', 200000, 100000, 1)], dialect='cutesv', seed=1, write_sam=False)
    for mode in ('1', '0', '1', '0'):
        os.environ['DUET_DEVICE_ROWS'] = mode
        sv_phasing(home, 50, 2, 4, False)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            sv_phasing(home, 50, 2, 4, False)
            ts.append(time.perf_counter() - t0)
        print('rows on %s: min %.1f ms  median %.1f ms' % ('device' if mode == '1' else 'host  ', min(ts) * 1e3, sorted(ts)[2] * 1e3))
finally:
    shutil.rmtree(home, ignore_errors=True)
