#!/usr/bin/env python3
"""GPU-side diagnostic: end to end of a 24-contig work dir (caller VCF + haplotagged BAMs on disk -> phased_sv.vcf) through the
single-process path and through `gpus` ranks in the one-GPU plumbing mode (DUET_ONE_GPU=1: every rank on device 0, gloo) --
each from a fresh interpreter, wall time of the whole call incl. interpreter start, torch import and the rendezvous.

    python3 tools/e2e_sharded.py [marks, default 4000000] [threads, default 8]
"""
import os, shutil, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from duet_amd import synth

marks = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
home = tempfile.mkdtemp(prefix='duet_e2e_sh_')
try:
    synth.write_workdir(home, synth.bench_genome(marks, 3), dialect='cutesv', seed=3, write_sam=False)
    env = dict(os.environ, PYTHONPATH=REPO, DUET_ONE_GPU='1')
    ref = None
    for gpus in (1, 2, 4, 8):
        best = 1e9
        for _ in range(2):
            if os.path.exists(home + '/phased_sv.vcf'):
                os.remove(home + '/phased_sv.vcf')
            code = ('import time\nfrom duet_amd.sv_phasing import sv_phasing\nt0 = time.perf_counter()\n'
                    'sv_phasing(%r, 50, 2, %d, False, gpus=%d)\nprint("CALL %%.4f" %% (time.perf_counter() - t0))\n' % (home, threads, gpus))
            t0 = time.perf_counter()
            out = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            wall = time.perf_counter() - t0
            assert out.returncode == 0, out.stderr.decode()[-1500:]
            call = float([l for l in out.stdout.decode().splitlines() if l.startswith('CALL')][-1].split()[1])
            best = min(best, call)
        with open(home + '/phased_sv.vcf', 'rb') as f:
            text = f.read()
        if ref is None:
            ref = text
        print('gpus %d (-t %d): sv_phasing() %.0f ms; whole process %.0f ms; output %s' % (
            gpus, threads, best * 1e3, wall * 1e3, 'identical' if text == ref else 'DIFFERS'))
finally:
    shutil.rmtree(home, ignore_errors=True)
