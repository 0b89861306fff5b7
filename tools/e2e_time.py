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

# stage by stage (same inputs, fresh work dir)
import ctypes
from duet_amd import engine
from duet_amd.native import NativeIngest, load
from duet_amd.read_file import init_chrom_list
home = tempfile.mkdtemp(prefix='duet_e2e_')
try:
    synth.write_workdir(home, [synth.bench_contig('1', 200000, 100000, 1)], dialect='cutesv', seed=1, write_sam=False)
    chroms = init_chrom_list(False, home)
    lib = load()
    ctx = engine.default_context()
    best = {}
    for rep in range(5):
        t = [time.perf_counter()]
        names = (ctypes.c_char_p * len(chroms))(*[c.encode() for c in chroms])
        h = lib.duet_ingest_create(len(chroms), names)
        lib.duet_ingest_add_bam(h, 0, (home + '/snp_phasing/chr1.bam').encode(), 4); t.append(time.perf_counter())
        lib.duet_ingest_parse_vcf(h, (home + '/sv_calling/variants.vcf').encode(), 4); t.append(time.perf_counter())
        lib.duet_ingest_destroy(h); t.append(time.perf_counter())
        ing = NativeIngest.load(home + '/sv_calling/variants.vcf', home + '/snp_phasing/', chroms, 4); t.append(time.perf_counter())
        rows = ing.rows(); t.append(time.perf_counter())
        body = ctx.ef_rows_host(ing.soa, rows, 50, 2)[0]; t.append(time.perf_counter())
        head = ing.header(False); t.append(time.perf_counter())
        with open(home + '/phased_sv.vcf', 'wb') as f:
            f.write(head + body)
        t.append(time.perf_counter())
        ing.close(); t.append(time.perf_counter())
        for name, d in zip(('bam', 'vcf', 'destroy', 'load(bam+vcf+views)', 'rows()', 'ef_rows_host', 'header', 'write', 'close'),
                           [b - a for a, b in zip(t, t[1:])]):
            best[name] = min(best.get(name, 1e9), d)
    print('  '.join('%s %.1f' % (k, v * 1e3) for k, v in best.items()), '(ms, best of 5)')
finally:
    shutil.rmtree(home, ignore_errors=True)
