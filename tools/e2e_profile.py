#!/usr/bin/env python3
"""GPU-side diagnostic: where the end-to-end time of config 2 goes (the steps of duet_amd.sv_phasing._native, timed one by one)."""
import os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from duet_amd import engine, synth
from duet_amd.native import NativeIngest
from duet_amd.read_file import init_chrom_list

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
M = int(float(sys.argv[2])) if len(sys.argv) > 2 else 0           # marks: 0 = BASELINE configs[1]; else the 24-contig genome of that size
home = tempfile.mkdtemp(prefix='duet_e2e_')
try:
    contigs = [synth.bench_contig('1', 200000, 100000, 1)] if not M else synth.bench_genome(M, 3)
    synth.write_workdir(home, contigs, dialect='cutesv', seed=1, write_sam=False)
    del contigs
    ctx = engine.default_context(0)
    vcf, out = home + '/sv_calling/variants.vcf', home + '/phased_sv.vcf'
    for rep in range(4):
        laps = []
        t = time.perf_counter()
        def lap(what):
            global t
            now = time.perf_counter()
            laps.append((what, (now - t) * 1e3))
            t = now
        chroms = init_chrom_list(False, home)
        lap('chrom list')
        os.environ['DUET_INGEST_TIMING'] = '1' if rep == 3 else ''
        if not os.environ['DUET_INGEST_TIMING']:
            del os.environ['DUET_INGEST_TIMING']
        ing = NativeIngest.load(vcf, home + '/snp_phasing/', chroms, T)
        lap('NativeIngest.load (BAM + VCF)')
        with open(out, 'wb') as f:
            f.write(ing.header(False))
        lap('header')
        ing.log_lines(chroms)
        lap('log lines')
        rows = ing.rows()
        lap('rows()')
        body = ctx.ef_rows_host(ing.soa, rows, 50, 2)[0]
        lap('ef_rows_host (H2D, E/F, rows, D2H)')
        ing.close()
        lap('close')
        with open(out, 'ab') as f:
            f.write(body)
        lap('append %d bytes' % len(body))
        if rep == 3:
            for w, ms in laps:
                print('%-40s %7.2f ms' % (w, ms))
            print('%-40s %7.2f ms' % ('total', sum(ms for _, ms in laps)))
finally:
    shutil.rmtree(home, ignore_errors=True)
