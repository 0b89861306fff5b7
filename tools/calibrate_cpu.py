#!/usr/bin/env python3
# coding=utf-8
"""CPU calibration of BASELINE.md section 4 (development container only: imports the reference from /root/reference).

Times, on the SAME config-2 inputs (duet_amd.synth.bench_contig('1', 200000, 100000, 1), cuteSV dialect):
  * upstream's Python: the whole sv_phasing() and, separately, its step-E/F region proper -- everything of
    generate_phased_callset after generate_callinfo has returned (filter, PS-class, seed sets, predict_hp, emission,
    sort: sv_phasing_fn.py:187-229);
  * oracle/ef_oracle.c (the scalar C restatement bench.py times as `cpu_baseline` on the GPU box) on the SoA of the
    same inputs -- the same region minus emission and sort.
Writes profiles/cpu_calibration.json: ratio = reference E/F seconds / port seconds.  bench.py multiplies nothing in;
it reports the port's measured rate on the GPU box and, beside it, that rate divided by the ratio.
"""
import gc
import json
import os
import statistics
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))

from duet_amd import engine, synth      # noqa: E402
import make_golden as G                 # noqa: E402


def main():
    if not os.path.isdir(G.REF_SRC):
        sys.exit('reference not present: development container only')
    sys.path.insert(0, G.REF_SRC)
    tmp = tempfile.mkdtemp(prefix='duet_calib_')
    G.install_shims(tmp)
    contig = synth.bench_contig('1', 200000, 100000, 1)
    home = os.path.join(tmp, 'w')
    synth.write_workdir(home, [contig], dialect='cutesv', seed=1, write_bam=False)
    soa = engine.soa_from_synth([contig])

    from duet import sv_phasing_fn as R
    from duet.sv_phasing import sv_phasing
    whole, callinfo, ef = [], [], []
    real_callinfo = R.generate_callinfo
    stamp = {}

    def timed_callinfo(*a, **kw):
        t0 = time.perf_counter()
        out = real_callinfo(*a, **kw)
        stamp['callinfo'] = time.perf_counter() - t0
        stamp['after'] = time.perf_counter()
        return out

    R.generate_callinfo = timed_callinfo
    try:
        for _ in range(3):
            gc.collect()
            t0 = time.perf_counter()
            rows = R.generate_phased_callset(home + '/sv_calling/variants.vcf', home + '/snp_phasing/', 50, 2, 4, False)
            t1 = time.perf_counter()
            callinfo.append(stamp['callinfo'])
            ef.append(t1 - stamp['after'])
            del rows
    finally:
        R.generate_callinfo = real_callinfo
    for _ in range(3):
        gc.collect()
        t0 = time.perf_counter()
        sv_phasing(home, 50, 2, 4, False)
        whole.append(time.perf_counter() - t0)

    from oracle import c_oracle
    c_oracle.ef(soa, 50, 2)
    port = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(20):
            c_oracle.ef(soa, 50, 2)
        port.append((time.perf_counter() - t0) / 20)
    model = None
    with open('/proc/cpuinfo') as f:
        for line in f:
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    ref_ef, port_s = statistics.median(ef), statistics.median(port)
    out = {
        'where': 'development container, %s, 1 core each, CPython %s' % (model, sys.version.split()[0]),
        'inputs': 'config 2: %d marks / %d candidates / %d tagged reads (synth.bench_contig seed 1, cuteSV dialect)' % (
            soa.n_marks, soa.n_cands, soa.n_reads),
        'reference_sv_phasing_whole_s': [round(x, 3) for x in whole],
        'reference_generate_callinfo_s': [round(x, 3) for x in callinfo],
        'reference_ef_region_s': [round(x, 3) for x in ef],
        'reference_ef_region_marks_per_s': soa.n_marks / ref_ef,
        'reference_whole_marks_per_s': soa.n_marks / statistics.median(whole),
        'port_ef_s': [round(x, 5) for x in port],
        'port_ef_marks_per_s': soa.n_marks / port_s,
        'ratio_reference_over_port': ref_ef / port_s,
        'region': 'reference: sv_phasing_fn.py:187-229 (after generate_callinfo returns: filter, PS-class, seed sets, '
                  'predict_hp, emission, sort); port: oracle/ef_oracle.c (filter .. decision on the SoA)',
    }
    with open(os.path.join(REPO, 'profiles', 'cpu_calibration.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
