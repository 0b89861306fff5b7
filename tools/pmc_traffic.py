#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from a round's counter passes and kernel-trace summaries in profiles/ (what bench.py's
`roofline.traffic` and README's table quote): per workload the FETCH_SIZE / WRITE_SIZE means per ef_classify launch, corrected
as tools/pmc_summary.py documents, beside the kernel-trace average and the algorithmic bytes 12 M + 22 C + 8 R of DESIGN.md.

    python3 tools/pmc_traffic.py r03 [collection date, e.g. 2026-10-03T14:10Z] > profiles/r03_pmc_traffic.json
"""
import csv, json, os, sys

ROUND = sys.argv[1] if len(sys.argv) > 1 else 'r03'
COLLECTED = sys.argv[2] if len(sys.argv) > 2 else None

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles')
WORK = [
    ('bench_config2', 'BASELINE configs[1]: 1 contig, 1001116 marks / 100000 candidates / 159991 tagged reads '
     '(python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra)', 1001116, 100000, 159991),
    ('ef_2e7', 'BASELINE configs[2] on one GPU: 24 contigs, 19999876 marks / 2000001 candidates / 3199468 tagged reads '
     '(python3 tools/prof_ef.py 20000000)', 19999876, 2000001, 3199468),
    ('ef_2e8', '24 contigs, 200002488 marks / 20000002 candidates / 31996306 tagged reads (python3 tools/prof_ef.py 200000000)',
     200002488, 20000002, 31996306),
]


def mean_counter(path, counter):
    per = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if 'ef_classify' in r['Kernel_Name'] and r['Counter_Name'] == counter:
                per[r['Dispatch_Id']] = per.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
    return (sum(per.values()) / len(per), len(per)) if per else (None, 0)


def trace_avg(path):
    with open(path) as f:
        for r in csv.DictReader(f):
            if 'ef_classify' in r['Name']:
                return float(r['AverageNs']) / 1e3, int(r['Calls'])
    return None, 0


out = []
for name, text, M, C, R in WORK:
    files = ['%s_%s_pmc_FETCH_SIZE.csv' % (ROUND, name), '%s_%s_pmc_WRITE_SIZE.csv' % (ROUND, name), '%s_%s_kernel_stats.csv' % (ROUND, name)]
    f_kb, nf = mean_counter(os.path.join(ROOT, files[0]), 'FETCH_SIZE')
    w_kb, nw = mean_counter(os.path.join(ROOT, files[1]), 'WRITE_SIZE')
    us, calls = trace_avg(os.path.join(ROOT, files[2]))
    traffic = int(round((2 * f_kb + w_kb) * 1024))
    alg = 12 * M + 22 * C + 8 * R
    out.append({'kernel': 'ef_classify', 'workload': text, 'collected': COLLECTED, 'FETCH_SIZE_KB_per_launch': f_kb, 'WRITE_SIZE_KB_per_launch': w_kb,
                'launches_counted': [nf, nw],
                'correction': 'counters are KB; FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads), WRITE_SIZE exact',
                'traffic_bytes_per_launch': traffic, 'kernel_trace_avg_us': us, 'kernel_trace_calls': calls, 'files': files,
                'algorithmic_bytes_per_launch': alg, 'traffic_over_algorithmic': round(traffic / alg, 3),
                'algorithmic_GBs_at_kernel_trace_avg': round(alg / us / 1e3, 1), 'traffic_GBs_at_kernel_trace_avg': round(traffic / us / 1e3, 1),
                'note': 'traffic is below the algorithmic figure because the latter charges 8 B per gathered tag (SURVEY 8d) while the '
                        'distinct tag table (8 B per read) stays in L2 / Infinity Cache between its ~6 uses'})
print(json.dumps(out, indent=1))
