#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs of the same command) to HBM bytes per launch of one
kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes: the counters are in KB; on gfx950
FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is exact.

    python3 tools/pmc_summary.py <kernel-substring> <workload text> <fetch.csv> <write.csv> [<kernel_stats.csv>]
prints one JSON object (append it to profiles/r02_pmc_traffic.json's list).
"""
import csv, json, sys


def mean_counter(path, kernel, counter):
    vals = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel in r['Kernel_Name'] and r['Counter_Name'] == counter:
                vals.append(float(r['Counter_Value']))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def stats_avg_us(path, kernel):
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel in r['Name']:
                return float(r['AverageNs']) / 1e3, int(r['Calls'])
    return None, 0


kernel, workload, fetch_csv, write_csv = sys.argv[1:5]
f_kb, nf = mean_counter(fetch_csv, kernel, 'FETCH_SIZE')
w_kb, nw = mean_counter(write_csv, kernel, 'WRITE_SIZE')
out = {'kernel': kernel, 'workload': workload, 'FETCH_SIZE_KB_per_launch': f_kb, 'WRITE_SIZE_KB_per_launch': w_kb,
       'launches_counted': [nf, nw],
       'correction': 'counters are KB; FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads), WRITE_SIZE exact',
       'traffic_bytes_per_launch': int(round((2 * f_kb + w_kb) * 1024)) if f_kb is not None and w_kb is not None else None}
if len(sys.argv) > 5:
    us, calls = stats_avg_us(sys.argv[5], kernel)
    out['kernel_trace_avg_us'] = us
    out['kernel_trace_calls'] = calls
print(json.dumps(out))
