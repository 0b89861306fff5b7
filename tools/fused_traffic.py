#!/usr/bin/env python3
"""profiles/rNN_fused_traffic.json from the FETCH_SIZE / WRITE_SIZE counter passes of the fused clustered + phased pipeline
(rocprofv3 --pmc FETCH_SIZE -- python3 tools/prof_fused.py [big], and the same with WRITE_SIZE): per kernel the HBM-side bytes
of ONE pipeline run -- FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads: /opt/skills/guides/MI355X_MICROARCH.md,
section HBM), WRITE_SIZE as it is, both in KB -- and their sum, which bench.py puts on roofline_clustered_and_phased.traffic.

    python3 tools/fused_traffic.py <marks> <fetch counter_collection.csv> <write counter_collection.csv> <collected-utc> <command text>
prints one JSON object (a list entry of profiles/rNN_fused_traffic.json).
"""
import csv, json, sys
from collections import defaultdict


def per_kernel(path, counter):
    per_dispatch = defaultdict(float)
    names = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] != counter:
                continue
            per_dispatch[r['Dispatch_Id']] += float(r['Counter_Value'])
            names[r['Dispatch_Id']] = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    tot, cnt = defaultdict(float), defaultdict(int)
    for d, v in per_dispatch.items():
        tot[names[d]] += v
        cnt[names[d]] += 1
    return tot, cnt


marks, fetch_csv, write_csv, collected, command = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5]
ft, fc = per_kernel(fetch_csv, 'FETCH_SIZE')
wt, wc = per_kernel(write_csv, 'WRITE_SIZE')
pipeline = [k for k in ft if k.split('<')[0] in ('rs_hist', 'rs_hist_dig', 'rs_offsets_small', 'rs_col_reduce', 'rs_scatter',
                                                  'rx_local', 'rx_big', 'part_reduce', 'part_spine', 'part_apply', 'cl_box', 'cl_tight_big',
                                                  'cl_link_one', 'cl_fast_all', 'cl_tight_all', 'cl_tight_one', 'cl_tier2_one', 'cl_tier2_all',
                                                  'scan_reduce', 'scan_spine', 'scan_apply', 'cl_emit', 'plan_device_contigs', 'ef_classify',
                                                  'ef_seed_sort', 'ef_finalize', 'cl_keys', 'rx_hist', 'rx_offsets', 'rx_scatter', 'cl_find_big', 'cl_wide_big', 'cl_wide_list', 'cl_signal',
                                                  'cl_gate', 'cl_gate2', 'cl_pc_sums')]
def runs_of(c):       # one first-pass histogram (record sort) or one cl_keys (key-only sort) per run
    first = [v for k, v in c.items() if k.startswith('rs_hist<true')]
    return max(1, first[0] if first else c.get('cl_keys', 1))


runs_f, runs_w = runs_of(fc), runs_of(wc)
table = {}
for k in sorted(pipeline):
    fetch_b = 2.0 * ft[k] * 1024.0 / runs_f
    write_b = wt.get(k, 0.0) * 1024.0 / runs_w
    table[k] = {'fetch_bytes': int(round(fetch_b)), 'write_bytes': int(round(write_b)), 'launches_per_run': round(fc[k] / runs_f, 2)}
total = sum(v['fetch_bytes'] + v['write_bytes'] for v in table.values())
print(json.dumps({'marks': marks, 'traffic_bytes_per_run': int(total), 'per_kernel': table, 'runs_counted': [runs_f, runs_w],
                  'collected': collected, 'command': command,
                  'correction': 'counters in KB; FETCH_SIZE doubled (gfx950), WRITE_SIZE exact; Infinity-Cache hits are counted (the counters sit on the L2 memory side)',
                  'algorithmic_bytes_per_run_for_comparison': 'B_A0 + B_EF = 18 M + 12 M + 27 C + 8 R (bench.py roofline_clustered_and_phased)'}, indent=1))
