#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc counter_collection.csv files to one table: per kernel (name cut at the first '(') the mean of every
counter over its dispatches.

    python3 tools/pmc_table.py <counter_collection.csv> [...]
"""
import csv, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        per_dispatch = defaultdict(float)            # a counter is reported per dimension (XCD / SE ...): sum them per dispatch
        names = {}
        for r in csv.DictReader(f):
            key = (r['Dispatch_Id'], r['Counter_Name'])
            per_dispatch[key] += float(r['Counter_Value'])
            names[r['Dispatch_Id']] = r['Kernel_Name']
        for (d, c), v in per_dispatch.items():
            k = names[d].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
            acc[k][c].append(v)
counters = sorted({c for k in acc for c in acc[k]})
print('kernel'.ljust(44) + ''.join(c.rjust(22) for c in counters) + '   dispatches')
for k in sorted(acc):
    row = k[:43].ljust(44)
    n = 0
    for c in counters:
        v = acc[k].get(c)
        row += (f'{sum(v) / len(v):.1f}' if v else '-').rjust(22)
        n = max(n, len(v) if v else 0)
    print(row + f'   {n}')
