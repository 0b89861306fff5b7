#!/usr/bin/env python3
"""Timeline of one run in a rocprofv3 --kernel-trace CSV: every dispatch from an occurrence of the anchor kernel to the next,
with its start offset, duration and queue -- shows which stream is the critical path.  The run shown is the SHORTEST of
the trace (under the profiler the host sometimes queues a run's launches late and the device idles between them; the
shortest run is the one the device was kept busy in); --last takes the last one instead.

    python3 tools/timeline.py <dir-or-kernel_trace.csv> [anchor-substring=cl_keys] [--stats] [--last]
"""
import csv
import glob
import os
import sys


def find_csv(path):
    if os.path.isfile(path):
        return path
    hits = sorted(glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True))
    if not hits:
        sys.exit('no *kernel_trace.csv under ' + path)
    return hits[-1]


def short(name):
    name = name.replace('(anonymous namespace)::', '')
    cut = name.find('(')
    name = name if cut < 0 else name[:cut]
    return name.replace('void ', '')[:64]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    path = find_csv(args[0])
    anchor = args[1] if len(args) > 1 else 'cl_keys'
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'),
                         r.get('VGPR_Count', ''), r.get('LDS_Block_Size', ''), r.get('Grid_Size', ''), r.get('Workgroup_Size', '')))
    rows.sort()
    if '--stats' in sys.argv:
        agg = {}
        for s, e, n, *_ in rows:
            a = agg.setdefault(short(n), [0, 0])
            a[0] += 1
            a[1] += e - s
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print('%-64s n=%6d avg=%9.2f us total=%10.1f us' % (n, c, t / c / 1e3, t / 1e3))
        return
    starts = [i for i, r in enumerate(rows) if anchor in r[2]]
    if not starts:
        sys.exit('no dispatch of ' + anchor)
    runs = [(a, b) for a, b in zip(starts, starts[1:] + [len(rows)])]
    span = lambda ab: max(r[1] for r in rows[ab[0]:ab[1]]) - rows[ab[0]][0]
    last, stop = runs[-1] if '--last' in sys.argv else min(runs, key=span)
    t0 = rows[last][0]
    end = t0
    print('%-64s %9s %9s %9s  %s' % ('kernel', 'start us', 'dur us', 'end us', 'queue vgpr lds grid wg'))
    for s, e, n, q, v, l, g, w in rows[last:stop]:
        print('%-64s %9.1f %9.1f %9.1f  %s %s %s %s %s' % (short(n), (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, q, v, l, g, w))
        end = max(end, e)
    print('span %.1f us' % ((end - t0) / 1e3))


if __name__ == '__main__':
    main()
