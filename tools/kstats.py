#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel name, calls, average and total time)."""
import sqlite3, sys
for f in sys.argv[1:]:
    c = sqlite3.connect(f)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    ks = [t for t in tabs if 'kernel_symbol' in t][0]
    rows = c.execute("select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from %s d join %s s "
                     "on d.kernel_id=s.id group by s.kernel_name order by 4 desc" % (kd, ks)).fetchall()
    print(f)
    for r in rows[:24]:
        print('%-72s n=%5d avg=%10.1f us total=%10.1f us' % (r[0][:72], r[1], r[2] / 1e3, r[3] / 1e3))
