#!/usr/bin/env python
"""Print per-kernel PMC sums from a rocprofv3 rocpd SQLite result (counters_collection view).
    python tools/pmc_summary.py results.db [kernel-name-substring]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
pat = f'%{sys.argv[2]}%' if len(sys.argv) > 2 else '%'
cur = con.cursor()
rows = cur.execute('select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection '
                   'where kernel_name like ? group by kernel_name, counter_name', (pat,)).fetchall()
for name, ctr, tot, n in rows:
    print(f'{name[:60]:60s} {ctr:32s} total={tot:.6g} dispatches={n} per_dispatch={tot / max(n, 1):.6g}')
for name, avg, n in cur.execute('select name, avg(duration), count(*) from kernels where name like ? group by name', (pat,)):
    print(f'{name[:60]:60s} avg_duration_ns={avg:.1f} launches={n}')
