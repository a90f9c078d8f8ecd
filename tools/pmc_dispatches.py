#!/usr/bin/env python
"""Per-DISPATCH PMC values of a rocprofv3 rocpd SQLite result, in launch order (the microbenchmarks launch the same
kernel on different regions: sums per kernel name would hide what each launch moved).
    python tools/pmc_dispatches.py results.db [kernel-name-substring]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
pat = f'%{sys.argv[2]}%' if len(sys.argv) > 2 else '%'
rows = con.execute('select dispatch_id, kernel_name, counter_name, sum(value) from counters_collection where kernel_name like ? '
                   'group by dispatch_id, kernel_name, counter_name order by dispatch_id', (pat,)).fetchall()
for d, name, ctr, v in rows:
    print(f'{d:6d} {name[:40]:40s} {ctr:14s} {v:.6g}')
