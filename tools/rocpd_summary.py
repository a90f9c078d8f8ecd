#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.x) rocpd SQLite result: per-kernel stats (the `--stats` view) as CSV
and, optionally, the per-launch trace of kernels matching a substring.

    python tools/rocpd_summary.py gpurun_out/prof/r01_results.db profiles/r01_kernel_stats.csv \
        [--trace sparse_conv profiles/r01_conv_trace.csv]
"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = cur.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                       'from kernels group by name order by sum(duration) desc').fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100.0 * r[2] / total, 3), r[4], r[5]])
    if '--trace' in sys.argv:
        i = sys.argv.index('--trace')
        pat, tout = sys.argv[i + 1], sys.argv[i + 2]
        rows = cur.execute('select name, start, duration, grid_x, workgroup_x, lds_size, vgpr_count, '
                           'accum_vgpr_count from kernels where name like ? order by start', (f'%{pat}%',)).fetchall()
        with open(tout, 'w', newline='') as f:
            w = csv.writer(f)
            w.writerow(['Name', 'StartNs', 'DurationNs', 'GridX', 'WorkgroupX', 'LdsBytes', 'VGPR', 'AGPR'])
            w.writerows(rows)


if __name__ == '__main__':
    main()
