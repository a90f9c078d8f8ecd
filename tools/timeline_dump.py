#!/usr/bin/env python
"""Dump every kernel launch of a rocprofv3 rocpd result (name, start, end, queue, stream, grid, workgroup) as a gzipped CSV
-- the raw material of tools/timeline_stats.py (how much of the wall time of a multi-stream run has 0 / 1 / 2 / 3 kernels
in flight, which kernels stretch next to others).

    python tools/timeline_dump.py kt_results.db out.csv.gz
"""
import csv
import gzip
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute('PRAGMA table_info(kernels)').fetchall()]
    print('kernels view columns:', cols, file=sys.stderr)
    want = [c for c in ('name', 'start', 'end', 'duration', 'queue_id', 'stream_id', 'queue', 'stream', 'tid', 'grid_x',
                        'workgroup_x', 'lds_size', 'vgpr_count', 'accum_vgpr_count') if c in cols]
    rows = cur.execute('select %s from kernels order by start' % ', '.join(want)).fetchall()
    names, ids = {}, []
    for r in rows:
        ids.append(names.setdefault(r[0], len(names)))
    with gzip.open(out, 'wt', newline='') as f:
        w = csv.writer(f)
        w.writerow(['#names'] + [n for n, _ in sorted(names.items(), key=lambda kv: kv[1])])
        w.writerow(want)
        for r, i in zip(rows, ids):
            w.writerow((i,) + tuple(r[1:]))
    print(len(rows), 'launches,', len(names), 'kernel names ->', out, file=sys.stderr)


if __name__ == '__main__':
    main()
