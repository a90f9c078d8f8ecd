#!/usr/bin/env python
"""HBM traffic of the sparse-conv kernels from two rocprofv3 PMC passes (rocpd SQLite results):
    python tools/pmc_traffic.py fetch_results.db write_results.db > profiles/rNN_conv_hbm_traffic.json
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reads 1/2 of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section) -> fetch bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is used as is.
A "conv launch" is one layer launch (MFMA kernel, small-Cin kernel or fused conv1); its reduce pass is
counted into the same launch."""
import json
import sqlite3
import sys

LAYER_KERNELS = ('sparse_conv_mfma', 'conv_small_cin_kernel', 'conv_small_cin_row_kernel', 'conv_cin6_quad_kernel', 'identity_conv_kernel', 'conv1_grid_kernel', 'conv1_grid_mfma', 'conv1_probe_kernel')


def sums(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute('select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection '
                       'where counter_name = ? group by kernel_name', (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


def main():
    fetch, write = sums(sys.argv[1], 'FETCH_SIZE'), sums(sys.argv[2], 'WRITE_SIZE')
    kernels, launches, fkb, wkb = {}, 0, 0.0, 0.0
    for name in sorted(set(fetch) | set(write)):
        f, n = fetch.get(name, (0.0, 0))
        w, n2 = write.get(name, (0.0, 0))
        kernels[name] = {'dispatches': max(n, n2), 'FETCH_SIZE_KB': f, 'WRITE_SIZE_KB': w}
        fkb += f
        wkb += w
        if any(k in name for k in LAYER_KERNELS) and 'conv1_probe' not in name:
            launches += max(n, n2)
    out = {'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --kernel-include-regex '
                      '"sparse_conv|reduce_rows|conv_small|conv1_" -- python bench.py --streams 1 --pairs-per-step 4 '
                      '--steps 2 --warmup 1 --no-cpu-baseline (two separate passes)',
           'note': 'fetch_bytes_corrected = 2 x FETCH_SIZE x 1024 (gfx950 correction); WRITE_SIZE x 1024 as is',
           'kernels': kernels, 'conv_launches': launches,
           'per_conv_launch_bytes': {'fetch_corrected': 2 * fkb * 1024 / max(launches, 1),
                                     'write': wkb * 1024 / max(launches, 1),
                                     'total': (2 * fkb + wkb) * 1024 / max(launches, 1)}}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
