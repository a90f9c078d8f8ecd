#!/bin/bash
# A/B of two builds of libdgr_hip.so on the GPU box: bench line (one stream, 4 pairs per batch) per library and a
# bitwise comparison of the network outputs.   bash tools/ab_libs.sh <tag> <libA> <libB> [...]
R=$PWD; tag=$1; shift; O=$R/gpurun_out/ab_$tag; mkdir -p $O
for lib in "$@"; do
  n=$(basename $(dirname $lib))
  DGR_HIP_LIB=$R/$lib timeout 300 python $R/bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 10 --warmup 2 > $O/bench_$n.json 2> $O/bench_$n.err
  DGR_HIP_LIB=$R/$lib timeout 300 python $R/tests/aux/net_modes_dump.py $O/dump_$n.npz > $O/dump_$n.log 2>&1
  python - <<PY
import json
j = json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1])
r = j['roofline']
print('$n', 'pairs/s %.1f' % j['value'], 'ms/step %.2f' % j['ms_per_step'], r['kernel'], 'avg_launch_us %.0f' % r['avg_launch_us'], j['stage_ms_per_batch'])
for k, g in r['by_kernel'].items():
    print('   ', k, g['launches'], 'ms %.3f main %.3f' % (g['ms'], g['main_kernel_ms']))
PY
done
python - <<PY
import glob, numpy as np
fs = sorted(glob.glob('$O/dump_*.npz'))
a = np.load(fs[0])
for f in fs[1:]:
    b = np.load(f)
    print(f, 'logit bitwise equal:', bool((a['logit'] == b['logit']).all()), 'F bitwise equal:', bool((a['F'] == b['F']).all()),
          'max dlogit', float(np.abs(a['logit'] - b['logit']).max()),
          'relative to max |logit|', float(np.abs(a['logit'] - b['logit']).max() / max(1e-30, np.abs(a['logit']).max())))
    for k in a.files:
        if k.startswith('i6_') and k in b.files:
            print('   ', k, 'max |d| / max |a| =', float(np.abs(a[k] - b[k]).max() / max(1e-30, np.abs(a[k]).max())), 'bitwise', bool((a[k] == b[k]).all()))
PY
