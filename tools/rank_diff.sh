#!/bin/bash
R=$PWD; O=$R/gpurun_out/rankdiff; mkdir -p $O
C="--steps 2 --warmup 1 --total-pairs 6 --pairs-per-step 2 --streams 1 --n-raw 12000 --conv1-ks 5 --no-parity"
export DGR_BENCH_BACKEND=gloo DGR_BENCH_ONE_GPU=1
python bench.py --gpus 1 $C --dump-results $O/r1.npz > /dev/null 2>$O/r1.err
python bench.py --gpus 2 $C --dump-results $O/r2.npz > /dev/null 2>$O/r2.err
python bench.py --gpus 2 $C --dump-results $O/r2b.npz > /dev/null 2>$O/r2b.err
python - <<PY
import numpy as np
a = np.load('$O/r1.npz'); o = np.argsort(a['ids'])
for k in ('r2', 'r2b'):
    b = np.load('$O/%s.npz' % k); p = np.argsort(b['ids'])
    print(k, 'ids order', b['ids'].tolist(), 'max |dT| per pair', np.abs(a['T'][o] - b['T'][p]).max(axis=(1, 2)))
    print('   stats 1-rank', a['stats'][o].tolist()); print('   stats', k, b['stats'][p].tolist())
PY
grep -i "dealt\|strong" $O/r2.err | head -4
