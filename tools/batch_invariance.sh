#!/bin/bash
# Is a pair's result independent of the batch it is registered in, and of the run?  (tests/test_gpu_bench_ranks.py)
R=$PWD; O=$R/gpurun_out/binv; mkdir -p $O
C="--steps 2 --warmup 1 --total-pairs 6 --streams 1 --n-raw 12000 --conv1-ks 5 --no-parity"
python bench.py $C --pairs-per-step 2 --dump-results $O/a.npz > /dev/null 2>$O/a.err
python bench.py $C --pairs-per-step 2 --dump-results $O/b.npz > /dev/null 2>$O/b.err
python bench.py $C --pairs-per-step 3 --dump-results $O/c.npz > /dev/null 2>$O/c.err
python bench.py $C --pairs-per-step 1 --dump-results $O/d.npz > /dev/null 2>$O/d.err
python - <<PY
import numpy as np
r = {k: np.load('$O/%s.npz' % k) for k in 'abcd'}
def srt(x): o = np.argsort(x['ids']); return x['T'][o], x['stats'][o], x['status'][o]
Ta, sa, _ = srt(r['a'])
for k in 'bcd':
    T, s, _ = srt(r[k])
    print('a vs', k, 'max |dT|', np.abs(T - Ta).max(axis=(1, 2)), 'iterations', s[:, 0], 'vs', sa[:, 0])
PY
