#!/bin/bash
# Does a result change when another process shares the GPU?  Two single-rank benches run concurrently, twice: with the
# split-operand kernels and with DGR_EXACT_F32=1; each dump is compared with a solo run.
R=$PWD; O=$R/gpurun_out/race; mkdir -p $O
C="--steps 6 --warmup 1 --total-pairs 6 --streams 1 --n-raw 12000 --conv1-ks 5 --no-parity --pairs-per-step 2"
for mode in split f32; do
  [ $mode = f32 ] && export DGR_EXACT_F32=1 || unset DGR_EXACT_F32
  python bench.py $C --dump-results $O/${mode}_solo.npz > /dev/null 2>$O/${mode}_solo.err
  python bench.py $C --dump-results $O/${mode}_c1.npz > /dev/null 2>$O/${mode}_c1.err &
  python bench.py $C --dump-results $O/${mode}_c2.npz > /dev/null 2>$O/${mode}_c2.err &
  wait
done
python - <<PY
import numpy as np
for mode in ('split', 'f32'):
    a = np.load('$O/%s_solo.npz' % mode); o = np.argsort(a['ids'])
    for k in ('c1', 'c2'):
        b = np.load('$O/%s_%s.npz' % (mode, k)); p = np.argsort(b['ids'])
        print(mode, k, 'max |dT| per pair', np.abs(a['T'][o] - b['T'][p]).max(axis=(1, 2)), 'iterations equal', bool((a['stats'][o][:, 0] == b['stats'][p][:, 0]).all()))
PY
