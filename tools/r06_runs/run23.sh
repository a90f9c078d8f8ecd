#!/bin/bash
# round 6, run 23: transposed convs by parity class of the output rows (conv_up.hip) against the list-based kernel
# (DGR_NO_UP=1): per-layer times of an 8-cloud FCGF forward, output features, the network parity tests
R=$PWD; O=$R/gpurun_out/run23; mkdir -p $O; rm -rf $O/*
cd $R
run() { AB_TAG=$1 AB_SAVE=1 timeout 300 python tools/ab_fcgf.py > $O/ab_$1.txt 2>&1; echo "== $1"; grep -E "fwd ms|maps_3d|L12|L15|L18|Error|error" $O/ab_$1.txt; }
DGR_NO_UP=1 run lists
run up
python - <<'P'
import numpy as np
a = np.load('gpurun_out/ab_F_lists.npy'); b = np.load('gpurun_out/ab_F_up.npy')
print('F: max |up - lists| =', float(np.abs(a - b).max()), '(unit-norm rows), bitwise equal:', bool((a == b).all()), 'finite:', bool(np.isfinite(b).all()))
P
rm -f gpurun_out/ab_F_*.npy
timeout 1200 python -m pytest tests/test_gpu_dense_conv.py tests/test_gpu_resunet.py tests/test_gpu_model_golden.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5
