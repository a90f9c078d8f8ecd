#!/bin/bash
# round 6, run 13: dense-tile kernel (conv_dense.hip) with (a) the weight-ring requests pinned between the MFMA steps
# (scheduling barriers) and (b) the middle tensor of a residual block handed on as ready-made operand pieces ("dense
# split rows").  A/B per layer on one FCGF forward of 8 clouds: HEAD's library, the new one without (b), the new one;
# output features compared bitwise.
R=$PWD; O=$R/gpurun_out/run13; mkdir -p $O; rm -rf $O/*
cd $R
for v in base nods new; do
  case $v in
    base) export DGR_HIP_LIB=$R/deepglobalregistration_amd/lib_base/libdgr_hip.so; unset DGR_NO_DSPLIT;;
    nods) unset DGR_HIP_LIB; export DGR_NO_DSPLIT=1;;
    new)  unset DGR_HIP_LIB; unset DGR_NO_DSPLIT;;
  esac
  AB_TAG=$v AB_SAVE=1 timeout 300 python tools/ab_fcgf.py > $O/ab_$v.txt 2>&1
  grep -E "fwd ms|dense|maps_3d" $O/ab_$v.txt
done
unset DGR_HIP_LIB DGR_NO_DSPLIT
python - <<'P'
import numpy as np
a = np.load('gpurun_out/ab_F_base.npy')
for v in ('nods', 'new'):
    b = np.load('gpurun_out/ab_F_%s.npy' % v)
    print(v, 'F bitwise equal to base:', bool((a == b).all()), 'max |d|', float(np.abs(a - b).max()))
P
timeout 900 python -m pytest tests/test_gpu_dense_conv.py tests/test_gpu_resunet.py tests/test_gpu_model_golden.py -m gpu -x -q 2>&1 | tail -3
