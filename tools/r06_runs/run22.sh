#!/bin/bash
# round 6, run 22: list-based output-stationary kernel with gathered rows and weights requested TWO phases ahead (PF = 2)
R=$PWD; O=$R/gpurun_out/run22; mkdir -p $O; rm -rf $O/*
cd $R
run() { AB_TAG=$1 AB_SAVE=1 timeout 300 python tools/ab_fcgf.py > $O/ab_$1.txt 2>&1; echo "== $1"; grep -E "fwd ms|conv_os|maps_3d" $O/ab_$1.txt; }
DGR_OS_PF=1 run pf1
DGR_OS_PF=2 run pf2
DGR_OS_PF=2 DGR_OS_MB3=32 run pf2_mb3_32
python - <<'P'
import numpy as np
a = np.load('gpurun_out/ab_F_pf1.npy')
for v in ('pf2', 'pf2_mb3_32'):
    b = np.load('gpurun_out/ab_F_%s.npy' % v)
    print(v, 'F bitwise equal to pf1:', bool((a == b).all()), 'max |d|', float(np.abs(a - b).max()))
P
rm -f gpurun_out/ab_F_*.npy
