#!/bin/bash
# round 6, run 17: 128 channels per phase combined with larger row blocks at the coarse levels
R=$PWD; O=$R/gpurun_out/run17; mkdir -p $O; rm -rf $O/*
cd $R
export DGR_HIP_LIB=$R/deepglobalregistration_amd/lib_ck128/libdgr_hip.so
run() { AB_TAG=$1 timeout 300 python tools/ab_fcgf.py > $O/ab_$1.txt 2>&1; echo "== $1"; grep -E "fwd ms|conv_os|maps_3d" $O/ab_$1.txt | grep -vE "L 3|L 6|L18"; }
DGR_OS_MB3=32 run ck128_mb3_32
DGR_OS_MB3=64 run ck128_mb3_64
DGR_OS_MB2=64 run ck128_mb2_64
