#!/bin/bash
# round 6, run 16: list-based output-stationary kernel with 128 input channels per pipeline phase (half the phases --
# barrier + dependent LDS / L2 round trips each -- at Cin >= 128) against 64
R=$PWD; O=$R/gpurun_out/run16; mkdir -p $O; rm -rf $O/*
cd $R
run() { AB_TAG=$1 timeout 300 python tools/ab_fcgf.py > $O/ab_$1.txt 2>&1; echo "== $1"; grep -E "fwd ms|conv_os|maps_3d" $O/ab_$1.txt; }
run default
DGR_HIP_LIB=$R/deepglobalregistration_amd/lib_ck128/libdgr_hip.so run ck128
