#!/bin/bash
# round 6, run 33: dense-tile kernel at three waves per SIMD (register budget 168: half-depth weight ring, 11-13 spilled VGPRs)
R=$PWD; O=$R/gpurun_out/run33; mkdir -p $O; rm -rf $O/*
cd $R
run() { AB_TAG=$1 timeout 300 python tools/ab_fcgf.py > $O/ab_$1.txt 2>&1; echo "== $1"; grep -E "fwd ms|dense|maps_3d" $O/ab_$1.txt; }
run occ2
DGR_HIP_LIB=$R/deepglobalregistration_amd/lib_occ3/libdgr_hip.so run occ3
