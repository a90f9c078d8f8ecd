#!/bin/bash
# round 6, run 18: dense-tile kernel, rows per wave (16-row groups RG) and weight-ring depth: the vector-memory path
# carries 16 KB of weight fragments per wave and offset next to 4 KB x RG of gathered rows
R=$PWD; O=$R/gpurun_out/run18; mkdir -p $O; rm -rf $O/*
cd $R
run() { AB_TAG=$1 timeout 300 python tools/ab_fcgf.py > $O/ab_$1.txt 2>&1; echo "== $1"; grep -E "fwd ms|dense|maps_3d" $O/ab_$1.txt; }
run rg2
for v in rg3 rg3wd4 rg4wd4; do DGR_HIP_LIB=$R/deepglobalregistration_amd/lib_$v/libdgr_hip.so run $v; done
