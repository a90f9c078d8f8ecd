#!/bin/bash
# round 6, run 15: rows per workgroup of the list-based output-stationary kernel at the coarse levels (weights are
# re-streamed per row block: 16-row blocks at level 3 move 1.5 GB of weight fragments per 256 -> 256 launch)
R=$PWD; O=$R/gpurun_out/run15; mkdir -p $O; rm -rf $O/*
cd $R
run() { AB_TAG=$1 timeout 300 python tools/ab_fcgf.py > $O/ab_$1.txt 2>&1; echo "== $1"; grep -E "fwd ms|conv_os|maps_3d" $O/ab_$1.txt; }
run default
DGR_OS_MB3=32 run mb3_32
DGR_OS_MB3=64 run mb3_64
DGR_OS_MB2=64 run mb2_64
DGR_OS_MB2=16 run mb2_16
DGR_OS_MB2=64 DGR_OS_MB3=64 run mb23_64
