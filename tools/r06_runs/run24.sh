#!/bin/bash
# round 6, run 24: parity-class transposed conv, 64 against 128 rows per workgroup
R=$PWD; O=$R/gpurun_out/run24; mkdir -p $O; rm -rf $O/*
cd $R
run() { AB_TAG=$1 timeout 300 python tools/ab_fcgf.py > $O/ab_$1.txt 2>&1; echo "== $1"; grep -E "fwd ms|maps_3d|L12|L15|L18|rror" $O/ab_$1.txt; }
DGR_UP_RG=2 run rg2
DGR_UP_RG=1 run rg1
run auto
