#!/bin/bash
# round 6, run 14: is the dense-tile kernel bounded by the gather of rows in first-occurrence (random) order?  The same
# A/B as run 13 with the rows of every cloud in Morton order (AB_SORT=1: what a spatially sorted row numbering would give)
R=$PWD; O=$R/gpurun_out/run14; mkdir -p $O; rm -rf $O/*
cd $R
for s in 0 1; do
for v in base new; do
  case $v in
    base) export DGR_HIP_LIB=$R/deepglobalregistration_amd/lib_base/libdgr_hip.so;;
    new)  unset DGR_HIP_LIB;;
  esac
  if [ $s = 1 ]; then export AB_SORT=1; else unset AB_SORT; fi
  AB_TAG=$v timeout 300 python tools/ab_fcgf.py > $O/ab_${v}_sort$s.txt 2>&1
  cat $O/ab_${v}_sort$s.txt | grep -v "^$"
done
done
