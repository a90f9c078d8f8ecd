#!/bin/bash
# run 18: the dense-tile kernel at three workgroups per CU (168 VGPRs, 13 spilled) against the shipped build (two), same box
mkdir -p gpurun_out/run18
for v in base occ3 base occ3; do
  if [ $v = occ3 ]; then export DGR_HIP_LIB=$PWD/lib_occ3/libdgr_hip.so; else unset DGR_HIP_LIB; fi
  timeout 120 python tools/r04_runs/row_order.py gpurun_out/run18/$v.json first_occurrence 2>&1 | grep first_occurrence | cut -c1-330
done
