#!/bin/bash
# round 6, run 21: stage clocks of the list-based output-stationary kernel (instrumented build), natural and parity-sorted rows
R=$PWD; O=$R/gpurun_out/run21; mkdir -p $O; rm -rf $O/*
cd $R
export DGR_HIP_LIB=$R/deepglobalregistration_amd/lib_clk/libdgr_hip.so
timeout 300 python tools/os_stage_clk.py 2>&1 | grep -v amdgpu.ids | tee $O/stage_clk.txt
AB_PARITY=1 timeout 300 python tools/os_stage_clk.py 2>&1 | grep -v amdgpu.ids | tee $O/stage_clk_parity.txt
