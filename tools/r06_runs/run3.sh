#!/bin/bash
# round 6, run 3: the WITHDRAWN cluster registration kernel (commit 8857b39 built in build_wt_cluster/) next to the pipeline
# competitor: default build against a build of reg.hip with -fno-slp-vectorize (the flag that makes the f32-partials kernel
# of round 3 reproducible: run2)
R=$PWD/build_wt_cluster; O=$PWD/gpurun_out/contention6c; mkdir -p $O; rm -f $O/*
cd $R
python -c "import torch" 2>/dev/null
run() {  # name, lib dir, runs
  lib=$R/deepglobalregistration_amd/$2/libdgr_hip.so
  DGR_HIP_LIB=$lib timeout 400 python tools/repro_stress.py 100000 12000 > $O/comp_$1.txt 2>&1 &
  CP=$!
  sleep 20
  DGR_HIP_LIB=$lib timeout 300 python tools/repro_reg.py $3 2>&1 | tail -1 > $O/reg_$1.txt
  kill $CP 2>/dev/null; wait $CP 2>/dev/null
  echo "== $1 ($3 runs next to the pipeline competitor): $(cat $O/reg_$1.txt)"
}
run cluster lib 6000
run cluster_noslp lib_noslp 6000
run cluster lib 6000
run cluster_noslp lib_noslp 6000
