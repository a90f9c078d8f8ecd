#!/bin/bash
# round 6, run 12: one level below "the SLP vectoriser's code": the shipped source of reg.hip built WITH the vectoriser
# (88 packed-f32 instructions) in three ways, 6000 runs each next to the pipeline competitor:
#   slp        default flags                                   (expected: a few % of the runs differ, run3)
#   slp_wait0  + -mllvm -amdgpu-waitcnt-forcezero              (every s_waitcnt waits for everything: a missing wait cannot bite)
#   slp_nopk   + -target-feature -packed-fp32-ops              (the vectoriser's IR, but no v_pk_*_f32 instructions selected)
R=$PWD; O=$R/gpurun_out/run12; mkdir -p $O; rm -rf $O/*
python -c "import torch" 2>/dev/null
run() {
  lib=$R/deepglobalregistration_amd/lib_v/$1/libdgr_hip.so
  DGR_HIP_LIB=$lib timeout 400 python tools/repro_stress.py 100000 12000 > $O/comp_$1.txt 2>&1 &
  CP=$!
  sleep 20
  DGR_HIP_LIB=$lib timeout 300 python tools/repro_reg.py $2 2>&1 | tail -1 > $O/reg_$1.txt
  kill $CP 2>/dev/null; wait $CP 2>/dev/null
  echo "== $1 ($2 runs next to the pipeline competitor): $(cat $O/reg_$1.txt)"
}
run slp 6000
run slp_wait0 6000
run slp_nopk 6000
run slp 6000
run slp_wait0 6000
