#!/bin/bash
# round 6, run 2: is the SHIPPED one-workgroup registration kernel reproducible next to a second process at 10x the
# exposure of round 3 / 5 (20 000 runs)?  Positive control: the f32-partials build (round 3: 4 % of 3000 runs differ).
R=$PWD; O=$R/gpurun_out/contention6; mkdir -p $O; rm -f $O/*
python -c "import torch" 2>/dev/null
run() {  # variant, runs
  lib=$R/deepglobalregistration_amd/lib_v/$1/libdgr_hip.so
  DGR_HIP_LIB=$lib timeout 400 python tools/repro_stress.py 100000 12000 > $O/comp_$1.txt 2>&1 &
  CP=$!
  sleep 20
  DGR_HIP_LIB=$lib timeout 300 python tools/repro_reg.py $2 2>&1 | tail -1 > $O/reg_$1.txt
  kill $CP 2>/dev/null; wait $CP 2>/dev/null
  echo "== $1 ($2 runs next to the pipeline competitor): $(cat $O/reg_$1.txt)"
}
run f32part 3000
run f32noslp 3000
run base 20000
echo "== base alone: $(DGR_HIP_LIB=$R/deepglobalregistration_amd/lib_v/base/libdgr_hip.so timeout 200 python tools/repro_reg.py 5000 2>&1 | tail -1)"
