#!/bin/bash
# Is the registration kernel reproducible when another process shares the GPU?  Variant libraries (only reg.o differs)
# run the kernel-only stress (tools/repro_reg.py: fixed inputs, every run compared bit for bit with the first) next to a
# full-pipeline competitor process (tools/repro_stress.py, which checks itself too).
#   bash tools/contention_reg.sh build      here: builds deepglobalregistration_amd/lib_v/<variant>/libdgr_hip.so
#   bash tools/contention_reg.sh            on the GPU box, from the repo root
# Measured (round 3, 3000 runs each, twice): default (f64 partial sums) 0 differing runs; -DDGR_REG_F32_PARTIALS
# 126 / 115; the same with a second barrier per iteration 105; with reciprocal multiplies instead of divisions 196;
# f32 partials built with -fno-slp-vectorize (no packed-f32 chains) 0; any variant without the competitor 0.
R=$PWD
if [ "$1" = build ]; then
  cd $R/deepglobalregistration_amd/csrc && make -s && mkdir -p build_v || exit 1
  for v in "base:" "f32part:-DDGR_REG_F32_PARTIALS" "f32noslp:-DDGR_REG_F32_PARTIALS -fno-slp-vectorize"; do
    n=${v%%:*}; f=${v#*:}; mkdir -p ../lib_v/$n
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-gpu-rdc -I../../include $f -c reg.hip -o build_v/reg_$n.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_v/$n/libdgr_hip.so $(ls build/*.o | grep -v "/reg.o") build_v/reg_$n.o || exit 1
  done
  exit 0
fi
O=$R/gpurun_out/contention; mkdir -p $O; rm -f $O/*
python -c "import torch" 2>/dev/null
for v in ${VARIANTS:-base f32part f32noslp}; do
  lib=$R/deepglobalregistration_amd/lib_v/$v/libdgr_hip.so
  [ -f $lib ] || { echo "missing $lib"; continue; }
  DGR_HIP_LIB=$lib timeout 300 python tools/repro_stress.py ${NCOMP:-1500} 12000 > $O/comp_$v.txt 2>&1 &
  CP=$!
  sleep 8
  DGR_HIP_LIB=$lib timeout 200 python tools/repro_reg.py ${NREG:-3000} 2>&1 | tail -1 > $O/reg_$v.txt
  wait $CP
  echo "== $v: $(cat $O/reg_$v.txt)"; tail -1 $O/comp_$v.txt | cut -c1-200
done
