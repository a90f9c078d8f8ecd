#!/bin/bash
# Drives tools/microbench/handoff (see the header of handoff.hip) on the GPU box, from the repo root:
#   phase A  alone;  phase B  next to `handoff hammer` (a second PROCESS);  phase C  next to the full-pipeline competitor
#   (tools/repro_stress.py, the instrument of rounds 3 and 5).  Output: gpurun_out/handoff/table.txt
R=$PWD; O=$R/gpurun_out/handoff; mkdir -p $O; rm -f $O/*
H=$R/tools/microbench/handoff
[ -x $H ] || { echo "build first: hipcc -O3 --offload-arch=gfx950 tools/microbench/handoff.hip -o tools/microbench/handoff"; exit 1; }
L=${L:-4000}
{
echo "## phase A: alone"
timeout 120 $H arith $L
for v in 0 2; do timeout 120 $H xchg $L $v; done
echo "## phase B: next to a second process (handoff hammer)"
timeout 400 $H hammer 300 > $O/hammer.txt 2>&1 &
HP=$!
sleep 3
timeout 120 $H arith $L
for v in ${VARIANTS:-0 1 2 4 8 16 10}; do timeout 120 $H xchg $L $v; done
kill $HP 2>/dev/null; wait $HP 2>/dev/null
cat $O/hammer.txt
if [ "$1" != nopipe ]; then
  echo "## phase C: next to the full-pipeline competitor (tools/repro_stress.py)"
  python -c "import torch" 2>/dev/null
  timeout 400 python $R/tools/repro_stress.py 100000 12000 > $O/comp.txt 2>&1 &
  CP=$!
  sleep 25
  timeout 120 $H arith $L
  for v in 0 2 8; do timeout 120 $H xchg $L $v; done
  kill $CP 2>/dev/null; wait $CP 2>/dev/null
  tail -2 $O/comp.txt | cut -c1-300
fi
} 2>&1 | tee $O/table.txt
