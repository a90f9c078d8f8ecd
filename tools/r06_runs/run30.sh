#!/bin/bash
# round 6, run 30: is maps_6d = 3.2 ms (runs 28, 29) the box or the code?  The evidence commit's library (8a9cde0, 1.99 ms
# on its box) and today's on ONE box, one stream
R=$PWD; O=$R/gpurun_out/run30; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
for v in 8a9 head 8a9 head; do
  if [ $v = 8a9 ]; then export DGR_HIP_LIB=$R/deepglobalregistration_amd/lib_8a9/libdgr_hip.so; else unset DGR_HIP_LIB; fi
  timeout 300 python $R/bench.py --no-parity --streams 1 --steps 30 > $O/b_s1_$v.json 2> $O/b_s1_$v.err
  python - <<P
import json
d=json.loads([l for l in open('$O/b_s1_$v.json') if l.startswith('{')][-1])
print('$v', 'pairs/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), d['stage_ms_per_batch'])
P
done
rocm-smi --showclocks 2>/dev/null | head -20
