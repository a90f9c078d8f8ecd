#!/bin/bash
# round 6, run 37: every stream on its OWN share of the compute units (hipExtStreamCreateWithCUMask) instead of three
# streams competing for all of them (run36: 3 kernels in flight 73 % of the time, each stretched 2.3x on average; small
# kernels wait for a persistent wide kernel to leave: 10 us -> 850 us).  The wide kernel's grid follows the share.
R=$PWD; O=$R/gpurun_out/run37; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-parity --no-cpu-baseline --no-exact-leg"
show() { python - <<P
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']
print('$1'.split('/')[-1], 'pairs/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), 'dominant us %.0f (one stream %.0f)' % (r['avg_launch_us'], r['avg_launch_us_one_stream']))
P
}
run() { # name, streams, wide cus, extra env
  env DGR_BENCH_CU_SPLIT=1 DGR_WIDE_CUS=$3 timeout 300 $B --steps 30 --streams $2 > $O/$1.json 2> $O/$1.err; show $O/$1.json; grep 'compute units' $O/$1.err | head -4
}
run s2_c128 2 128
run s3_c80 3 80
run s4_c64 4 64
run s3_c256 3 0
run s6_c40 6 40
