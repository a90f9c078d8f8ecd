#!/bin/bash
# round 6, run 38: CU partitions per stream, three layouts (a share of every XCD spread over its shader engines / inside
# as few shader engines as possible / whole XCDs with their own L2), 2 .. 8 streams
R=$PWD; O=$R/gpurun_out/run38; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-parity --no-cpu-baseline --no-exact-leg"
show() { python - <<P
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']
print('$1'.split('/')[-1], 'pairs/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), 'dominant us %.0f (one stream %.0f)' % (r['avg_launch_us'], r['avg_launch_us_one_stream']))
P
}
run() { # layout, streams, wide cus, batch
  n=${1}_s${2}_c${3}_b${4}
  env DGR_BENCH_CU_SPLIT=$1 DGR_WIDE_CUS=$3 timeout 300 $B --steps 24 --streams $2 --pairs-per-step $4 > $O/$n.json 2> $O/$n.err; show $O/$n.json; grep 'compute units' $O/$n.err | head -1
}
run slots 3 80 6
run slots 4 64 6
run se 4 64 6
run xcd 4 64 6
run xcd 2 128 6
run slots 5 48 6
run xcd 8 32 6
run xcd 8 32 3
run slots 4 64 8
