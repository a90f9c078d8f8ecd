#!/bin/bash
# round 6, run 35: does the persistent wide kernel pay for leaving a few CUs per XCD to the other streams' latency-bound
# launches?  DGR_WIDE_CUS caps its grid (results do not depend on the grid: product rows are per pair).
R=$PWD; O=$R/gpurun_out/run35; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-parity --no-cpu-baseline --no-exact-leg"
show() { python - <<P
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']
print('$1'.split('/')[-1], 'pairs/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), 'dominant us %.0f (one stream %.0f)' % (r['avg_launch_us'], r['avg_launch_us_one_stream']))
P
}
for c in 0 248 240 224 208 192; do
  DGR_WIDE_CUS=$c timeout 300 $B --steps 40 > $O/b_s3_c$c.json 2> $O/b_s3_c$c.err; show $O/b_s3_c$c.json
done
for c in 240 224; do
  DGR_WIDE_CUS=$c timeout 300 $B --steps 40 --streams 4 > $O/b_s4_c$c.json 2> $O/b_s4_c$c.err; show $O/b_s4_c$c.json
done
DGR_WIDE_CUS=240 timeout 300 $B --steps 40 --streams 1 > $O/b_s1_c240.json 2> $O/b_s1_c240.err; show $O/b_s1_c240.json
