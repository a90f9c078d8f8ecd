#!/bin/bash
# round 6, run 45: more than one context per share of the compute units (contexts x shares), against the default 4 x 4
R=$PWD; O=$R/gpurun_out/run45; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-parity --no-cpu-baseline --no-exact-leg --steps 30"
show() { python - <<P
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']
print('$1'.split('/')[-1], 'pairs/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), d['config']['cu_partition'])
P
}
run() { timeout 400 $B --streams $1 --cu-shares $2 --pairs-per-step $3 > $O/s$1_n$2_b$3.json 2> $O/s$1_n$2_b$3.err || tail -3 $O/s$1_n$2_b$3.err; show $O/s$1_n$2_b$3.json; }
run 4 4 6
run 4 2 6
run 6 2 6
run 8 4 6
run 8 4 3
run 6 2 4
run 4 4 6
