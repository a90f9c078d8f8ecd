#!/bin/bash
# round 6, run 49: same-box A/B, three times each: four contexts on quarters (the default) against four contexts on halves
R=$PWD; O=$R/gpurun_out/run49; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-parity --no-cpu-baseline --no-exact-leg --steps 40"
show() { python - <<P
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1])
print('$1'.split('/')[-1], 'pairs/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']))
P
}
for i in 1 2 3; do
  timeout 300 $B --streams 4 --cu-shares 4 > $O/q_$i.json 2> $O/q_$i.err; show $O/q_$i.json
  timeout 300 $B --streams 4 --cu-shares 2 > $O/h_$i.json 2> $O/h_$i.err; show $O/h_$i.json
done
