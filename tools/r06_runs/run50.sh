#!/bin/bash
# round 6, run 50: the small-batch configurations on HALVES of the compute units (four contexts, two per half)
R=$PWD; O=$R/gpurun_out/run50; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-parity --no-cpu-baseline --no-exact-leg --streams 4 --cu-shares 2"
show() { python - <<P
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1])
print('$1'.split('/')[-1], 'pairs/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']))
P
}
timeout 200 $B --kind outdoor --n-raw 120000 --voxel 0.3 --conv1-ks 5 --pairs-per-step 4 --steps 60 > $O/c3_halves.json 2> $O/c3_halves.err; show $O/c3_halves.json
timeout 200 $B --n-raw 200000 --voxel 0.025 --pairs-per-step 1 --steps 40 > $O/c5_halves.json 2> $O/c5_halves.err; show $O/c5_halves.json
