#!/bin/bash
# round 6, run 39: role streams (dgr_ctx_create_role_streams) -- the 6-D conv layers of every worker on a HEAVY set of
# compute units, everything else on the LIGHT set: heavy_cus x workers
R=$PWD; O=$R/gpurun_out/run39; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-exact-leg"
show() { python - <<P
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']
print('$1'.split('/')[-1], 'pairs/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), 'dominant us %.0f (one stream %.0f)' % (r['avg_launch_us'], r['avg_launch_us_one_stream']), d['config'].get('parity_ok'))
P
}
run() { # heavy, streams, batch, extra
  n=h${1}_s${2}_b${3}
  timeout 400 $B --steps 24 --streams $2 --pairs-per-step $3 --heavy-cus $1 $4 > $O/$n.json 2> $O/$n.err || tail -5 $O/$n.err; show $O/$n.json
}
run 0 3 6 --no-parity
run 128 3 6 --no-parity
run 160 3 6 --no-parity
run 192 3 6 --no-parity
run 160 4 6 --no-parity
run 192 4 6 --no-parity
run 160 2 6 --no-parity
run 160 3 6
