#!/bin/bash
# round 6, run 48: the default path after the partition probe (short), and the explicit plain-stream fallback
R=$PWD; O=$R/gpurun_out/run48; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --steps 20 --no-parity --no-cpu-baseline --no-exact-leg > $O/b.json 2> $O/b.err || tail -20 $O/b.err
python - <<P
import json
d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1]); print(d['value'], d['config']['workload'][:60], d['config']['cu_partition'])
P
grep -i "partition\|stream" $O/b.err | head -5
