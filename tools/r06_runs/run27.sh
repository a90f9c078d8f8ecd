#!/bin/bash
# round 6, run 27: the three mask-count launches of the 6-D map builder as one, words requested together
R=$PWD; O=$R/gpurun_out/run27; mkdir -p $O; rm -rf $O/*
cd $R && timeout 900 python -m pytest tests/test_gpu_maps.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --no-parity --streams 1 --steps 40 > $O/b_s1.json 2> $O/b_s1.err
python - <<P
import json
d=json.loads([l for l in open('$O/b_s1.json') if l.startswith('{')][-1])
print('pairs/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), d['stage_ms_per_batch'])
P
