#!/bin/bash
# round 6, run 40: results + error flag through pinned host memory, copies enqueued in front of the batch's one wait
# (two blocking pageable copies + two synchronisations behind the wait until now): latency of one pair, one stream, default
R=$PWD; O=$R/gpurun_out/run40; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-exact-leg"
show() { python - <<P
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']
print('$1'.split('/')[-1], 'pairs/s %.1f ms/step %.3f' % (d['value'], d['ms_per_step']), 'host cpu/step %.4f' % d['host_cpu_s_per_step_per_rank'], d['config'].get('parity_ok'))
P
}
cd $R && timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_register_e2e.py tests/test_gpu_maps.py -x -q -m gpu 2>&1 | tail -3; cd /tmp
timeout 300 $B --no-parity --streams 1 --pairs-per-step 1 --steps 200 > $O/b_s1b1.json 2> $O/b_s1b1.err; show $O/b_s1b1.json
timeout 300 $B --no-parity --streams 1 --steps 40 > $O/b_s1.json 2> $O/b_s1.err; show $O/b_s1.json
timeout 600 $B --steps 60 > $O/b_default.json 2> $O/b_default.err; show $O/b_default.json
DGR_SPIN_SYNC=1 timeout 300 $B --no-parity --streams 1 --pairs-per-step 1 --steps 200 > $O/b_s1b1_spin.json 2> $O/b_s1b1_spin.err; show $O/b_s1b1_spin.json
