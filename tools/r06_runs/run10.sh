#!/bin/bash
# round 6, run 10: predictive sleeping wait (one sleep of 80 % of the predicted remaining time, then polling): single-pair
# latency and CPU per step against napping (run 9) and spinning
R=$PWD; O=$R/gpurun_out/run10; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
show() { python - <<P
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']
print('$1'.split('/')[-1], 'pairs/s %.1f ms/step %.2f host_cpu_s/step %.4f driver %.4f | maps_6d %.3f' % (d['value'], d['ms_per_step'], d['host_cpu_s_per_step_per_rank'], d['host_cpu_s_per_step_driver_threads'], d['stage_ms_per_batch']['maps_6d']))
P
}
timeout 300 $B --no-parity --steps 40 > $O/b_default.json 2> $O/b_default.err; show $O/b_default.json
DGR_SPIN_SYNC=1 timeout 300 $B --no-parity --steps 40 > $O/b_spin.json 2> $O/b_spin.err; show $O/b_spin.json
timeout 300 $B --no-parity --streams 1 --pairs-per-step 1 --steps 80 > $O/b_s1b1.json 2> $O/b_s1b1.err; show $O/b_s1b1.json
DGR_SPIN_SYNC=1 timeout 300 $B --no-parity --streams 1 --pairs-per-step 1 --steps 80 > $O/b_s1b1_spin.json 2> $O/b_s1b1_spin.err; show $O/b_s1b1_spin.json
timeout 300 $B --no-parity --streams 1 --steps 40 > $O/b_s1.json 2> $O/b_s1.err; show $O/b_s1.json
cd $R && timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_maps.py -m gpu -x -q 2>&1 | tail -3
