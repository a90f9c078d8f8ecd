#!/bin/bash
# round 6, run 9: placing pass with two records in flight per lane (maps_6d), adaptive nap of the batch wait (single-pair
# latency), a streams x batch sweep now that the host threads sleep
R=$PWD; O=$R/gpurun_out/run9; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
show() { python - <<P
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']
print('$1'.split('/')[-1], 'pairs/s %.1f ms/step %.2f host_cpu_s/step %.3f | maps_6d %.3f reg %.2f' % (d['value'], d['ms_per_step'], d['host_cpu_s_per_step_per_rank'], d['stage_ms_per_batch']['maps_6d'], d['stage_ms_per_batch']['registration']))
P
}
timeout 300 $B --no-parity --steps 30 > $O/b_default.json 2> $O/b_default.err; show $O/b_default.json
timeout 300 $B --no-parity --streams 1 --pairs-per-step 1 --steps 50 > $O/b_s1b1.json 2> $O/b_s1b1.err; show $O/b_s1b1.json
DGR_SPIN_SYNC=1 timeout 300 $B --no-parity --streams 1 --pairs-per-step 1 --steps 50 > $O/b_s1b1_spin.json 2> $O/b_s1b1_spin.err; show $O/b_s1b1_spin.json
for sb in "4 6" "3 8" "2 8" "4 4"; do set -- $sb
  timeout 300 $B --no-parity --streams $1 --pairs-per-step $2 --steps 20 > $O/b_s$1_b$2.json 2> $O/b_s$1_b$2.err; show $O/b_s$1_b$2.json
done
cd $R && timeout 900 python -m pytest tests/test_gpu_maps.py tests/test_gpu_resunet.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -4
