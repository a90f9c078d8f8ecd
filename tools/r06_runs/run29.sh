#!/bin/bash
# round 6, run 29: the three strided 6-D maps searched from the fine side (parity of the fine row's coordinates names its
# candidate parents: 11 full-key look-ups per fine row on average) against the coarse-side bucket search (DGR_KMAP_COARSE_SIDE=1)
R=$PWD; O=$R/gpurun_out/run29; mkdir -p $O; rm -rf $O/*
cd $R && timeout 900 python -m pytest tests/test_gpu_maps.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for v in coarse fine; do
  if [ $v = coarse ]; then export DGR_KMAP_COARSE_SIDE=1; else unset DGR_KMAP_COARSE_SIDE; fi
  timeout 300 python $R/bench.py --no-parity --streams 1 --steps 40 > $O/b_s1_$v.json 2> $O/b_s1_$v.err
  python - <<P
import json
d=json.loads([l for l in open('$O/b_s1_$v.json') if l.startswith('{')][-1])
print('$v', 'pairs/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), 'maps_6d', d['stage_ms_per_batch']['maps_6d'], 'inlier_net', d['stage_ms_per_batch']['inlier_net'])
P
done
for v in coarse fine coarse fine; do
  if [ $v = coarse ]; then export DGR_KMAP_COARSE_SIDE=1; else unset DGR_KMAP_COARSE_SIDE; fi
  timeout 300 python $R/bench.py --no-parity --steps 60 > $O/b_default_$v.json 2> $O/b_default_$v.err
  python - <<P
import json
d=json.loads([l for l in open('$O/b_default_$v.json') if l.startswith('{')][-1])
print('$v default', 'pairs/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), 'maps_6d', d['stage_ms_per_batch']['maps_6d'], 'dominant 3-stream us %.0f' % d['roofline']['avg_launch_us'])
P
done
