#!/bin/bash
# round 6, run 19: the shipped combination -- dense split rows for the middle tensors, per-layer phase width / row blocks
# of the list-based kernel -- against HEAD's library: per-layer times, output features bitwise; the whole GPU suite; the
# driver's command
R=$PWD; O=$R/gpurun_out/run19; mkdir -p $O; rm -rf $O/*
cd $R
for v in base new; do
  if [ $v = base ]; then export DGR_HIP_LIB=$R/deepglobalregistration_amd/lib_base/libdgr_hip.so; else unset DGR_HIP_LIB; fi
  AB_TAG=$v AB_SAVE=1 timeout 300 python tools/ab_fcgf.py > $O/ab_$v.txt 2>&1
  grep -vE "amdgpu.ids" $O/ab_$v.txt
done
unset DGR_HIP_LIB
python - <<'P'
import numpy as np
a = np.load('gpurun_out/ab_F_base.npy'); b = np.load('gpurun_out/ab_F_new.npy')
print('F bitwise equal to HEAD:', bool((a == b).all()), 'max |d|', float(np.abs(a - b).max()))
P
rm -f gpurun_out/ab_F_*.npy
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<P
import json
d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1]); r=d['roofline']
print('pairs/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), d['stage_ms_per_batch'])
print({k: r[k] for k in ('frac','frac_one_stream','c_le_64_hbm_frac','c_le_64_frac_own_pipe','c_le_64_ms_per_batch','exact_f32_pairs_per_s')})
print(d['config'].get('parity_ok'), d['config'].get('parity_within_1e-4'))
P
