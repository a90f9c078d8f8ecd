#!/bin/bash
# round 6, run 7: the whole GPU suite at HEAD (cluster registration, device-side weights, f64 arbiter, in-kernel spans) and
# the driver's command
R=$PWD; O=$R/gpurun_out/run7; mkdir -p $O; rm -rf $O/*
DGR_PARITY_REPORT=$O/parity timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<P
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'host cpu', d['host_cpu_s_per_step_per_rank'], 'driver threads', d['host_cpu_s_per_step_driver_threads'])
print(json.dumps(d['roofline'])[:1500]); print(d['config']); print(d['startup_s']); print(d['stage_ms_per_batch'])
P
tail -3 $O/bench_default.err | cut -c1-300
