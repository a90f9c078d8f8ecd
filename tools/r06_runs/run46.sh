#!/bin/bash
# round 6, run 46: the default line of the final bench.py (the driver's command), stamped
R=$PWD; O=$R/gpurun_out/run46; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
C=$(tr -d '\n' < $R/tools/COMMIT)
timeout 900 python $R/bench.py > $O/bench_c1_default.json 2> $O/bench_c1_default.err
python - <<P
import json
d=json.loads([l for l in open('$O/bench_c1_default.json') if l.startswith('{')][-1]); d['commit']='$C'; json.dump(d, open('$O/bench_c1_default.json','w'))
r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['frac_vs_whole_gpu_peak'], r['frac_one_stream'], d['host_cpu_s_per_step_per_rank'], d['config']['parity_ok'], d['config']['parity_within_1e-4'])
print(json.dumps(d['parity']['per_stream'])[:1500])
P
