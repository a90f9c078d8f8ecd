#!/bin/bash
# round 6, run 6: weights prepared on the device (bit identity tests, the forced one-rank RCCL group), sleeping stream
# wait (host CPU per step), in-kernel spans after the memset initialisation
R=$PWD; O=$R/gpurun_out/run6; mkdir -p $O; rm -rf $O/*
timeout 1200 python -m pytest tests/test_gpu_device_weights.py tests/test_gpu_bench_ranks.py tests/test_gpu_shared_weights.py -m gpu -x -q 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
show() { python - <<P
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']
print('$1'.split('/')[-1], 'pairs/s %.1f ms/step %.2f host_cpu_s/step %.3f | us timed-config %.1f one-stream %.1f frac %.3f frac_one_stream %.3f startup %s' % (d['value'], d['ms_per_step'], d['host_cpu_s_per_step_per_rank'], r['avg_launch_us'], r['avg_launch_us_one_stream'], r['frac'], r['frac_one_stream'], d['startup_s']))
P
}
timeout 300 $B --no-parity --steps 30 > $O/b_default.json 2> $O/b_default.err; show $O/b_default.json
DGR_SPIN_SYNC=1 timeout 300 $B --no-parity --steps 30 > $O/b_spin.json 2> $O/b_spin.err; show $O/b_spin.json
timeout 300 $B --no-parity --streams 1 --pairs-per-step 1 --steps 50 > $O/b_s1b1.json 2> $O/b_s1b1.err; show $O/b_s1b1.json
DGR_BENCH_FORCE_PG=1 timeout 300 $B --no-parity --steps 10 > $O/b_forcepg.json 2> $O/b_forcepg.err; show $O/b_forcepg.json
for cfg in "kt3:"; do
  n=${cfg%%:*}; f=${cfg#*:}
  timeout 300 rocprofv3 --kernel-trace -d $O/$n -o kt -- $B --no-parity --steps 10 $f > $O/$n.log 2>&1
  python $R/tools/rocpd_summary.py $O/$n/kt_results.db $O/kernel_stats_$n.csv
  python - <<P
import csv, json
d = json.loads([l for l in open('$O/$n.log') if l.startswith('{')][-1]); k = d['roofline']['kernel']
us = d['roofline']['avg_launch_us']; us1 = d['roofline']['avg_launch_us_one_stream']
rows = [r for r in csv.DictReader(open('$O/kernel_stats_$n.csv')) if k in r['Name']]
avg = float(rows[0]['AverageNs']) / 1e3
print(f'$n: {k}: line timed-config {us:.1f} us / one-stream re-run {us1:.1f} us; rocprofv3 average of the whole process {avg:.1f} us over {rows[0]["Calls"]} calls')
P
  rm -rf $O/$n
done
