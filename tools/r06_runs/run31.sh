#!/bin/bash
# round 6, run 31: which kernel of the 6-D map build is slower in today's library than in 8a9cde0's (same box)?
R=$PWD; O=$R/gpurun_out/run31; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
for v in 8a9 head; do
  if [ $v = 8a9 ]; then export DGR_HIP_LIB=$R/deepglobalregistration_amd/lib_8a9/libdgr_hip.so; else unset DGR_HIP_LIB; fi
  timeout 300 rocprofv3 --kernel-trace -d $O/kt_$v -o kt -- python $R/bench.py --no-parity --streams 1 --steps 5 > $O/kt_$v.log 2>&1
  python $R/tools/rocpd_summary.py $O/kt_$v/kt_results.db $O/kernel_stats_$v.csv
  rm -rf $O/kt_$v
done
python - <<P
import csv
def load(f):
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(f)) if r.get('TotalDurationNs')}
a, b = load('$O/kernel_stats_8a9.csv'), load('$O/kernel_stats_head.csv')
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    rows.append((tb - ta, k, ca, ta, cb, tb))
for d, k, ca, ta, cb, tb in sorted(rows, key=lambda r: -abs(r[0]))[:14]:
    print('%+9.3f ms  %-70s 8a9: %5d calls %9.3f ms | head: %5d calls %9.3f ms' % (d / 1e6, k[:70], ca, ta / 1e6, cb, tb / 1e6))
P
