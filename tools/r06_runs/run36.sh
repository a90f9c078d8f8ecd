#!/bin/bash
# round 6, run 36: full kernel timeline of the timed 3-stream configuration (and of one stream) -- where do three streams
# lose the time one stream's kernels add up to?
R=$PWD; O=$R/gpurun_out/run36; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-parity --no-cpu-baseline --no-exact-leg"
timeout 400 rocprofv3 --kernel-trace -d $O/kt3 -o kt -- $B --steps 8 --warmup 3 > $O/kt3.log 2>&1
python $R/tools/timeline_dump.py $O/kt3/kt_results.db $O/timeline_s3_b6.csv.gz
timeout 400 rocprofv3 --kernel-trace -d $O/kt1 -o kt -- $B --steps 8 --warmup 3 --streams 1 > $O/kt1.log 2>&1
python $R/tools/timeline_dump.py $O/kt1/kt_results.db $O/timeline_s1_b6.csv.gz
grep '^{' $O/kt3.log | tail -1 | cut -c1-300; grep '^{' $O/kt1.log | tail -1 | cut -c1-300
rm -rf $O/kt3 $O/kt1
ls -la $O
