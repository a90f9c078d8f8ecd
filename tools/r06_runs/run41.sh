#!/bin/bash
# round 6, run 41: kernel timeline of ONE pair per call on one stream (latency): where do 5.9 ms go?
R=$PWD; O=$R/gpurun_out/run41; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-parity --no-cpu-baseline --no-exact-leg"
timeout 400 rocprofv3 --kernel-trace -d $O/kt -o kt -- $B --steps 40 --warmup 5 --streams 1 --pairs-per-step 1 > $O/kt.log 2>&1
python $R/tools/timeline_dump.py $O/kt/kt_results.db $O/timeline_s1_b1.csv.gz
grep '^{' $O/kt.log | tail -1 | cut -c1-260
rm -rf $O/kt
