#!/bin/bash
# Round evidence bundle, run on the GPU box from the repo root: bench lines, rocprofv3 kernel stats and the
# PMC HBM-traffic passes.  Everything lands in gpurun_out/final/ (copy what should be judged into profiles/).
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O; rm -f $O/*
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py > $O/bench_default_3x4.json 2> $O/bench_default.err
timeout 300 python $R/bench.py --streams 1 --pairs-per-step 1 --no-cpu-baseline > $O/bench_s1_b1.json 2> $O/bench_s1_b1.err
timeout 300 rocprofv3 --kernel-trace -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --steps 5 > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $O/kt/kt_results.db $O/kernel_stats.csv --trace sparse_conv $O/conv_trace.csv
# the roofline leg of bench.py is a single-stream re-run (4 pairs per batch): the same command with one stream gives
# kernel durations free of time-slicing, comparable with roofline.dominant_kernel.avg_launch_us
timeout 300 python $R/bench.py --streams 1 --pairs-per-step 4 --no-cpu-baseline > $O/bench_s1_b4.json 2> $O/bench_s1_b4.err
timeout 300 rocprofv3 --kernel-trace -d $O/kt1 -o kt -- python $R/bench.py --streams 1 --pairs-per-step 4 --no-cpu-baseline --steps 5 > $O/kt1.log 2>&1
python $R/tools/rocpd_summary.py $O/kt1/kt_results.db $O/kernel_stats_s1_b4.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "sparse_conv|reduce_rows|conv_small|conv1_" -d $O/p_$c -o p -- \
    python $R/bench.py --streams 1 --pairs-per-step 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/p_$c.log 2>&1
done
python $R/tools/pmc_traffic.py $O/p_FETCH_SIZE/p_results.db $O/p_WRITE_SIZE/p_results.db > $O/conv_hbm_traffic.json
rm -rf $O/kt $O/kt1 $O/p_FETCH_SIZE $O/p_WRITE_SIZE
ls -la $O; tail -c 600 $O/bench_default_3x4.json
