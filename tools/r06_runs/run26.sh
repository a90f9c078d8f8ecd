#!/bin/bash
# round 6, run 26: (a) the bench line against the profiler LAUNCH BY LAUNCH (the plain averages of tools/evidence.sh's first
# version compared 30 stamped launches of one region with all 114 of the process: 23 % apart on this box); (b) the
# tiny-cloud edge cases of the parity-class kernel; (c) run-to-run / batch-independence checks with the new kernels
R=$PWD; O=$R/gpurun_out/run26; mkdir -p $O; rm -rf $O/*
COMMIT=$(tr -d '\n' < $R/tools/COMMIT 2>/dev/null || echo unknown)
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
timeout 300 rocprofv3 --kernel-trace -d $O/kt3 -o kt -- $B --no-parity --steps 5 > $O/kt3.log 2>&1
python $R/tools/rocpd_summary.py $O/kt3/kt_results.db $O/kernel_stats_s3_b6.csv --trace sparse_conv_wide_f16x2 $O/wide_trace_s3_b6.csv
sed -n '/^python - <<P | tee \$O\/line_vs_rocprof.txt/,/^P$/p' $R/tools/evidence.sh > $O/check.sh
O=$O COMMIT=$COMMIT R=$R bash $O/check.sh
rm -rf $O/kt3 $O/check.sh
cd $R
timeout 600 python -m pytest tests/test_gpu_resunet.py -m gpu -x -q -k "tiny" 2>&1 | tail -3
bash tools/batch_invariance.sh 2>&1 | tail -4 | tee $O/batch_invariance.txt
timeout 600 python tools/repro_stress.py 300 2>&1 | tail -2 | tee $O/repro_alone.txt
(timeout 600 python tools/repro_stress.py 300 > $O/repro_competitor.txt 2>&1 &) ; timeout 600 python tools/repro_stress.py 300 2>&1 | tail -2 | tee $O/repro_next_to_a_second_process.txt; sleep 20; tail -1 $O/repro_competitor.txt
