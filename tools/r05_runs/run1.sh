# first GPU run of round 5: the merged dense-ring library -- whole GPU suite, default line, one-stream line, kernel stats
set -x
cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/r5_1; mkdir -p $O
export TMPDIR=/tmp
(DGR_PARITY_REPORT=$O/parity timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.log)
timeout 600 python bench.py > $O/bench_c1_default.json 2> $O/bench_c1_default.err
timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity > $O/bench_c1_s1_b4.json 2> $O/bench_c1_s1_b4.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/kt1 -o kt -- python $R/bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 5 > $O/kt1.log 2>&1
python $R/tools/rocpd_summary.py $O/kt1/kt_results.db $O/kernel_stats_s1_b4.csv --trace sparse_conv $O/conv_trace_s1_b4.csv
rm -rf $O/kt1
cat $O/pytest_gpu.log; tail -c 300 $O/bench_c1_default.err
