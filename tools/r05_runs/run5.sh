# round 5, run 5: quad-per-voxel 6-D conv1, cluster exchange with paired polls, deterministic bucket numbering, hooks out
# of the product class -- the tests that touch them, the default line and the one-stream line, kernel stats
set -x
cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/r5_5; mkdir -p $O
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_knn_reg.py tests/test_gpu_shared_weights.py tests/test_gpu_pipeline.py tests/test_gpu_register_e2e.py tests/test_gpu_maps.py tests/test_gpu_resunet.py tests/test_gpu_me_conventions.py tests/test_gpu_model_golden.py -m gpu -q 2>&1 | tail -25 > $O/pytest_new.log)
timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity > $O/bench_c1_s1_b4.json 2> $O/bench_c1_s1_b4.err
timeout 300 python bench.py --streams 1 --pairs-per-step 1 --no-parity > $O/bench_c1_s1_b1.json 2> $O/bench_c1_s1_b1.err
timeout 300 python bench.py --no-parity --steps 30 > $O/bench_default_noparity.json 2> $O/bench_default_noparity.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/kt1 -o kt -- python $R/bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 5 > $O/kt1.log 2>&1
python $R/tools/rocpd_summary.py $O/kt1/kt_results.db $O/kernel_stats_s1_b4.csv
rm -rf $O/kt1
cat $O/pytest_new.log; head -30 $O/kernel_stats_s1_b4.csv | cut -c1-150
