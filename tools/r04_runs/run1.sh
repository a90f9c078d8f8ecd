set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests/test_gpu_maps.py tests/test_gpu_resunet.py tests/test_gpu_pipeline.py tests/test_gpu_register_e2e.py tests/test_gpu_bench_ranks.py tests/test_gpu_o3d.py -m gpu -x -q -s 2>&1 | tail -60) > gpurun_out/r1/pytest.log 2>&1
timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 20 > gpurun_out/r1/bench_s1_b4.json 2> gpurun_out/r1/bench_s1_b4.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r1/kt1 -o kt -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 5 > $GRAFT_REPO_ROOT/gpurun_out/r1/kt1.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocpd_summary.py gpurun_out/r1/kt1/kt_results.db gpurun_out/r1/kernel_stats_s1_b4.csv
rm -rf gpurun_out/r1/kt1
tail -5 gpurun_out/r1/pytest.log; tail -c 600 gpurun_out/r1/bench_s1_b4.err
