set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r9; mkdir -p $O
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests/test_gpu_maps.py tests/test_gpu_resunet.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py tests/test_gpu_split_f64.py -m gpu -x -q 2>&1 | tail -15) > $O/pytest.log 2>&1
for rep in 1 2; do
DGR_HIP_LIB=$PWD/deepglobalregistration_amd/lib_prev/libdgr_hip.so timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 30 > $O/bench_prev_$rep.json 2> $O/bench_prev_$rep.err
timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 30 > $O/bench_new_$rep.json 2> $O/bench_new_$rep.err
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/kt1 -o kt -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 5 > $GRAFT_REPO_ROOT/$O/kt1.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocpd_summary.py $O/kt1/kt_results.db $O/kernel_stats_s1_b4.csv
rm -rf $O/kt1
tail -8 $O/pytest.log
