set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_large_tensor.py tests/test_gpu_resunet.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py -m gpu -x -q -s 2>&1 | tail -40) > $O/pytest.log 2>&1
timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 20 > $O/bench_a.json 2> $O/bench_a.err
DGR_KMAP_ROWMAJOR=2 timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 20 > $O/bench_e_roworder.json 2> $O/bench_e.err
timeout 300 python bench.py --no-parity --steps 30 > $O/bench_s3.json 2> $O/bench_s3.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/kt1 -o kt -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 5 > $GRAFT_REPO_ROOT/$O/kt1.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocpd_summary.py $O/kt1/kt_results.db $O/kernel_stats_s1_b4.csv
rm -rf $O/kt1
cd /tmp && DGR_KMAP_ROWMAJOR=2 timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/kt2 -o kt -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 5 > $GRAFT_REPO_ROOT/$O/kt2.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocpd_summary.py $O/kt2/kt_results.db $O/kernel_stats_roworder.csv
rm -rf $O/kt2
tail -12 $O/pytest.log
