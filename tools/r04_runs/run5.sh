set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_large_tensor.py tests/test_gpu_resunet.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py tests/test_gpu_maps.py -m gpu -x -q -s 2>&1 | tail -30) > $O/pytest.log 2>&1
for rep in 1 2; do
DGR_HIP_LIB=$PWD/deepglobalregistration_amd/lib_wd2/libdgr_hip.so timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 30 > $O/bench_wd2_$rep.json 2> $O/bench_wd2_$rep.err
timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 30 > $O/bench_new_$rep.json 2> $O/bench_new_$rep.err
done
tail -12 $O/pytest.log
