# round 5, run 2: the new tests (batched kNN, shared weights, reference-run safeguard / ICP goldens, dense conv), then the
# same-box A/B of the library before (lib_base = merged dense-ring) and after (kNN batching + sampled first pass,
# multi-offset weight ring at the 32-channel dense shapes)
set -x
cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/r5_2; mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_knn_reg.py tests/test_gpu_shared_weights.py tests/test_gpu_register_e2e.py tests/test_gpu_dense_conv.py tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | tail -25 > $O/pytest_new.log)
bash tools/ab_libs.sh r5_2 deepglobalregistration_amd/lib_base/libdgr_hip.so deepglobalregistration_amd/lib/libdgr_hip.so > $O/ab.log 2>&1
timeout 300 python bench.py --no-parity --steps 30 > $O/bench_default_noparity.json 2> $O/bench_default_noparity.err
cat $O/pytest_new.log; cat $O/ab.log | tail -80
