# round 5, run 4: the registration kernel as a cluster of workgroups per pair (tests that exercise it at every cluster
# size), shared weights test, same-box A/B against the one-workgroup build (lib_cl1: -DDGR_REG_CLUSTER_MAX=1)
set -x
cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/r5_4; mkdir -p $O
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests/test_gpu_knn_reg.py tests/test_gpu_shared_weights.py tests/test_gpu_pipeline.py tests/test_gpu_register_e2e.py tests/test_gpu_bench_ranks.py -m gpu -q 2>&1 | tail -25 > $O/pytest_new.log)
bash tools/ab_libs.sh r5_4 deepglobalregistration_amd/lib_cl1/libdgr_hip.so deepglobalregistration_amd/lib/libdgr_hip.so > $O/ab.log 2>&1
for l in lib_cl1 lib; do DGR_HIP_LIB=$R/deepglobalregistration_amd/$l/libdgr_hip.so timeout 300 python bench.py --streams 1 --pairs-per-step 1 --no-parity --steps 20 > $O/bench_s1_b1_$l.json 2> $O/bench_s1_b1_$l.err; done
timeout 300 python bench.py --no-parity --steps 30 > $O/bench_default_noparity.json 2> $O/bench_default_noparity.err
cat $O/pytest_new.log; grep -v "^    " $O/ab.log | tail
