# round 5, run 3: new tests again (batched kNN with interleaved reference tiles, shared weights, reference-run safeguard /
# ICP goldens, dense conv, pipeline), same-box A/B of the first-pass sampling (every stage / every 2nd / every 4th), kernel stats
set -x
cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/r5_3; mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_knn_reg.py tests/test_gpu_shared_weights.py tests/test_gpu_register_e2e.py tests/test_gpu_dense_conv.py tests/test_gpu_pipeline.py -m gpu -q 2>&1 | tail -25 > $O/pytest_new.log)
bash tools/ab_libs.sh r5_3 deepglobalregistration_amd/lib_sub1/libdgr_hip.so deepglobalregistration_amd/lib_sub2/libdgr_hip.so deepglobalregistration_amd/lib/libdgr_hip.so > $O/ab.log 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/kt1 -o kt -- python $R/bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 5 > $O/kt1.log 2>&1
python $R/tools/rocpd_summary.py $O/kt1/kt_results.db $O/kernel_stats_s1_b4.csv
rm -rf $O/kt1
cat $O/pytest_new.log; grep -v "^    " $O/ab.log | tail; grep -i knn $O/kernel_stats_s1_b4.csv
