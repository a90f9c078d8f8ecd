# round 5, run 9: kNN second pass with one wave-level candidate test per tile and LDS fragments read a tile ahead,
# same-box A/B against the build before (lib_prev), kNN tests
set -x
cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/r5_9; mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_knn_reg.py -m gpu -q -k knn 2>&1 | tail -5 > $O/pytest_knn.log)
bash tools/ab_libs.sh r5_9 deepglobalregistration_amd/lib_prev/libdgr_hip.so deepglobalregistration_amd/lib/libdgr_hip.so > $O/ab.log 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/kt1 -o kt -- python $R/bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 5 > $O/kt1.log 2>&1
python $R/tools/rocpd_summary.py $O/kt1/kt_results.db $O/kernel_stats_s1_b4.csv
rm -rf $O/kt1
cat $O/pytest_knn.log; grep -v "^    " $O/ab.log | tail -6; grep -i "knn_mfma" $O/kernel_stats_s1_b4.csv | cut -c1-40,150-260
