# round 5, run 6: (1) Infinity-Cache probe (tools/microbench/mall_probe): write / read-back rates of reused regions of
# 32 MB .. 1 GB and the FETCH_SIZE / WRITE_SIZE counters of the same launches (calibrated on a 2-GB streaming read);
# (2) the kNN overflow path fixed (thread-per-query exact kernel, split scan): kNN tests, one-stream line, kernel stats
set -x
cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/r5_6; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 120 $R/tools/microbench/mall_probe > $O/mall_probe.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o f -- $R/tools/microbench/mall_probe > $O/pf.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o w -- $R/tools/microbench/mall_probe > $O/pw.log 2>&1
python $R/tools/pmc_dispatches.py $O/pf/f_results.db > $O/mall_fetch_size.txt 2>&1
python $R/tools/pmc_dispatches.py $O/pw/w_results.db > $O/mall_write_size.txt 2>&1
rm -rf $O/pf $O/pw
cd $R
(timeout 900 python -m pytest tests/test_gpu_knn_reg.py -m gpu -q -k knn 2>&1 | tail -8 > $O/pytest_knn.log)
timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity > $O/bench_c1_s1_b4.json 2> $O/bench_c1_s1_b4.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/kt1 -o kt -- python $R/bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 5 > $O/kt1.log 2>&1
python $R/tools/rocpd_summary.py $O/kt1/kt_results.db $O/kernel_stats_s1_b4.csv
rm -rf $O/kt1
cat $O/mall_probe.txt; cat $O/pytest_knn.log; grep -i knn $O/kernel_stats_s1_b4.csv | cut -c1-50,140-260
