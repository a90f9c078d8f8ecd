# round 5, run 11: kNN packing with one atomic per wave for the largest reference norm; kNN tests, one-stream line, kernel stats
set -x
cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/r5_11; mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_knn_reg.py tests/test_gpu_pipeline.py -m gpu -q 2>&1 | tail -5 > $O/pytest.log)
timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity > $O/bench_c1_s1_b4.json 2> $O/bench_c1_s1_b4.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/kt1 -o kt -- python $R/bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 5 > $O/kt1.log 2>&1
python $R/tools/rocpd_summary.py $O/kt1/kt_results.db $O/kernel_stats_s1_b4.csv
rm -rf $O/kt1
cat $O/pytest.log; grep -i "knn_pack" $O/kernel_stats_s1_b4.csv | cut -c1-30,150-260
python -c "import json;j=json.loads(open('$O/bench_c1_s1_b4.json').read().strip().splitlines()[-1]);print('s1_b4', j['value'], j['stage_ms_per_batch'])"
