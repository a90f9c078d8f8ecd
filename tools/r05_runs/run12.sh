# round 5, run 12: sanity at the final HEAD -- smoke(), the driver's no-flag line, the C-ABI / pipeline / register e2e tests
set -x
cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/r5_12; mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1)
(timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_register_e2e.py tests/test_gpu_o3d.py tests/test_gpu_shared_weights.py tests/test_gpu_knn_reg.py -m gpu -q 2>&1 | tail -5 > $O/pytest.log)
timeout 900 python bench.py > $O/bench_c1_default.json 2> $O/bench_c1_default.err
grep "smoke ok" $O/smoke.log; cat $O/pytest.log
python -c "import json;j=json.loads(open('$O/bench_c1_default.json').read().strip().splitlines()[-1]);print('default', j['value'], j['ms_per_step'], j['roofline']['c_le_64_frac'], j['roofline']['frac'], j['roofline']['traffic'], j['parity']['within_1e-4'], j['config'])"
