# round 5, run 8 (after the evidence bundle): the three heavy phases of the 6-D kernel-map build as ONE launch each for all
# seven maps; registration launches sliced at 64 spinning members (12 full-size pairs in
# ONE batch = two launches: bitwise the results of three batches of 4), smoke(), the tests touched since, the driver's line
set -x
cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/r5_8; mkdir -p $O
export TMPDIR=/tmp
C="--steps 1 --warmup 1 --total-pairs 12 --streams 1 --no-parity"
timeout 300 python bench.py $C --pairs-per-step 12 --dump-results $O/b12.npz > $O/b12.json 2> $O/b12.err
timeout 300 python bench.py $C --pairs-per-step 4 --dump-results $O/b4.npz > $O/b4.json 2> $O/b4.err
python - > $O/slices.txt 2>&1 <<PY
import numpy as np
a, b = np.load('$O/b12.npz'), np.load('$O/b4.npz')
oa, ob = np.argsort(a['ids']), np.argsort(b['ids'])
print('ids equal', np.array_equal(a['ids'][oa], b['ids'][ob]), 'T bitwise equal', np.array_equal(a['T'][oa], b['T'][ob]),
      'stats equal', np.array_equal(a['stats'][oa], b['stats'][ob]), 'iterations', a['stats'][oa][:, 0].tolist())
PY
(timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1)
(timeout 900 python -m pytest tests/test_gpu_maps.py tests/test_gpu_resunet.py tests/test_gpu_model_golden.py tests/test_gpu_o3d.py tests/test_gpu_knn_reg.py tests/test_gpu_pipeline.py tests/test_gpu_bench_ranks.py -m gpu -q 2>&1 | tail -8 > $O/pytest.log)
timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity > $O/bench_c1_s1_b4.json 2> $O/bench_c1_s1_b4.err
timeout 900 python bench.py > $O/bench_c1_default.json 2> $O/bench_c1_default.err
python -c "import json;j=json.loads(open('$O/bench_c1_s1_b4.json').read().strip().splitlines()[-1]);print('s1_b4', j['value'], j['stage_ms_per_batch'])"; cat $O/slices.txt; tail -3 $O/smoke.log; cat $O/pytest.log; tail -c 700 $O/bench_c1_default.json
