# round 5, run 7: stream x batch sweep at the current kernels (exchange slots padded to 512 B), o3d + f64 hand-over test
set -x
cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/r5_7; mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_o3d.py tests/test_gpu_knn_reg.py -m gpu -q 2>&1 | tail -8 > $O/pytest.log)
for cfg in "1 4" "2 4" "2 6" "3 4" "3 6" "4 3" "4 4" "2 8"; do set -- $cfg; timeout 200 python bench.py --streams $1 --pairs-per-step $2 --no-parity --steps 30 > $O/bench_s$1_b$2.json 2> $O/bench_s$1_b$2.err; done
cat $O/pytest.log
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_s*.json')):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value'],1), round(j['ms_per_step'],2), j['stage_ms_per_batch']['registration'], j['roofline']['c_le_64_frac'])
PY
