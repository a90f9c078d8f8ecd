set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r8; mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_maps.py tests/test_gpu_pipeline.py tests/test_gpu_resunet.py -m gpu -x -q 2>&1 | tail -5) > $O/pytest.log 2>&1
for rep in 1 2; do
timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 30 > $O/bench_s1_b4_$rep.json 2> $O/bench_s1_b4_$rep.err
done
timeout 300 python bench.py --no-parity --steps 50 > $O/bench_s3.json 2> $O/bench_s3.err
tail -3 $O/pytest.log
