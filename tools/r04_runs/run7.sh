set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/final2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench_c1_default.json 2> $O/bench_c1_default.err
timeout 600 python bench.py --kind outdoor --n-raw 120000 --voxel 0.3 --conv1-ks 5 > $O/bench_c3.json 2> $O/bench_c3.err
(timeout 900 python -m pytest tests/test_gpu_maps.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -5) > $O/pytest_maps.log 2>&1
tail -3 $O/pytest_maps.log; tail -c 1500 $O/bench_c1_default.json
