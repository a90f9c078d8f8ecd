#!/bin/bash
# run 14: the reference-code golden tests on the GPU + smoke + a short default bench at HEAD
mkdir -p gpurun_out/run14
timeout 500 python -m pytest tests/test_gpu_model_golden.py tests/test_gpu_register_e2e.py -m gpu -x -q -s > gpurun_out/run14/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/run14/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/run14/smoke.log 2>&1
timeout 300 python bench.py > gpurun_out/run14/bench.json 2> gpurun_out/run14/bench.err
tail -5 gpurun_out/run14/pytest.log; tail -2 gpurun_out/run14/smoke.log; head -c 600 gpurun_out/run14/bench.json
