#!/bin/bash
# run 17: loader options for the ME conventions: GPU test + a dry run of tools/check_me_conventions.py
mkdir -p gpurun_out/run17
timeout 200 python -m pytest tests/test_gpu_me_conventions.py -m gpu -x -q > gpurun_out/run17/pytest.log 2>&1
timeout 200 python tools/check_me_conventions.py --synthetic > gpurun_out/run17/tool.log 2>&1
tail -4 gpurun_out/run17/pytest.log; tail -8 gpurun_out/run17/tool.log
