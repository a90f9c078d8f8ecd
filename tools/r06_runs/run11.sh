#!/bin/bash
# round 6, run 11: eight ranks on one GPU over gloo (bitwise equal to one rank), the other bench-ranks tests
timeout 2400 python -m pytest tests/test_gpu_bench_ranks.py -m gpu -x -q 2>&1 | tail -6
