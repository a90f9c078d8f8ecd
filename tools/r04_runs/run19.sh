#!/bin/bash
# run 19: configs[2], configs[4] and the full-register line once more at the final kernels (no parity leg)
O=gpurun_out/run19; mkdir -p $O
timeout 100 python bench.py --kind outdoor --n-raw 120000 --voxel 0.3 --conv1-ks 5 --no-parity > $O/bench_c3.json 2> $O/c3.err
timeout 100 python bench.py --full-register --no-parity > $O/bench_c1_full_register.json 2> $O/fr.err
timeout 120 python bench.py --n-raw 200000 --voxel 0.025 --pairs-per-step 1 --no-parity > $O/bench_c5.json 2> $O/c5.err
for f in $O/*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],1), d['ms_per_step'])"; done
