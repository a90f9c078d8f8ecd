#!/bin/bash
# run 30 (branch dense-ring): the dense-tile kernel with its weights through a register ring (no LDS staging, no barrier)
O=gpurun_out/run30; mkdir -p $O
timeout 60 python tools/r04_runs/row_order.py $O/ring.json first_occurrence > $O/ring.log 2>&1
python -c "
import json; h=json.load(open('$O/ring.json'))['first_occurrence']; t=h['per_layer_ms']; print('ring', 'fwd', round(h['forward_ms'],3), 'L0 64->64', t[19], t[20], 'L1 64->64', t[4], t[5], t[16], t[17], 'L0 32->32', t[1], t[2])"
timeout 100 python -m pytest tests/test_gpu_dense_conv.py -m gpu -x -q 2>&1 | tail -2
