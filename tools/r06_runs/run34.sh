#!/bin/bash
# round 6, run 34: two RCCL ranks on the one GPU of the lease?
R=$PWD; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 tools/microbench/rccl_two_ranks_one_gpu.py 2>&1 | grep -v amdgpu.ids | tail -12
