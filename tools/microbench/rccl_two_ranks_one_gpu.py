"""Does RCCL accept two ranks on ONE GPU (both processes see the same device)?  If it does, the broadcast / gather of the
multi-GPU path can run with more than one rank on a 1-GPU lease.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 tools/microbench/rccl_two_ranks_one_gpu.py"""
import os, sys, torch, torch.distributed as dist
rank = int(os.environ['RANK'])
torch.cuda.set_device(0)
try:
    dist.init_process_group('nccl', device_id=torch.device('cuda:0'))
    x = torch.full((1024,), float(rank + 1), device='cuda:0')
    dist.all_reduce(x)
    torch.cuda.synchronize()
    b = torch.arange(8, device='cuda:0', dtype=torch.float32) if rank == 0 else torch.zeros(8, device='cuda:0')
    dist.broadcast(b, 0)
    torch.cuda.synchronize()
    print(f'rank {rank}: all_reduce -> {x[0].item()} (expected 3.0), broadcast -> {b.tolist()}', flush=True)
    dist.destroy_process_group()
except Exception as e:   # noqa: BLE001
    print(f'rank {rank}: RCCL refused: {type(e).__name__}: {str(e)[:300]}', flush=True)
    sys.exit(0)
