"""Multi-GPU sharding of pair batches: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

`register()` touches no cross-pair state (core/deep_global_registration.py:238-324), so pairs are
independent units: every rank registers a contiguous block of pairs and there is no collective
on the data path.  Exactly two collectives exist:

* `broadcast_checkpoint` -- rank 0's weights (~0.94 GB f32) are broadcast ONCE as a single flat
  buffer (one large ring broadcast is per-link bound on xGMI; many small ones would be latency bound);
* `gather_results`      -- per batch, `T [n,4,4]`, `status [n]`, `stats [n,4]` are gathered on rank 0
  (a few KB).
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def _skip_collectives():
    """No process group, or a single rank: nothing to exchange -- unless DGR_DIST_FORCE_COLLECTIVES=1 asks for the
    collectives anyway (bench.py with DGR_BENCH_FORCE_PG=1: a 1-GPU box executes the RCCL broadcast / all-gather /
    all-reduce of the multi-GPU path on a one-rank communicator, tests/test_gpu_bench_ranks.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size() == 1 and os.environ.get('DGR_DIST_FORCE_COLLECTIVES') != '1'


def shard_range(n_items, rank, world_size):
    """Contiguous block [lo, hi) of `n_items` for `rank`; blocks differ in size by at most one."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _flatten_state(sd):
    keys = [k for k in sd if not k.endswith('num_batches_tracked')]
    metas = [(k, tuple(np.asarray(sd[k]).shape)) for k in keys]
    flat = np.concatenate([np.asarray(sd[k], np.float32).reshape(-1) for k in keys]) if keys else np.zeros(0, np.float32)
    return metas, flat


def _unflatten_state(metas, flat):
    """`flat`: a numpy array or a torch tensor (the broadcast buffer itself, on whatever device it lives): the entries of
    the returned dict are VIEWS of it."""
    out, pos = {}, 0
    for k, shape in metas:
        n = int(np.prod(shape)) if len(shape) else 1
        out[k] = flat[pos:pos + n].reshape(shape)
        pos += n
    return out


def broadcast_checkpoint(ckpt, src=0, device=None):
    """Broadcast a checkpoint dict {'config','state_dict','state_dict_inlier'} from `src`.
    Non-source ranks pass ckpt=None.  The tensors travel as ONE flat float32 buffer per network."""
    if _skip_collectives():
        return ckpt
    rank = dist.get_rank()
    roundtrip = dist.get_world_size() == 1   # forced one-rank run: use what came back out of the broadcast buffer
    backend = dist.get_backend()
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    head = [None]
    flats = {}
    if rank == src:
        metas = {}
        for name in ('state_dict', 'state_dict_inlier'):
            if name in ckpt:
                metas[name], flats[name] = _flatten_state(ckpt[name])
        head = [{'config': ckpt['config'], 'metas': metas, 'sizes': {k: int(v.size) for k, v in flats.items()}}]
    dist.broadcast_object_list(head, src=src)
    head = head[0]
    out = {'config': head['config']}
    for name, metas in head['metas'].items():
        n = head['sizes'][name]
        if rank == src:
            buf = torch.from_numpy(flats[name]).to(device)
        else:
            buf = torch.empty(n, dtype=torch.float32, device=device)
        dist.broadcast(buf, src=src)
        if rank == src and not roundtrip:
            out[name] = ckpt[name]
        elif buf.is_cuda:
            # RCCL: the weights stay where the broadcast put them -- views of the flat device buffer, which the networks'
            # loader hands to dgr_net_create_device (no D2H copy, no host-side weight preparation on the receiving ranks)
            out[name] = _unflatten_state(metas, buf)
        else:
            out[name] = _unflatten_state(metas, buf.numpy())
        del buf
    return out


def gather_results(T, status, stats, dst=0, device=None):
    """Gather per-rank results on `dst`.  Ranks may hold different numbers of pairs.  Returns
    (T [sum,4,4] float64, status [sum] int32, stats [sum,4] float32) on dst, None elsewhere."""
    T = np.asarray(T, np.float64).reshape(-1, 16)
    status = np.asarray(status, np.int32).reshape(-1)
    stats = np.asarray(stats, np.float32).reshape(-1, 4)
    if _skip_collectives():
        return T.reshape(-1, 4, 4), status, stats
    world, rank = dist.get_world_size(), dist.get_rank()
    backend = dist.get_backend()
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    counts[rank] = len(status)
    dist.all_reduce(counts)
    nmax = int(counts.max().item())
    rec = torch.zeros((nmax, 21), dtype=torch.float64, device=device)   # 16 T + status + 4 stats
    if len(status):
        rec[:len(status), :16] = torch.from_numpy(T).to(device)
        rec[:len(status), 16] = torch.from_numpy(status.astype(np.float64)).to(device)
        rec[:len(status), 17:] = torch.from_numpy(stats.astype(np.float64)).to(device)
    bufs = [torch.zeros_like(rec) for _ in range(world)]
    dist.all_gather(bufs, rec)          # a few KB; all_gather keeps it backend-agnostic
    if rank != dst:
        return None
    rows = torch.cat([b[:int(c)] for b, c in zip(bufs, counts.tolist())]).cpu().numpy()
    return rows[:, :16].reshape(-1, 4, 4), rows[:, 16].astype(np.int32), rows[:, 17:].astype(np.float32)


def deal_by_cost(costs, world_size):
    """Static load balancing of independent units: sort the units by decreasing cost and deal them
    round-robin in snake order (rank 0..W-1, then W-1..0, ...), so that every rank gets the same
    number of units (+-1) and nearly the same total cost.  `costs[i]` is the cost proxy of unit i
    (N0 * N1 of a pair: the brute-force 1-NN and the 6-D maps scale with it).  Returns a list of
    `world_size` lists of unit indices, each in decreasing-cost order; identical on every rank."""
    costs = np.asarray(costs, np.float64).reshape(-1)
    order = np.lexsort((np.arange(len(costs)), -costs))       # ties: smaller index first (deterministic)
    shares = [[] for _ in range(world_size)]
    for j, u in enumerate(order.tolist()):
        r = j % (2 * world_size)
        shares[r if r < world_size else 2 * world_size - 1 - r].append(u)
    return shares


def all_gather_vector(values, total, lo, device=None):
    """Every rank contributes `values` for the contiguous unit block starting at `lo`; returns the
    full [total] float64 vector on every rank (one small all-reduce of a zero-padded vector)."""
    full = np.zeros(total, np.float64)
    full[lo:lo + len(values)] = np.asarray(values, np.float64)
    if _skip_collectives():
        return full
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.from_numpy(full).to(device)
    dist.all_reduce(t)
    return t.cpu().numpy()
