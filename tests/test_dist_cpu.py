"""world_size-2 gloo tests of the multi-GPU layer (deepglobalregistration_amd/dist.py): pair
sharding, the one-shot checkpoint broadcast and the result gather.  CPU only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tmpdir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from deepglobalregistration_amd import dist as ddist, synth
    # 1. broadcast: rank 0 owns the checkpoint, everybody ends up with identical tensors
    ck = synth.synth_checkpoint(seed=3, feat_conv1_kernel_size=3, with_inlier=False) if rank == 0 else None
    if rank == 0:
        ck['state_dict_inlier'] = synth.synth_state_dict(3, 6, 1, 3, seed=4)   # small stand-in for the 6-D net
    got = ddist.broadcast_checkpoint(ck, src=0)
    ref = synth.synth_checkpoint(seed=3, feat_conv1_kernel_size=3, with_inlier=False)
    ref['state_dict_inlier'] = synth.synth_state_dict(3, 6, 1, 3, seed=4)
    assert got['config'] == ref['config']
    for name in ('state_dict', 'state_dict_inlier'):
        keys = [k for k in ref[name] if not k.endswith('num_batches_tracked')]
        assert sorted(got[name]) == sorted(keys) or rank == 0
        for k in keys:
            np.testing.assert_array_equal(np.asarray(got[name][k]), ref[name][k])
            assert np.asarray(got[name][k]).shape == ref[name][k].shape
    # 2. sharding: 7 pairs over 2 ranks -> contiguous blocks covering everything exactly once
    lo, hi = ddist.shard_range(7, rank, world)
    # 3. gather: uneven shard sizes
    n = hi - lo
    T = np.tile(np.eye(4), (n, 1, 1))
    T[:, 0, 3] = np.arange(lo, hi)
    status = (np.arange(lo, hi) % 2).astype(np.int32)
    stats = np.stack([np.arange(lo, hi)] * 4, axis=1).astype(np.float32)
    out = ddist.gather_results(T, status, stats, dst=0)
    if rank == 0:
        Ta, sa, sta = out
        assert Ta.shape == (7, 4, 4) and sa.shape == (7,) and sta.shape == (7, 4)
        np.testing.assert_array_equal(Ta[:, 0, 3], np.arange(7))
        np.testing.assert_array_equal(sa, np.arange(7) % 2)
        np.testing.assert_array_equal(sta[:, 2], np.arange(7))
        open(os.path.join(tmpdir, 'ok'), 'w').write('ok')
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_shard_gather(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / 'ok').exists()


def test_shard_range_partitions():
    from deepglobalregistration_amd.dist import shard_range
    for n in (0, 1, 7, 64, 512, 513):
        for w in (1, 2, 3, 8):
            blocks = [shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(b[1] == blocks[i + 1][0] for i, b in enumerate(blocks[:-1]))
            sizes = [b[1] - b[0] for b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_passthrough():
    from deepglobalregistration_amd import dist as ddist
    ck = {'config': {'a': 1}, 'state_dict': {}}
    assert ddist.broadcast_checkpoint(ck) is ck
    T, s, st = ddist.gather_results(np.eye(4)[None], [0], np.zeros((1, 4)))
    assert T.shape == (1, 4, 4) and s.tolist() == [0]


def _forced_worker(rank, world, port, tmpdir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['DGR_DIST_FORCE_COLLECTIVES'] = '1'
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from deepglobalregistration_amd import dist as ddist, synth
    ck = synth.synth_checkpoint(seed=5, feat_conv1_kernel_size=3, with_inlier=False)
    got = ddist.broadcast_checkpoint(ck, src=0)
    assert got is not ck and got['config'] == ck['config']      # rebuilt from the broadcast buffer, not passed through
    for k, v in ck['state_dict'].items():
        if not k.endswith('num_batches_tracked'):
            np.testing.assert_array_equal(np.asarray(got['state_dict'][k]), np.asarray(v, np.float32))
    T, s, st = ddist.gather_results(np.eye(4)[None] * 3, [2], np.full((1, 4), 7.0))
    assert T.shape == (1, 4, 4) and T[0, 0, 0] == 3 and s.tolist() == [2] and st[0, 1] == 7
    v = ddist.all_gather_vector([1.5, 2.5], 4, 1)
    np.testing.assert_array_equal(v, [0, 1.5, 2.5, 0])
    open(os.path.join(tmpdir, 'ok'), 'w').write('ok')
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_group_runs_the_collectives_when_forced(tmp_path):
    """bench.py's DGR_BENCH_FORCE_PG mode (tests/test_gpu_bench_ranks.py runs it over RCCL): with a one-rank group and
    DGR_DIST_FORCE_COLLECTIVES=1 every helper goes through its collective instead of the single-process shortcut."""
    mp.spawn(_forced_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / 'ok').exists()
