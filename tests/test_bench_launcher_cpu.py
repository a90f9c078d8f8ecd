"""`bench.py --gpus N` is a real launcher: started plainly with N > 1 it re-executes itself under
`torch.distributed.run` with N ranks.  `--launch-check` runs the multi-rank plumbing of the benchmark
(rendezvous, weight broadcast, cost-balanced dealing of pairs, max-over-ranks timing, result gather)
on CPU tensors over gloo, without any GPU work.  CPU only."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = dict(os.environ, DGR_BENCH_BACKEND='gloo')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--launch-check', *flags], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    return json.loads(line)


def test_gpus_2_starts_two_ranks_and_covers_512_pairs():
    out = _run('--gpus', '2', '--total-pairs', '512')
    assert out['n_gpus'] == 2 and out['requested_gpus'] == 2
    assert out['pairs'] == 512 and out['all_pairs_covered_once']
    assert out['max_over_ranks'] == 2.0            # MAX over ranks of (rank + 1)
    assert out['weights_broadcast_keys'] > 100


def test_gpus_8_starts_eight_ranks_weak_and_strong():
    """The node the driver scales to: 8 ranks (gloo on this container's 8 cores).  Weak mode: 8 x S x B pairs, every pair
    registered once; strong mode: BASELINE configs[3], 512 pairs dealt 64 per rank by cost."""
    out = _run('--gpus', '8')
    assert out['n_gpus'] == 8 and out['requested_gpus'] == 8
    assert out["pairs"] == 8 * 4 * 6 and out["all_pairs_covered_once"]     # the defaults: 4 streams x batches of 6 per rank
    assert out['max_over_ranks'] == 8.0 and out['weights_broadcast_keys'] > 100
    out = _run('--gpus', '8', '--total-pairs', '512')
    assert out['n_gpus'] == 8 and out['pairs'] == 512 and out['all_pairs_covered_once']
    assert out['pairs_per_rank'] == [64] * 8


def test_gpus_1_runs_in_process():
    out = _run('--gpus', '1')
    assert out['n_gpus'] == 1 and out['all_pairs_covered_once']


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, DGR_BENCH_BACKEND='gloo', WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--launch-check', '--gpus', '4'], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'launcher started 1 rank' in (r.stderr + r.stdout)


def test_deal_by_cost_balances_and_covers():
    from deepglobalregistration_amd.dist import deal_by_cost
    rng = np.random.default_rng(0)
    cost = rng.uniform(5e8, 8e8, 512)               # N0 * N1 of 512 pairs
    for w in (1, 2, 4, 8):
        shares = deal_by_cost(cost, w)
        assert sorted(i for s in shares for i in s) == list(range(512))
        assert max(len(s) for s in shares) - min(len(s) for s in shares) <= 1
        tot = np.array([cost[s].sum() for s in shares])
        assert tot.max() / tot.min() < 1.002          # contiguous blocks of i.i.d. costs differ by ~1 %
    assert deal_by_cost([3, 1, 2], 2) == [[0], [2, 1]]  # snake: 0 -> rank 0, then ranks 1, 1, 0, 0, 1, ...
