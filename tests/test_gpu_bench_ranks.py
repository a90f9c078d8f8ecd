"""`bench.py --gpus 2` end to end on ONE GPU: both ranks on device 0 (DGR_BENCH_ONE_GPU=1), the tiny collectives over
gloo (DGR_BENCH_BACKEND=gloo; a 1-GPU box cannot run RCCL between two ranks) -- the launcher, the weight broadcast,
the cost-balanced dealing, the per-rank registration and the result gather, with real GPU work.  Every pair must be
covered exactly once with the status and BIT FOR BIT the pose, iteration count and loss of the 1-rank run (pairs are
independent units: SURVEY.md 8e, core/deep_global_registration.py:238-324 has no cross-pair state).

History: two processes sharing one GPU is the harder case for reproducibility.  Until round 3 the registration kernel
kept its per-thread partial sums in f32; that build is bitwise reproducible while a process has the GPU to itself, but
next to a second process 4 % of its runs ended a few ulps .. 2e-5 away (tools/contention_reg.sh: 126 and 115 of 3000
runs; the networks' outputs stayed bitwise identical).  With f64 partial sums (the default now, reg.hip) 0 of 3000,
twice.  DESIGN.md section 7."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ['--steps', '2', '--warmup', '1', '--total-pairs', '6', '--pairs-per-step', '2', '--streams', '1', '--n-raw', '12000',
          '--conv1-ks', '5', '--no-parity']


def _bench(tmp, gpus, tag='', **extra_env):
    env = dict(os.environ, DGR_BENCH_BACKEND='gloo', DGR_BENCH_ONE_GPU='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'DGR_BENCH_FORCE_PG', 'DGR_DIST_FORCE_COLLECTIVES'):
        env.pop(k, None)
    env.update(extra_env)
    for k in [k for k, v in env.items() if v is None]:
        env.pop(k)
    out = os.path.join(tmp, f'res{gpus}{tag}.npz')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(gpus), '--dump-results', out, *COMMON],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    return line, np.load(out), r.stderr


@pytest.fixture(scope='module')
def plain_run(tmp_path_factory):
    """The plain one-rank run (no process group) both tests compare with."""
    return _bench(str(tmp_path_factory.mktemp('plain')), 1)


def test_two_ranks_cover_every_pair_once_and_match_one_rank(tmp_path, plain_run):
    one, r1, _ = plain_run
    two, r2, _ = _bench(str(tmp_path), 2)
    assert one['n_gpus'] == 1 and two['n_gpus'] == 2 and two['scaling'] == 'strong'
    assert two['config']['pairs_per_step'] == 6 and two['value'] > 0
    assert sorted(r1['ids'].tolist()) == list(range(6)) and sorted(r2['ids'].tolist()) == list(range(6))
    o1, o2 = np.argsort(r1['ids']), np.argsort(r2['ids'])
    np.testing.assert_array_equal(r1['status'][o1], r2['status'][o2])
    np.testing.assert_array_equal(r1['T'][o1], r2['T'][o2])
    np.testing.assert_array_equal(r1['stats'][o1], r2['stats'][o2])    # iterations, loss, break count, sum of weights
    assert two['host_cpu_s_per_step_per_rank'] > 0


def test_eight_ranks_on_one_gpu_cover_every_pair_once_and_match_one_rank(tmp_path):
    """The node size the driver scales to, as far as a 1-GPU box can go: EIGHT ranks (eight processes, eight library
    contexts, eight copies of the weights) on device 0 over gloo, 16 pairs dealt by cost -- launcher, broadcast, dealing,
    gather at world size 8 with real GPU work, next to seven other processes on the same GPU.  Every pair covered once and
    bitwise the 1-rank results."""
    common8 = [a if a != '6' else '16' for a in COMMON]          # --total-pairs 16: two per rank
    def run(gpus, tag):
        env = dict(os.environ, DGR_BENCH_BACKEND='gloo', DGR_BENCH_ONE_GPU='1')
        for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'DGR_BENCH_FORCE_PG', 'DGR_DIST_FORCE_COLLECTIVES'):
            env.pop(k, None)
        out = os.path.join(str(tmp_path), f'res{tag}.npz')
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(gpus), '--dump-results', out, *common8],
                           env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1]), np.load(out)
    one, r1 = run(1, 'a')
    eight, r8 = run(8, 'b')
    assert eight['n_gpus'] == 8 and eight['config']['pairs_per_step'] == 16 and eight['scaling'] == 'strong'
    assert sorted(r1['ids'].tolist()) == list(range(16)) and sorted(r8['ids'].tolist()) == list(range(16))
    o1, o8 = np.argsort(r1['ids']), np.argsort(r8['ids'])
    np.testing.assert_array_equal(r1['status'][o1], r8['status'][o8])
    np.testing.assert_array_equal(r1['T'][o1], r8['T'][o8])
    np.testing.assert_array_equal(r1['stats'][o1], r8['stats'][o8])


def test_rccl_collectives_on_a_one_rank_group_match_the_plain_run(tmp_path, plain_run):
    """The RCCL half of the multi-GPU path on the hardware a 1-GPU box has: DGR_BENCH_FORCE_PG=1 makes the one-rank run
    call init_process_group('nccl', device_id=...), broadcast_object_list, the flat weight broadcast (the networks are
    then built from what came back OUT of the broadcast buffer), the cost all-reduce of the strong mode, barriers, the
    result all-gather and the MAX all-reduce of the step time -- every collective `bench.py --gpus 8` issues
    (SURVEY.md 8e), on device memory, through RCCL.  Results must equal the plain one-rank run bit for bit."""
    plain, r1, _ = plain_run
    forced, r2, err = _bench(str(tmp_path), 1, tag='pg', DGR_BENCH_BACKEND=None, DGR_BENCH_FORCE_PG='1')
    assert 'nccl process group up' in err, err[-2000:]
    assert forced['n_gpus'] == 1 and forced['config']['pairs_per_step'] == 6
    # ... and the networks were built ON THE DEVICE from the broadcast buffer (dgr_net_create_device), bit-identical to
    # the plain run's host-prepared weights (the equalities below)
    assert forced['startup_s']['weights_prepared_on_device_this_rank'] and not plain['startup_s']['weights_prepared_on_device_this_rank']
    o1, o2 = np.argsort(r1['ids']), np.argsort(r2['ids'])
    assert sorted(r2['ids'].tolist()) == list(range(6))
    np.testing.assert_array_equal(r1['status'][o1], r2['status'][o2])
    np.testing.assert_array_equal(r1['T'][o1], r2['T'][o2])
    np.testing.assert_array_equal(r1['stats'][o1], r2['stats'][o2])
