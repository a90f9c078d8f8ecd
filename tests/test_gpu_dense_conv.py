"""The dense-tile kernel of the same-stride C <= 64 layers of the 3-D net (conv_dense.hip) against the list-based
output-stationary kernel it replaces there (conv_os.hip, forced by DGR_OS_LISTS=1): both compute, per (output row,
offset), one product row from the same two-f16-piece operands with three MFMAs per 32-channel step and add it, scaled
back, in ascending offset order onto shift + residual (model/resunet.py:598-649).  The dense kernel feeds the input
channels of a 32-channel step to the matrix unit in a different order (its quad-coalesced gather), so the f32 sums inside
one MFMA round differently: the tensors of the FCGF forward must agree to a few f32 ulps of their scale, far below the
1e-4 parity tolerance; the unit-norm output features to 4e-6 (2e-6 until the transposed convs, too, took the dense
gather's channel order: conv_up.hip, round 6).  The list-based kernel is held to the oracle and to f64 by
tests/test_gpu_resunet.py / test_gpu_split_f64.py, and so -- being the default -- is the dense one."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def dumps(tmp_path_factory):
    d = tmp_path_factory.mktemp('dense')
    out = {}
    for name, env in (('dense', {}), ('lists', {'DGR_OS_LISTS': '1'}), ('nosplit', {'DGR_NO_DSPLIT': '1'})):
        e = {k: v for k, v in os.environ.items() if k not in ('DGR_OS_LISTS', 'DGR_EXACT_F32', 'DGR_NO_DSPLIT')}
        e.update(env)
        path = str(d / f'{name}.npz')
        subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'aux', 'fcgf_dump.py'), path], env=e, check=True, timeout=900)
        out[name] = np.load(path)
    return out


def test_each_variant_ran_its_kernels(dumps):
    for net in ('k7', 'k3'):
        kd, kl = dumps['dense'][net + '_kinds'].tolist(), dumps['lists'][net + '_kinds'].tolist()
        # block1 (32 -> 32 twice), block2 (64 -> 64 twice), block3_tr and block2_tr (64 -> 64 twice each): 8 layers
        assert sum(k.startswith('sparse_conv_dense_f16x2') for k in kd) == 8, kd
        # conv4_tr, conv3_tr, conv2_tr: by parity class of the output rows (conv_up.hip)
        assert sum(k.startswith('sparse_conv_up_f16x2') for k in kd) == 3, kd
        assert not any('dense' in k or '_up_' in k for k in kl) and sum(k.startswith('sparse_conv_os') for k in kl) >= 11


def test_every_tensor_agrees_with_the_list_based_kernel(dumps):
    assert dumps['dense']['n'].sum() % 128 != 0     # the last row block is partial
    for net in ('k7', 'k3'):
        for n in ('s1', 's2', 's4', 's8', 's4_tr', 's2_tr', 's1_tr', 'F'):
            a, b = dumps['dense'][f'{net}_{n}'], dumps['lists'][f'{net}_{n}']
            assert np.isfinite(a).all() and np.abs(a).max() > 0 and a.shape == b.shape
            err = float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())
            print(f'{net} {n:6s} max |dense - lists| / max |lists| = {err:.1e}')
            assert err < 4e-6, (net, n, err)


def test_middle_tensors_as_operand_pieces_change_no_bit(dumps):
    """Round 6: the tensor between the two convs of a residual block is written by the first conv's epilogue as the
    second conv's ready-made f16 operand pieces ("dense split rows") and as nothing else.  Same row scale, same pending
    ReLU, same two roundings as the consumer-side split (split.h): every tensor of the forward keeps its bits against the
    build-time switch that turns the hand-over off (DGR_NO_DSPLIT=1)."""
    for net in ('k7', 'k3'):
        kd, kn = dumps['dense'][net + '_kinds'].tolist(), dumps['nosplit'][net + '_kinds'].tolist()
        # the second conv of block1, block2, block3_tr, block2_tr gathers pieces
        assert sum(k.endswith(', ps>') for k in kd) == 4, kd
        assert not any(k.endswith(', ps>') for k in kn), kn
        for n in ('s1', 's2', 's4', 's8', 's4_tr', 's2_tr', 's1_tr', 'F'):
            a, b = dumps['dense'][f'{net}_{n}'], dumps['nosplit'][f'{net}_{n}']
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (net, n)

