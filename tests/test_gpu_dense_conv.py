"""The dense-tile kernel of the same-stride C <= 64 layers of the 3-D net (conv_dense.hip) against the list-based
output-stationary kernel it replaces there (conv_os.hip, forced by DGR_OS_LISTS=1): both compute, per (output row,
offset), the same product row with the same MFMA sequence and add it in ascending offset order onto shift + residual,
so every tensor of the FCGF forward must come out BIT FOR BIT equal (model/resunet.py:598-649).  The list-based kernel
is held to the oracle and to f64 by tests/test_gpu_resunet.py / test_gpu_split_f64.py; this test carries that over."""
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
    for name, env in (('dense', {}), ('lists', {'DGR_OS_LISTS': '1'})):
        e = {k: v for k, v in os.environ.items() if k not in ('DGR_OS_LISTS', 'DGR_EXACT_F32')}
        e.update(env)
        path = str(d / f'{name}.npz')
        subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'aux', 'fcgf_dump.py'), path], env=e, check=True, timeout=900)
        out[name] = np.load(path)
    return out


def test_each_variant_ran_its_kernels(dumps):
    for net in ('k7', 'k3'):
        kd, kl = dumps['dense'][net + '_kinds'].tolist(), dumps['lists'][net + '_kinds'].tolist()
        # block1 (32 -> 32 twice), block2 (64 -> 64 twice), block2_tr and block1_tr (64 -> 64 twice each): 8 layers
        assert sum(k.startswith('sparse_conv_dense_f16x2') for k in kd) == 8, kd
        assert not any('dense' in k for k in kl) and sum(k.startswith('sparse_conv_os') for k in kl) >= 8


def test_every_tensor_bit_for_bit(dumps):
    assert dumps['dense']['n'].sum() % 128 != 0
    for net in ('k7', 'k3'):
        for n in ('s1', 's2', 's4', 's8', 's4_tr', 's2_tr', 's1_tr', 'F'):
            a, b = dumps['dense'][f'{net}_{n}'], dumps['lists'][f'{net}_{n}']
            assert np.isfinite(a).all() and np.abs(a).max() > 0
            np.testing.assert_array_equal(a, b, err_msg=f'{net} {n}')
