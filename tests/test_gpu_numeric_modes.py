"""The conv kernels' two arithmetic modes against each other and against the oracle.

Default: every f32 operand as two f16 pieces under exact power-of-two row / layer scales, three MFMA products per
MAC (conv_wide.hip, conv_os.hip).  DGR_EXACT_F32=1: v_mfma_f32_* on the f32 operands themselves (the reference's
arithmetic).  Both are held to the oracle (CPU f32) at the 1e-4 parity tolerance and to EACH OTHER at 2e-5 -- the
split-operand arithmetic must not be distinguishable from an f32 MFMA chain at the level the parity tests can see
(tests/test_gpu_split_f64.py measures both against f64)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from deepglobalregistration_amd import synth
from oracle import resunet as oresunet

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = {'f16x2': {}, 'f32': {'DGR_EXACT_F32': '1'}}


@pytest.fixture(scope='module')
def dumps(tmp_path_factory):
    d = tmp_path_factory.mktemp('modes')
    out = {}
    for name, env in MODES.items():
        e = {k: v for k, v in os.environ.items() if k != 'DGR_EXACT_F32'}
        e.update(env)
        path = str(d / f'{name}.npz')
        subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'aux', 'net_modes_dump.py'), path], env=e,
                       check=True, timeout=600)
        out[name] = np.load(path)
    return out


def test_every_mode_ran_its_own_kernels(dumps):
    k = {m: dumps[m]['kinds'].tolist() for m in MODES}
    split = lambda n: 'f16x2' in n
    assert any(n.startswith('sparse_conv_wide_f16x2') for n in k['f16x2']) and any(n.endswith('f16x2>') for n in k['f16x2'])
    assert not any(split(n) for n in k['f32']) and any(n.endswith('f32>') for n in k['f32'])


def test_modes_agree_with_each_other_and_the_oracle(dumps):
    ref = dumps['f32']
    sd6 = synth.synth_state_dict(6, 6, 1, 3, 11)
    sd3 = synth.synth_state_dict(3, 1, 32, 7, 0)
    o_logit = np.asarray(oresunet.resunet_forward(sd6, ref['coords6'], ref['feats6'], 6, 3, False))
    o_F = np.asarray(oresunet.resunet_forward(sd3, ref['c3'], np.ones((len(ref['c3']), 1), np.float32), 3, 7, True))
    s_l, s_f = np.abs(o_logit).max(), np.abs(o_F).max()
    for m in MODES:
        d = dumps[m]
        assert np.abs(d['logit'] - o_logit).max() / s_l < 1e-4, m     # the parity tolerance
        assert np.abs(d['F'] - o_F).max() / s_f < 1e-4, m
        assert np.abs(d['logit'] - ref['logit']).max() / s_l < 2e-5, m
        assert np.abs(d['F'] - ref['F']).max() / s_f < 2e-5, m
