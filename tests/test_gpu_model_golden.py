"""The HIP networks (`dgr_net_create` / `dgr_resunet_forward` through `ops.NetHandle`) against outputs of the
reference's own `ResUNetBN2C` code (tests/golden/make_golden_model.py; same fixtures tests/test_oracle_model_golden.py
holds the oracle to): 3-D FCGF nets with 7^3 and 5^3 first kernels, the 6-D inlier net."""
import numpy as np
import pytest
import torch

from test_oracle_model_golden import CASES, GOLDEN, weights_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('tag', CASES)
def test_hip_forward_equals_the_reference_model(tag):
    from deepglobalregistration_amd import ops
    golden = np.load(GOLDEN)
    sd, (D, cin, cout, ks, normalize) = weights_of(golden, tag)
    net = ops.NetHandle(sd, D, cin, cout, ks, normalize)
    coords = torch.from_numpy(golden[f'{tag}_coords']).cuda()
    feats = torch.from_numpy(golden[f'{tag}_feats']).cuda()
    out = net.forward(coords, feats).cpu().numpy()
    ref = golden[f'{tag}_out']
    assert out.shape == ref.shape and np.isfinite(out).all()
    err = float(np.abs(out.astype(np.float64) - ref).max() / np.abs(ref).max())
    print(f'{tag}: max |HIP - reference model| / max |reference| = {err:.1e}')
    assert err < 1e-4, (tag, err)                     # north_star's tolerance for features / logits
