"""Loader options for the two MinkowskiEngine conventions a checkpoint depends on
(deepglobalregistration_amd/model/me_conventions.py).  What is checked: IF MinkowskiEngine enumerated kernel offsets
with the last axis fastest, or paired a transposed convolution's kernel index with the mirrored offset, a checkpoint
written under that convention and converted by the loader gives the library's reading exactly the network ME would
have run -- the alternative ME is the oracle with the convention flipped in its one function (oracle/me_semantics.py
A5 / A6).  Nothing here says which convention ME has (unpinned)."""
import numpy as np
import pytest

from deepglobalregistration_amd import synth
from deepglobalregistration_amd.model import me_conventions as mc
from oracle import me_semantics as me
from oracle import resunet as oresunet


def offsets_last_axis_fastest(D, ks):
    K = ks ** D
    offs = np.zeros((K, D), np.int64)
    j = np.arange(K)
    for d in range(D - 1, -1, -1):
        offs[:, d] = (j % ks) - ks // 2
        j = j // ks
    return offs


def transposed_mirrored(default_map):
    def fn(coords_coarse, coords_fine, D, ks, ts_fine):
        k, i, o = default_map(coords_coarse, coords_fine, D, ks, ts_fine)
        k = ks ** D - 1 - k
        order = np.lexsort((o, k))
        return k[order], i[order], o[order]
    return fn


def small_input(D, rng):
    n = 260 if D == 3 else 120
    pts = np.unique(np.floor(rng.normal(0, 2.5, (n, D))).astype(np.int32), axis=0)
    coords = np.c_[np.zeros(len(pts), np.int32), pts]
    return coords, rng.standard_normal((len(pts), 3 if D == 3 else 6)).astype(np.float32)


@pytest.mark.parametrize('D,ks', [(3, 5), (6, 3)])
@pytest.mark.parametrize('order', mc.KERNEL_ORDERS)
@pytest.mark.parametrize('mirrored', [False, True])
def test_converted_checkpoint_runs_the_network_the_other_me_would(monkeypatch, D, ks, order, mirrored):
    rng = np.random.default_rng(10 * D + ks)
    cin = 3 if D == 3 else 6
    sd = synth.synth_state_dict(D, cin, 8, ks, seed=41)                 # "written under the checkpoint's convention"
    coords, feats = small_input(D, rng)
    with monkeypatch.context() as m:                                     # the ME that has that convention
        if order == 'last_axis_fastest':
            m.setattr(me, 'kernel_offsets', offsets_last_axis_fastest)
        if mirrored:
            m.setattr(me, 'transposed_kernel_map', transposed_mirrored(me.transposed_kernel_map))
        want = oresunet.resunet_forward(sd, coords, feats, D, ks, False)
    got = oresunet.resunet_forward(mc.convert_state_dict(sd, D, order, mirrored), coords, feats, D, ks, False)
    err = float(np.abs(got - want).max() / np.abs(want).max())
    assert err < 5e-6, err
    if order != 'first_axis_fastest' or mirrored:                        # ... and the conventions are not equivalent
        plain = oresunet.resunet_forward(sd, coords, feats, D, ks, False)
        assert np.abs(plain - want).max() / np.abs(want).max() > 1e-2


def test_conversion_touches_what_it_should():
    sd = synth.synth_state_dict(3, 1, 32, 7, seed=2)
    assert mc.convert_state_dict(sd, 3) is sd                            # defaults: the identity, no copies
    both = mc.convert_state_dict(sd, 3, 'last_axis_fastest', True)
    back = mc.convert_state_dict(both, 3, 'last_axis_fastest', True)     # both re-indexings are involutions
    for name, v in sd.items():
        assert np.array_equal(back[name], v), name
        changed = not np.array_equal(both[name], v)
        assert changed == (name.endswith('.kernel') and np.ndim(v) == 3), name
    only_tr = mc.convert_state_dict(sd, 3, 'first_axis_fastest', True)
    assert sorted(n for n in sd if not np.array_equal(only_tr[n], sd[n])) == ['conv2_tr.kernel', 'conv3_tr.kernel', 'conv4_tr.kernel']
    with pytest.raises(ValueError):
        mc.convert_kernel(np.zeros((27, 2, 2)), 3, 'middle_out')
    with pytest.raises(ValueError):
        mc.convert_kernel(np.zeros((28, 2, 2)), 3, 'last_axis_fastest')


def test_model_applies_the_stated_convention_on_load():
    """`ResUNetBN2C.me_conventions` is read by `load_state_dict` (no GPU needed: the handle is created lazily)."""
    from deepglobalregistration_amd.model import load_model
    sd = synth.synth_state_dict(3, 1, 32, 3, seed=3)
    net = load_model('ResUNetBN2C')(1, 32, conv1_kernel_size=3, normalize_feature=True, D=3)
    assert net.me_conventions == mc.DEFAULT
    net.load_state_dict(sd)
    assert np.array_equal(net._state['conv2.kernel'], sd['conv2.kernel'])
    net.me_conventions = {'kernel_order': 'last_axis_fastest', 'transposed_mirrored': True}
    net.load_state_dict(sd)
    want = mc.convert_state_dict(sd, 3, 'last_axis_fastest', True)
    assert all(np.array_equal(net._state[k], want[k]) for k in sd)
    assert not np.array_equal(net._state['conv2.kernel'], sd['conv2.kernel'])
