"""Golden vectors from the reference's OWN model code: `model/resunet.py::ResUNetBN2C` (with `model/residual_block.py`,
`model/common.py`) imported unchanged from /root/reference and executed on the CPU, with `tests/golden/me_stub` standing
in for the MinkowskiEngine import the container does not have.  Run in the build container only:

    python tests/golden/make_golden_model.py          # writes tests/golden/resunet_model.npz  (~1 minute)

What this pins (tests/test_oracle_model_golden.py, tests/test_gpu_resunet.py): the network TOPOLOGY the oracle and the
HIP net restate -- which layer follows which, where the norms and ReLUs sit, residual adds, the order of `ME.cat`, the
1x1 tail, the bias, feature normalisation -- and the STATE-DICT LAYOUT: the synthetic weights are loaded into the
reference model with `load_state_dict(strict=True)` exactly as `core/deep_global_registration.py:96-129` loads a
checkpoint, so every key name and shape `synth.synth_state_dict` produces is one the reference's modules declare.

What it does not pin: MinkowskiEngine's own arithmetic (coordinate maps, kernel-offset order).  The stub restates that
independently of `oracle/me_semantics.py` (dictionaries and Python loops instead of sorted-key searches), which makes
two implementations of one reading of ME 0.5.4 agree; it stays "parity unpinned" against ME itself.

The weights are not stored (0.94 GB for the 6-D net): the generator and the tests both call `synth.synth_state_dict`
with the seeds below; a float64 checksum over all parameters is stored and re-checked.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get('DGR_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

CASES = {                                    # tag: (D, Cin, Cout, conv1 kernel size, normalize_feature, weight seed)
    'fcgf_k7': (3, 1, 32, 7, True, 21),      # the 3DMatch FCGF checkpoint's shape (feat_conv1_kernel_size 7)
    'fcgf_k5': (3, 1, 16, 5, True, 22),      # the KITTI one (kernel 5, 16-d features)
    'inlier6': (6, 6, 1, 3, False, 23),      # the inlier net, feature type 'coords'
}


def reference_model_module():
    """`model.resunet` of the reference WITHOUT running `model/__init__.py` (which imports two model families that use
    more of ME than the stub has): a bare package object whose path is the reference's directory."""
    sys.path.insert(0, os.path.join(HERE, 'me_stub'))
    pkg = types.ModuleType('model')
    pkg.__path__ = [os.path.join(REF, 'model')]
    sys.modules['model'] = pkg
    return importlib.import_module('model.resunet')


def checksum(sd):
    return float(sum(np.asarray(v, np.float64).sum() for k, v in sorted(sd.items())))


def inputs(tag, D, cin, rng):
    """Small inputs shaped like the path's: two batch entries, negative coordinates, duplicates removed."""
    if D == 3:
        pts = np.concatenate([rng.normal(0, 3.0, (260, 3)), rng.normal(2, 1.5, (160, 3))])
        c3 = np.unique(np.floor(pts).astype(np.int32), axis=0)
        rng.shuffle(c3)
        half = len(c3) // 2
        coords = np.concatenate([np.c_[np.zeros(half, np.int32), c3[:half]],
                                 np.c_[np.ones(len(c3) - half, np.int32), c3[half:] + np.int32([1, -2, 0])]])
        feats = np.ones((len(coords), cin), np.float32)                    # core/deep_global_registration.py:155-159
    else:
        # correspondences (voxel of fragment 0, voxel of fragment 1) along a surface patch, as `register()` builds them
        # (:261-262): neighbouring pairs share neighbouring voxels on both sides
        a = np.unique(np.floor(rng.normal(0, 3.0, (170, 3))).astype(np.int32), axis=0)
        b = a + rng.integers(-1, 2, a.shape).astype(np.int32) + np.int32([3, -1, 2])
        c6 = np.unique(np.c_[a, b], axis=0)
        rng.shuffle(c6)
        coords = np.c_[np.zeros(len(c6), np.int32), c6]
        feats = np.cos(c6.astype(np.float32) * np.float32(0.05)).astype(np.float32)   # 'coords' features, :199-201
    return np.ascontiguousarray(coords, dtype=np.int32), feats


def main():
    sys.path.insert(0, ROOT)
    from deepglobalregistration_amd import synth
    resunets = reference_model_module()
    import MinkowskiEngine as ME
    assert 'me_stub' in ME.__file__
    torch.set_num_threads(1)
    rng = np.random.default_rng(4321)
    out = {}
    for tag, (D, cin, cout, ks, normalize, seed) in CASES.items():
        sd = synth.synth_state_dict(D, cin, cout, ks, seed)
        net = resunets.ResUNetBN2C(cin, cout, bn_momentum=0.05, conv1_kernel_size=ks, normalize_feature=normalize, D=D)
        net.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
        net.eval()
        coords, feats = inputs(tag, D, cin, rng)
        taps = {}
        hooks = [getattr(net, name).register_forward_hook(lambda m, i, o, name=name: taps.__setitem__(name, o.F.detach().numpy().copy()))
                 for name in ('block1', 'block4', 'block4_tr', 'block2_tr', 'conv1_tr')]
        with torch.no_grad():
            y = net(ME.SparseTensor(torch.from_numpy(feats), coordinates=torch.from_numpy(coords)))
        for h in hooks:
            h.remove()
        assert np.array_equal(y.C.numpy(), coords)                           # rows stay aligned with the input
        keys = sorted(net.state_dict().keys())
        out.update({f'{tag}_coords': coords, f'{tag}_feats': feats, f'{tag}_out': y.F.numpy(),
                    f'{tag}_spec': np.array([D, cin, cout, ks, int(normalize), seed], np.int64),
                    f'{tag}_weights_checksum': np.float64(checksum(sd)),
                    f'{tag}_keys': np.array(keys),
                    f'{tag}_shapes': np.array([','.join(map(str, net.state_dict()[k].shape)) for k in keys])})
        # activations on the way (full resolution only: coarse rows are in the stub's order, not a contract)
        for name in ('block1', 'block2_tr', 'conv1_tr'):
            out[f'{tag}_{name}'] = taps[name]
        out[f'{tag}_n_coarse'] = np.array([len(taps['block4']), len(taps['block4_tr'])], np.int64)   # rows at strides 8, 4
        print(f"{tag}: N = {len(coords)}, rows at stride 8 / 4 = {out[f'{tag}_n_coarse'].tolist()}, "
              f"max |out| = {np.abs(y.F.numpy()).max():.3f}, {len(keys)} state-dict entries")
    np.savez_compressed(os.path.join(HERE, 'resunet_model.npz'), **out)


if __name__ == '__main__':
    main()
