"""Independent arithmetic check of the sparse-conv restatement (CPU only).

`oracle.resunet.sparse_conv` + `oracle.me_semantics.kernel_map / transposed_kernel_map` were written
from recalled MinkowskiEngine semantics (parity unpinned, see oracle/__init__.py).  This file checks
them against code the oracle's author did not write: on a densified voxel grid a sparse convolution
with zero-filled inactive sites is an ordinary cross-correlation, so for D = 3

* same-stride conv (k = 3, 5, 7)            == torch.nn.functional.conv3d(padding = k // 2)
* stride-2 conv (k = 3)                     == conv3d(stride = 2, padding = 1) at the even cells
* transposed conv (k = 3, stride 2)         == conv_transpose3d(stride = 2, padding = 1)

evaluated at the active output sites (`model/residual_block.py:15-80` builds exactly these three
layer kinds for `model/resunet.py:443-566`).  The kernel layout under test is `[K, Cin, Cout]` with
the offset index j = sum_d (delta_d + k//2) k^d, first spatial dimension fastest.
For D = 6 there is no dense torch operator; a shift-and-accumulate formulation over a dense 6-D
array (no coordinate look-ups at all) plays the same role.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

from oracle import me_semantics as me
from oracle import resunet as oresunet


def _cloud(rng, n, lo, hi, D=3, batch=(0,)):
    out = []
    for b in batch:
        p = rng.integers(lo, hi, (n, D))
        _, first = np.unique(p, axis=0, return_index=True)
        p = p[np.sort(first)]
        out.append(np.concatenate([np.full((len(p), 1), b), p], axis=1))
    return np.concatenate(out).astype(np.int32)


def _dense_weight(W, ks, D=3):
    """[K, Cin, Cout] with first-spatial-dimension-fastest offsets -> torch conv layout
    [Cout, Cin, k_x, k_y, k_z] (tensor dims ordered x, y, z)."""
    K, ci, co = W.shape
    w = W.reshape((ks,) * D + (ci, co))              # C order: LAST of the D axes = dimension 0 (fastest)
    w = w.permute(*reversed(range(D)), D, D + 1)     # -> [k_x, k_y, k_z, ci, co]
    return w.permute(D + 1, D, *range(D)).contiguous()


def _densify(coords, feats, lo, shape, b):
    sel = coords[:, 0] == b
    g = torch.zeros((feats.shape[1],) + tuple(shape), dtype=feats.dtype)
    c = coords[sel, 1:] - lo
    g[:, c[:, 0], c[:, 1], c[:, 2]] = feats[sel].T
    return g[None]


@pytest.mark.parametrize('ks', [3, 5, 7])
def test_same_stride_conv_equals_conv3d(ks):
    rng = np.random.default_rng(ks)
    coords = _cloud(rng, 500, -6, 7, batch=(0, 1))            # ~20 % occupancy, negatives, 2 batch elements
    cin, cout = 3, 5
    x = torch.from_numpy(rng.standard_normal((len(coords), cin))).double()
    W = torch.from_numpy(rng.standard_normal((ks ** 3, cin, cout))).double()
    y = oresunet.sparse_conv(x, me.kernel_map(coords, coords, 3, ks, 1), W, len(coords))
    lo, shape = np.array([-6, -6, -6]), (13, 13, 13)
    for b in (0, 1):
        dense = Fn.conv3d(_densify(coords, x, lo, shape, b), _dense_weight(W, ks), padding=ks // 2)[0]
        sel = coords[:, 0] == b
        c = coords[sel, 1:] - lo
        ref = dense[:, c[:, 0], c[:, 1], c[:, 2]].T
        assert torch.allclose(y[sel], ref, rtol=1e-12, atol=1e-12), (ks, b, (y[sel] - ref).abs().max())


@pytest.mark.parametrize('ts', [1, 2])
def test_stride2_conv_equals_strided_conv3d(ts):
    """in at tensor stride ts, out at 2 ts: out[o] = sum_k in[o + delta_k ts] W[k]."""
    rng = np.random.default_rng(10 + ts)
    coords = _cloud(rng, 400, -5, 6, batch=(0, 1))
    coords[:, 1:] *= ts                                        # a level-ts coordinate map
    cout_map = me.stride_coords(coords, 2 * ts)
    cin, cout = 4, 6
    x = torch.from_numpy(rng.standard_normal((len(coords), cin))).double()
    W = torch.from_numpy(rng.standard_normal((27, cin, cout))).double()
    y = oresunet.sparse_conv(x, me.kernel_map(coords, cout_map, 3, 3, ts), W, len(cout_map))
    # dense grid in units of ts; lo even so that output cell q sits at coordinate lo + 2 q
    lo, shape = np.array([-6, -6, -6]), (12, 12, 12)
    for b in (0, 1):
        fine = coords.copy(); fine[:, 1:] //= ts
        dense = Fn.conv3d(_densify(fine, x, lo, shape, b), _dense_weight(W, 3), stride=2, padding=1)[0]
        sel = cout_map[:, 0] == b
        q = (cout_map[sel, 1:] // ts - lo) // 2
        ref = dense[:, q[:, 0], q[:, 1], q[:, 2]].T
        assert torch.allclose(y[sel], ref, rtol=1e-12, atol=1e-12), (y[sel] - ref).abs().max()


@pytest.mark.parametrize('ts', [1, 2])
def test_transposed_conv_equals_conv_transpose3d(ts):
    """in at 2 ts (coarse), out on the existing fine map: out[f] += in[c] W[k], f = c + delta_k ts."""
    rng = np.random.default_rng(20 + ts)
    fine = _cloud(rng, 400, -5, 6, batch=(0, 1))
    fine[:, 1:] *= ts
    coarse = me.stride_coords(fine, 2 * ts)
    cin, cout = 5, 3
    x = torch.from_numpy(rng.standard_normal((len(coarse), cin))).double()
    W = torch.from_numpy(rng.standard_normal((27, cin, cout))).double()
    y = oresunet.sparse_conv(x, me.transposed_kernel_map(coarse, fine, 3, 3, ts), W, len(fine))
    lo = np.array([-6, -6, -6])
    wt = _dense_weight(W, 3).permute(1, 0, 2, 3, 4).contiguous()    # conv_transpose3d wants [Cin, Cout, ...]
    for b in (0, 1):
        cq = coarse.copy(); cq[:, 1:] = (coarse[:, 1:] // ts - lo) // 2
        dense = Fn.conv_transpose3d(_densify(cq, x, np.zeros(3, int), (6, 6, 6), b), wt, stride=2, padding=1,
                                    output_padding=1)[0]           # cell p <-> coordinate lo + p (units of ts)
        sel = fine[:, 0] == b
        p = fine[sel, 1:] // ts - lo
        ref = dense[:, p[:, 0], p[:, 1], p[:, 2]].T
        assert torch.allclose(y[sel], ref, rtol=1e-12, atol=1e-12), (y[sel] - ref).abs().max()


@pytest.mark.parametrize('mode', ['same', 'down', 'up'])
def test_6d_conv_equals_dense_shift_and_accumulate(mode):
    """D = 6 (K = 729): dense 6-D arrays, one shifted slice per offset, no coordinate look-ups."""
    rng = np.random.default_rng({'same': 1, 'down': 2, 'up': 3}[mode])
    D, n = 6, 4                                                # 4^6 = 4096 cells
    fine = _cloud(rng, 700, 0, n, D=D)
    cin, cout = 2, 3
    W = torch.from_numpy(rng.standard_normal((729, cin, cout))).double()
    offs = me.kernel_offsets(D, 3)
    pad = 2

    def dense(coords, feats, scale):
        g = np.zeros((n + 2 * pad,) * D + (feats.shape[1],))
        c = coords[:, 1:] // scale + pad
        g[tuple(c.T)] = feats.numpy()
        return g

    def window(g, delta):                                      # g[cell + delta] for every un-padded cell
        return g[tuple(slice(pad + d, pad + d + n) for d in delta)]

    if mode == 'same':
        x = torch.from_numpy(rng.standard_normal((len(fine), cin))).double()
        y = oresunet.sparse_conv(x, me.kernel_map(fine, fine, D, 3, 1), W, len(fine))
        g = dense(fine, x, 1)
        acc = sum(window(g, offs[k]) @ W[k].numpy() for k in range(729))
        ref = acc[tuple(fine[:, 1:].T)]
    elif mode == 'down':
        coarse = me.stride_coords(fine, 2)
        x = torch.from_numpy(rng.standard_normal((len(fine), cin))).double()
        y = oresunet.sparse_conv(x, me.kernel_map(fine, coarse, D, 3, 1), W, len(coarse))
        g = dense(fine, x, 1)
        acc = sum(window(g, offs[k]) @ W[k].numpy() for k in range(729))     # value at EVERY cell ...
        ref = acc[tuple(coarse[:, 1:].T)]                                      # ... read at the even ones
    else:
        coarse = me.stride_coords(fine, 2)
        x = torch.from_numpy(rng.standard_normal((len(coarse), cin))).double()
        y = oresunet.sparse_conv(x, me.transposed_kernel_map(coarse, fine, D, 3, 1), W, len(fine))
        g = dense(coarse, x, 1)                                               # coarse rows at their own (even) cells
        # out[f] = sum_k in[f - delta_k] W[k]
        acc = sum(window(g, -offs[k]) @ W[k].numpy() for k in range(729))
        ref = acc[tuple(fine[:, 1:].T)]
    np.testing.assert_allclose(y.numpy(), ref, rtol=1e-12, atol=1e-12)


def test_full_forward_3d_equals_dense_torch_network():
    """The whole `ResUNet2.forward` (model/resunet.py:598-649) for D = 3 as a dense torch network with an
    occupancy mask per level, against `oracle.resunet.resunet_forward`."""
    from deepglobalregistration_amd import synth
    rng = np.random.default_rng(5)
    coords = _cloud(rng, 1500, -8, 8)                          # one batch element, 37 % occupancy
    ks, cin, cout = 5, 2, 8
    sd = {k: torch.as_tensor(v).double() if np.asarray(v).dtype.kind == 'f' else torch.as_tensor(v)
          for k, v in synth.synth_state_dict(3, cin, cout, ks, seed=9).items()}
    feats = rng.standard_normal((len(coords), cin))
    ref = oresunet.resunet_forward(sd, coords, feats, 3, ks, True, dtype=torch.float64)

    lo, n = -8, 16
    occ = {1: torch.zeros(1, 1, n, n, n, dtype=torch.float64)}
    c = coords[:, 1:] - lo
    occ[1][0, 0, c[:, 0], c[:, 1], c[:, 2]] = 1
    for ts in (2, 4, 8):                                        # a coarse cell exists iff any of its 8 children does
        occ[ts] = Fn.max_pool3d(occ[ts // 2], 2)
    x = torch.zeros(1, cin, n, n, n, dtype=torch.float64)
    x[0, :, c[:, 0], c[:, 1], c[:, 2]] = torch.from_numpy(feats).T

    def bn(t, p):
        sh = (1, -1, 1, 1, 1)
        return ((t - sd[p + '.bn.running_mean'].reshape(sh)) / torch.sqrt(sd[p + '.bn.running_var'].reshape(sh) + 1e-5)
                * sd[p + '.bn.weight'].reshape(sh) + sd[p + '.bn.bias'].reshape(sh))

    def conv(t, name, k=3, stride=1):
        return Fn.conv3d(t, _dense_weight(sd[name + '.kernel'], k), stride=stride, padding=k // 2)

    def conv_tr(t, name):
        w = _dense_weight(sd[name + '.kernel'], 3).permute(1, 0, 2, 3, 4).contiguous()
        return Fn.conv_transpose3d(t, w, stride=2, padding=1, output_padding=1)

    def block(t, name, m):
        y = torch.relu(bn(conv(t, name + '.conv1'), name + '.norm1')) * m
        y = bn(conv(y, name + '.conv2'), name + '.norm2') * m
        return torch.relu(y + t)

    s1 = block(bn(conv(x, 'conv1', ks), 'norm1') * occ[1], 'block1', occ[1])
    s2 = block(bn(conv(s1, 'conv2', stride=2), 'norm2') * occ[2], 'block2', occ[2])
    s4 = block(bn(conv(s2, 'conv3', stride=2), 'norm3') * occ[4], 'block3', occ[4])
    s8 = block(bn(conv(s4, 'conv4', stride=2), 'norm4') * occ[8], 'block4', occ[8])
    u4 = block(bn(conv_tr(s8, 'conv4_tr'), 'norm4_tr') * occ[4], 'block4_tr', occ[4])
    u2 = block(bn(conv_tr(torch.cat((u4, s4), 1), 'conv3_tr'), 'norm3_tr') * occ[2], 'block3_tr', occ[2])
    u1 = block(bn(conv_tr(torch.cat((u2, s2), 1), 'conv2_tr'), 'norm2_tr') * occ[1], 'block2_tr', occ[1])
    h = torch.cat((u1, s1), 1)
    h = torch.relu(torch.einsum('bcxyz,cd->bdxyz', h, sd['conv1_tr.kernel']))
    h = torch.einsum('bcxyz,cd->bdxyz', h, sd['final.kernel']) + sd['final.bias'].reshape(1, -1, 1, 1, 1)
    out = h[0, :, c[:, 0], c[:, 1], c[:, 2]].T
    out = out / (out.norm(dim=1, keepdim=True) + 1e-8)
    np.testing.assert_allclose(ref, out.numpy(), rtol=1e-9, atol=1e-10)
