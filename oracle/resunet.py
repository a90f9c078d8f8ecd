"""CPU restatement of `ResUNetBN2C.forward` (3-D FCGF net and 6-D inlier net).

TEST INFRASTRUCTURE (see oracle/__init__.py).  **Parity unpinned** for the
MinkowskiEngine part (see me_semantics.py); topology, channel tables and op
order follow the reference exactly -- pinned by outputs of the reference's own
model classes (tests/golden/make_golden_model.py -> resunet_model.npz,
tests/test_oracle_model_golden.py):

* topology / forward order      `model/resunet.py:419-649`
* channel tables (ResUNetBN2C)  `model/resunet.py:662-665`
* residual block                `model/residual_block.py:83-134`
* conv / conv_tr factories      `model/residual_block.py:15-80`
  (`conv()` never forwards `has_bias` => only `final` has a bias)
* batch norm (eval, eps=1e-5)   `model/common.py:11-21`

The per-conv arithmetic is what ME's CPU backend does: for every kernel offset
k, gather the input rows of the rule, one dense `[P_k,Cin] @ [Cin,Cout]`
product, scatter-add into the output rows.
"""
import numpy as np
import torch

from . import me_semantics as me

CHANNELS = [None, 32, 64, 128, 256]      # model/resunet.py:664
TR_CHANNELS = [None, 64, 64, 64, 128]    # model/resunet.py:665
BN_EPS = 1e-5                            # torch.nn.BatchNorm1d default


class SparseMaps:
    """Coordinate maps + kernel maps of one sparse tensor (one per forward)."""

    def __init__(self, coords, D, conv1_ks):
        self.D = D
        self.coords = {1: np.asarray(coords).astype(np.int32)}
        for ts in (2, 4, 8):
            self.coords[ts] = me.stride_coords(self.coords[ts // 2], ts)
        self._cache = {}
        self.conv1_ks = conv1_ks

    def same(self, ts, ks=3):
        key = ('same', ts, ks)
        if key not in self._cache:
            c = self.coords[ts]
            self._cache[key] = me.kernel_map(c, c, self.D, ks, ts)
        return self._cache[key]

    def down(self, ts_in):
        key = ('down', ts_in)
        if key not in self._cache:
            self._cache[key] = me.kernel_map(self.coords[ts_in], self.coords[2 * ts_in],
                                             self.D, 3, ts_in)
        return self._cache[key]

    def up(self, ts_in):
        """transposed conv: in at ts_in (coarse), out at ts_in/2 (fine)."""
        key = ('up', ts_in)
        if key not in self._cache:
            self._cache[key] = me.transposed_kernel_map(self.coords[ts_in],
                                                        self.coords[ts_in // 2],
                                                        self.D, 3, ts_in // 2)
        return self._cache[key]


def sparse_conv(feat, kmap, kernel, n_out):
    """out[o] += in[i] @ W[k] over the kernel-map pairs (ME CPU algorithm)."""
    k, i, o = kmap
    W = kernel if kernel.dim() == 3 else kernel.unsqueeze(0)
    out = torch.zeros(n_out, W.shape[2], dtype=feat.dtype)
    if len(k) == 0:
        return out
    k_t = torch.from_numpy(k)
    i_t = torch.from_numpy(i)
    o_t = torch.from_numpy(o)
    # pairs are sorted by k: walk the rule boundaries
    bounds = np.flatnonzero(np.diff(k)) + 1
    starts = np.concatenate([[0], bounds])
    ends = np.concatenate([bounds, [len(k)]])
    for s, e in zip(starts, ends):
        kk = int(k[s])
        out.index_add_(0, o_t[s:e], feat[i_t[s:e]] @ W[kk])
    return out


def batch_norm(x, sd, prefix):
    """`ME.MinkowskiBatchNorm` in eval mode == BatchNorm1d on .F with running stats."""
    w, b = sd[prefix + '.bn.weight'], sd[prefix + '.bn.bias']
    m, v = sd[prefix + '.bn.running_mean'], sd[prefix + '.bn.running_var']
    return (x - m) / torch.sqrt(v + BN_EPS) * w + b


def basic_block(x, kmap, sd, prefix, n):
    """`BasicBlockBase.forward`, model/residual_block.py:118-134."""
    out = sparse_conv(x, kmap, sd[prefix + '.conv1.kernel'], n)
    out = torch.relu(batch_norm(out, sd, prefix + '.norm1'))
    out = sparse_conv(out, kmap, sd[prefix + '.conv2.kernel'], n)
    out = batch_norm(out, sd, prefix + '.norm2')
    out = out + x
    return torch.relu(out)


def resunet_forward(sd, coords, feats, D, conv1_ks, normalize_feature, maps=None,
                    return_intermediates=False, dtype=torch.float32):
    """`ResUNet2.forward`, model/resunet.py:598-649.  `sd` is a state_dict
    with MinkowskiEngine key names; coords int32 [N,1+D]; feats f32 [N,Cin].
    Returns the output feature matrix [N,Cout] row-aligned with the input."""
    sd = {k: (torch.as_tensor(v) if not torch.is_tensor(v) else v) for k, v in sd.items()}
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}   # f32 = the reference's arithmetic
    x = torch.as_tensor(np.asarray(feats), dtype=dtype)
    if maps is None:
        maps = SparseMaps(coords, D, conv1_ks)
    n = {ts: len(c) for ts, c in maps.coords.items()}
    inter = {}

    out = sparse_conv(x, maps.same(1, conv1_ks), sd['conv1.kernel'], n[1])
    out = batch_norm(out, sd, 'norm1')
    out_s1 = basic_block(out, maps.same(1), sd, 'block1', n[1])
    inter['s1'] = out_s1
    out = torch.relu(out_s1)

    out = sparse_conv(out, maps.down(1), sd['conv2.kernel'], n[2])
    out = batch_norm(out, sd, 'norm2')
    out_s2 = basic_block(out, maps.same(2), sd, 'block2', n[2])
    inter['s2'] = out_s2
    out = torch.relu(out_s2)

    out = sparse_conv(out, maps.down(2), sd['conv3.kernel'], n[4])
    out = batch_norm(out, sd, 'norm3')
    out_s4 = basic_block(out, maps.same(4), sd, 'block3', n[4])
    inter['s4'] = out_s4
    out = torch.relu(out_s4)

    out = sparse_conv(out, maps.down(4), sd['conv4.kernel'], n[8])
    out = batch_norm(out, sd, 'norm4')
    out_s8 = basic_block(out, maps.same(8), sd, 'block4', n[8])
    inter['s8'] = out_s8
    out = torch.relu(out_s8)

    out = sparse_conv(out, maps.up(8), sd['conv4_tr.kernel'], n[4])
    out = batch_norm(out, sd, 'norm4_tr')
    out = basic_block(out, maps.same(4), sd, 'block4_tr', n[4])
    out_s4_tr = torch.relu(out)
    inter['s4_tr'] = out_s4_tr
    out = torch.cat((out_s4_tr, out_s4), dim=1)          # ME.cat: decoder first

    out = sparse_conv(out, maps.up(4), sd['conv3_tr.kernel'], n[2])
    out = batch_norm(out, sd, 'norm3_tr')
    out = basic_block(out, maps.same(2), sd, 'block3_tr', n[2])
    out_s2_tr = torch.relu(out)
    inter['s2_tr'] = out_s2_tr
    out = torch.cat((out_s2_tr, out_s2), dim=1)

    out = sparse_conv(out, maps.up(2), sd['conv2_tr.kernel'], n[1])
    out = batch_norm(out, sd, 'norm2_tr')
    out = basic_block(out, maps.same(1), sd, 'block2_tr', n[1])
    out_s1_tr = torch.relu(out)
    inter['s1_tr'] = out_s1_tr
    out = torch.cat((out_s1_tr, out_s1), dim=1)

    w = sd['conv1_tr.kernel']
    out = out @ (w if w.dim() == 2 else w[0])              # k=1 conv, no bias, no BN
    out = torch.relu(out)
    w = sd['final.kernel']
    out = out @ (w if w.dim() == 2 else w[0]) + sd['final.bias'].reshape(1, -1)

    if normalize_feature:                                  # resunet.py:643-647
        out = out / (torch.norm(out, p=2, dim=1, keepdim=True) + 1e-8)
    if return_intermediates:
        return out.numpy(), {k: v.numpy() for k, v in inter.items()}
    return out.numpy()


def conv_layer_specs(D, cin, cout, conv1_ks):
    """(name, K, Cin, Cout, has_bn) for every conv of ResUNetBN2C in state_dict order."""
    C, T = CHANNELS, TR_CHANNELS
    k3 = 3 ** D
    specs = [('conv1', conv1_ks ** D, cin, C[1], 'norm1')]

    def block(name, c):
        return [(name + '.conv1', k3, c, c, name + '.norm1'),
                (name + '.conv2', k3, c, c, name + '.norm2')]
    specs += block('block1', C[1])
    specs += [('conv2', k3, C[1], C[2], 'norm2')] + block('block2', C[2])
    specs += [('conv3', k3, C[2], C[3], 'norm3')] + block('block3', C[3])
    specs += [('conv4', k3, C[3], C[4], 'norm4')] + block('block4', C[4])
    specs += [('conv4_tr', k3, C[4], T[4], 'norm4_tr')] + block('block4_tr', T[4])
    specs += [('conv3_tr', k3, C[3] + T[4], T[3], 'norm3_tr')] + block('block3_tr', T[3])
    specs += [('conv2_tr', k3, C[2] + T[3], T[2], 'norm2_tr')] + block('block2_tr', T[2])
    specs += [('conv1_tr', 1, C[1] + T[2], T[1], None), ('final', 1, T[1], cout, None)]
    return specs
