"""The split-operand conv kernels (two f16 pieces per f32 operand, three MFMA products per MAC: conv_wide.hip,
conv_os.hip, conv_dense.hip) against float64 truth, next to the exact-f32 MFMA kernels (DGR_EXACT_F32=1) on the same inputs.

Every device result is produced by the real layer kernels -- through `dgr_resunet_forward` for (a) and (b), through
`dgr_debug_conv_layer` for (c) -- in two processes (the arithmetic mode is fixed when a net is built) and compared with
an f64 forward of the oracle (`oracle.resunet.resunet_forward(dtype=float64)`, model/resunet.py:598-649,
model/common.py:11-21) or an f64 numpy evaluation of the one layer:

  (a) the full-size BASELINE configs[1] pair (27.9 k voxels; FCGF net and 6-D inlier net);
  (b) a checkpoint whose folded batch-norm scales span 1e-4 .. 1e4 inside several layers (ONE weight scale per layer
      has to serve all output channels);
  (c) single layers on adversarial rows: magnitudes over 12 decades, small channels inside every row, one channel
      2^20 above the rest, a row holding one value only, zero rows, denormal rows, rows near the f32 limit.

Required: error(split) <= 1.5 x error(exact f32) per saved activation / per layer (with a floor of 2e-6 of the
activation scale, below which both are rounding noise of one f32 ulp per term).  The measured table is printed and, when
DGR_PARITY_REPORT names a directory, appended to <dir>/split_vs_f64.txt (committed as profiles/r03_split_vs_f64.txt).
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'aux'))
import split_f64_inputs as inp   # noqa: E402
from oracle import me_semantics as me, resunet as oresunet   # noqa: E402

pytestmark = pytest.mark.gpu
NAMES = ('s1', 's2', 's4', 's8', 's4_tr', 's2_tr', 's1_tr', 'out')
FLOOR = 2e-6
_lines = []


def report(line):
    print(line)
    _lines.append(line)


@pytest.fixture(scope='module')
def dumps(tmp_path_factory):
    d = tmp_path_factory.mktemp('split_f64')
    out = {}
    for name, env in (('split', {}), ('f32', {'DGR_EXACT_F32': '1'})):
        e = {k: v for k, v in os.environ.items() if k != 'DGR_EXACT_F32'}
        e.update(env)
        path = str(d / f'{name}.npz')
        subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'aux', 'split_f64_dump.py'), path], env=e, check=True,
                       timeout=900)
        out[name] = np.load(path)
    yield out
    rep = os.environ.get('DGR_PARITY_REPORT')
    if rep and _lines:
        os.makedirs(rep, exist_ok=True)
        with open(os.path.join(rep, 'split_vs_f64.txt'), 'w') as f:
            f.write('\n'.join(_lines) + '\n')


def f64_forward(sd, coords, feats, D, ks, norm):
    out, inter = oresunet.resunet_forward(sd, coords, feats, D, ks, norm, return_intermediates=True, dtype=torch.float64)
    inter = dict(inter)
    inter['out'] = out
    return inter


def compare(tag, dumps, prefix, truth):
    worst = 0.0
    for n in NAMES:
        t = np.asarray(truth[n], np.float64)
        scale = max(np.abs(t).max(), 1e-300)
        es = float(np.abs(dumps['split'][prefix + n] - t).max() / scale)
        ef = float(np.abs(dumps['f32'][prefix + n] - t).max() / scale)
        report(f'{tag:34s} {n:6s} max|x - f64| / max|f64|:  split {es:.2e}   exact-f32 {ef:.2e}   ratio {es / max(ef, 1e-300):.2f}')
        assert np.isfinite(dumps['split'][prefix + n]).all()
        assert es <= max(1.5 * ef, FLOOR), (tag, n, es, ef)
        worst = max(worst, es)
    return worst


def test_modes_ran_their_own_kernels(dumps):
    ks, kf = dumps['split']['kinds'].tolist(), dumps['f32']['kinds'].tolist()
    assert any(k.startswith('sparse_conv_wide_f16x2<256, 2, 1>') for k in ks) and not any('f16x2' in k for k in kf)


def test_full_size_pair_against_f64(dumps):
    fs = inp.fullsize_case()
    t3 = f64_forward(fs['sd3'], fs['c0'], np.ones((len(fs['c0']), 1)), 3, 7, True)
    compare('configs[1] FCGF net (27.9 k voxels)', dumps, 'a3_', t3)
    t6 = f64_forward(fs['sd6'], fs['c6'], fs['f6'], 6, 3, False)
    w = compare('configs[1] 6-D inlier net', dumps, 'a6_', t6)
    assert w < 1e-5   # far inside the 1e-4 parity tolerance


def test_batch_norm_scales_over_eight_decades(dumps):
    wr = inp.wide_range_case()
    compare('BN gamma 1e-4..1e4, 6-D net', dumps, 'b6_', f64_forward(wr['sd6'], wr['c6'], wr['f6'], 6, 3, False))
    compare('BN gamma 1e-4..1e4, 3-D net', dumps, 'b3_',
            f64_forward(wr['sd3'], wr['c3'], np.ones((len(wr['c3']), 1)), 3, 5, True))


def layer_truth(sd, name, norm, coords, x, D, relu):
    """f64: out[o] = shift + sum over the 3^D map of max(x[i], 0)? @ (W[k] * scale), and the magnitude bound
    sum |x| @ |W * scale| per output element (the scale an f32 evaluation's rounding error is relative to)."""
    W = np.asarray(sd[name + '.kernel'], np.float64)
    g, b = np.asarray(sd[norm + '.bn.weight'], np.float64), np.asarray(sd[norm + '.bn.bias'], np.float64)
    m, v = np.asarray(sd[norm + '.bn.running_mean'], np.float64), np.asarray(sd[norm + '.bn.running_var'], np.float64)
    s = g / np.sqrt(v + 1e-5)
    Wf = W * s[None, None, :]
    shift = b - m * s
    xx = np.asarray(x, np.float64)
    if relu:
        xx = np.maximum(xx, 0)
    k, i, o = me.kernel_map(coords, coords, D, 3, 1)
    out = np.tile(shift, (len(coords), 1))
    bound = np.tile(np.abs(shift), (len(coords), 1))
    for kk in np.unique(k):
        sel = k == kk
        np.add.at(out, o[sel], xx[i[sel]] @ Wf[kk])
        np.add.at(bound, o[sel], np.abs(xx[i[sel]]) @ np.abs(Wf[kk]))
    return out, bound


@pytest.mark.parametrize('which', ['6', '3'])
def test_single_layers_on_adversarial_rows(dumps, which):
    ad = inp.adversarial_case()
    sd, name, norm = (ad['sd6'], 'block4.conv1', 'block4.norm1') if which == '6' else (ad['sd3'], 'block2.conv1', 'block2.norm1')
    coords, x, D = (ad['c6'], ad['x6'], 6) if which == '6' else (ad['c3'], ad['x3'], 3)
    for relu in (0, 1):
        truth, bound = layer_truth(sd, name, norm, coords, x, D, relu)
        rb = bound.max(axis=1)                      # per output row: the magnitude its terms reach
        ok = (rb < 1e37) & (rb > 1e-30)             # rows whose terms overflow f32 / are all denormal: checked apart
        res = {}
        for mode in ('split', 'f32'):
            y = np.asarray(dumps[mode][f'c{which}_relu{relu}'], np.float64)
            assert np.isfinite(y[ok]).all(), mode
            res[mode] = float((np.abs(y - truth)[ok].max(axis=1) / rb[ok]).max())
            tiny = rb <= 1e-30
            if tiny.any():
                assert np.abs(y - truth)[tiny].max() <= 1e-36, mode   # denormal neighbourhoods: absolutely negligible
        kern = 'sparse_conv_wide_f16x2<256, 2, 1>' if which == '6' else 'sparse_conv_dense_f16x2<64, 64>'
        report(f'{kern:34s} relu={relu} max over rows of max|y - f64| / max sum|x||w|:  split {res["split"]:.2e}   '
               f'exact-f32 {res["f32"]:.2e}   ratio {res["split"] / max(res["f32"], 1e-300):.2f}')
        assert res['split'] <= max(1.5 * res['f32'], 2e-7), (which, relu, res)
