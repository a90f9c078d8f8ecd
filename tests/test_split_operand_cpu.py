"""The split-operand arithmetic of the conv kernels (conv_wide.hip, conv_os.hip, conv_dense.hip), restated in numpy on the CPU.

An f32 operand x under a power-of-two scale s is represented as s x = h + m + d with h = rn16(s x),
m = rn16(s x - h); the kernels keep the products w_h x_h + w_h x_m + w_m x_h.  This test states the two claims
DESIGN.md 4.2 makes about it, without a GPU: (1) the representation error per operand is <= 2^-22 relative (and
zero for most operands), rows of any magnitude included; (2) a dot product formed from the three kept products is
closer to the exact result than an f32 FMA chain over the unsplit operands.  The GPU-side measurement of the real
kernels is tests/test_gpu_split_f64.py (profiles/r03_split_vs_f64.txt)."""
import numpy as np


def row_scale(x):
    """dgr_row_scale: the power of two that moves a row's largest |x| into [2^14, 2^15); 1 for rows of zeros."""
    mx = np.abs(x).max(axis=-1, keepdims=True).astype(np.float32)
    e = np.clip((mx.view(np.uint32) >> 23) & 0xff, 20, 240).astype(np.uint32)
    s = ((268 - e) << 23).astype(np.uint32).view(np.float32)
    return np.where(mx == 0, np.float32(1), s)


def split2(x, s):
    xs = (x * s).astype(np.float32)
    h = xs.astype(np.float16)
    m = (xs - h.astype(np.float32)).astype(np.float32).astype(np.float16)
    return h, m


def test_two_pieces_hold_22_bits_at_any_row_magnitude():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((64, 256)) * 10.0 ** rng.integers(-30, 30, (64, 1))).astype(np.float32)
    x[3] = 0
    x[:, ::7] *= np.float32(1e-4)               # channels far below their row's maximum
    s = row_scale(x)
    assert np.all((np.abs(x * s).max(1)[np.abs(x).max(1) > 0] >= 2.0 ** 14) & (np.abs(x * s).max(1)[np.abs(x).max(1) > 0] < 2.0 ** 15))
    h, m = split2(x, s)
    assert np.isfinite(h.astype(np.float32)).all()
    rec = (h.astype(np.float64) + m.astype(np.float64)) / s.astype(np.float64)
    err = np.abs(rec - x.astype(np.float64))
    ok = np.abs(x) > 0
    big = ok & (np.abs(x) >= np.abs(x).max(1, keepdims=True) * 2.0 ** -17)   # both pieces normal: the full 22 bits
    assert (err[big] <= 2.0 ** -22 * np.abs(x[big])).all()
    assert (err[ok] <= 2.0 ** -38 * np.abs(x).max(1, keepdims=True).repeat(256, 1)[ok] + 2.0 ** -22 * np.abs(x[ok])).all()
    assert (err[big] == 0).mean() > 0.2                                      # many operands are exact


def test_three_products_beat_an_f32_fma_chain():
    rng = np.random.default_rng(1)
    for n in (256, 256 * 27):
        x = np.maximum(rng.standard_normal((32, n)), 0).astype(np.float32) * rng.lognormal(0, 2, (32, 1)).astype(np.float32)
        w = (rng.standard_normal((n, 16)) * 0.05).astype(np.float32)
        ref = x.astype(np.float64) @ w.astype(np.float64)
        sx = row_scale(x)
        sw = row_scale(w.reshape(1, -1))[0, 0]
        xh, xm = (a.astype(np.float64) for a in split2(x, sx))
        wh, wm = (a.astype(np.float64) for a in split2(w, sw))
        # f32 accumulation in blocks of 16 channels like the MFMA (small terms first)
        acc = np.zeros((32, 16), np.float32)
        for k0 in range(0, n, 16):
            sl = slice(k0, k0 + 16)
            for a, b in ((xh, wm), (xm, wh), (xh, wh)):
                acc = (acc + a[:, sl] @ b[sl]).astype(np.float32)
        got = acc.astype(np.float64) / sx / sw
        chain = np.zeros((32, 16), np.float32)
        for k in range(n):                                                   # the yardstick: one fused multiply-add per term
            chain = (chain + x[:, k:k + 1].astype(np.float64) * w[k:k + 1].astype(np.float64)).astype(np.float32)
        scale = np.abs(ref).max(1, keepdims=True)
        e_split, e_chain = (np.abs(got - ref) / scale).max(), (np.abs(chain - ref) / scale).max()
        assert e_split < 2e-6 and e_split <= e_chain * 1.5, (n, e_split, e_chain)
