"""A context's own share of the compute units (dgr_ctx_create_partition_stream, include/dgr_hip.h): the fused batched
path on a CU-masked stream returns the SAME BITS as on a plain stream (the partition changes grids, never results), two
contexts on two shares run side by side from two host threads, bad arguments are refused."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
VOXEL = 0.05


def _batch(dgr, synth, seeds):
    x0, c0, x1, c1, off0, off1, ovr = [], [], [], [], [0], [0], []
    for p, s in enumerate(seeds):
        a, b, Tg = synth.synth_pair(s, n_raw=6000)
        xa, ca, _ = dgr.preprocess(a, batch_index=p)
        xb, cb, _ = dgr.preprocess(b, batch_index=p)
        g = synth.gt_correspondences(xa.cpu().numpy(), xb.cpu().numpy(), Tg, VOXEL, seed=s)
        ovr.append(np.where(g >= 0, g + off1[-1], -1))
        x0.append(xa); c0.append(ca); x1.append(xb); c1.append(cb)
        off0.append(off0[-1] + len(xa)); off1.append(off1[-1] + len(xb))
    return (torch.cat(c0), torch.cat(x0), off0, torch.cat(c1), torch.cat(x1), off1), torch.from_numpy(np.concatenate(ovr)).cuda()


def _run(dgr, ops, bt, ovr):
    T, status, stats = dgr.register_voxelized(*bt, override_idx1=ovr)
    torch.cuda.synchronize()
    return T, status, stats, ops.batch_output('cuda', 'logit').cpu().numpy(), ops.batch_output('cuda', 'F0').cpu().numpy()


def test_partition_stream_changes_no_bit_and_two_shares_run_side_by_side():
    from deepglobalregistration_amd import _lib, ops, synth
    from deepglobalregistration_amd.core.deep_global_registration import DeepGlobalRegistration
    dev = torch.device('cuda')
    ck = synth.synth_checkpoint(seed=0, voxel_size=VOXEL, feat_conv1_kernel_size=7)
    ctxs = [_lib.new_ctx(dev) for _ in range(2)]
    dgrs, refs, bts = [], [], []
    for i, ctx in enumerate(ctxs):
        _lib.use_ctx(ctx)
        cfg = {'weights': ck, 'clip_weight_thresh': 0.05}
        if dgrs:
            cfg['share_weights_with'] = dgrs[0]
        dgrs.append(DeepGlobalRegistration(cfg, dev))
        bts.append(_batch(dgrs[i], synth, (10 + 2 * i, 11 + 2 * i)))
        refs.append(_run(dgrs[i], ops, *bts[i]))            # plain stream, all compute units
    _lib.use_ctx(None)
    # (a) one context on a quarter of the GPU: the same bits
    _lib.use_ctx(ctxs[0])
    s = ops.partition_stream(dev, 1, 4)
    assert s is not None
    with torch.cuda.stream(s):
        got = _run(dgrs[0], ops, *bts[0])
    for a, b in zip(got, refs[0]):
        np.testing.assert_array_equal(a, b)
    assert ops.partition_stream(dev, 0, 1) is None           # dropped again: plain streams, all compute units
    for a, b in zip(_run(dgrs[0], ops, *bts[0]), refs[0]):
        np.testing.assert_array_equal(a, b)
    _lib.use_ctx(None)
    # (b) two contexts, two host threads, the two halves of the GPU at once, three batches each
    out = [None, None]

    def work(i):
        _lib.use_ctx(ctxs[i])
        st = ops.partition_stream(dev, i, 2)
        with torch.cuda.stream(st):
            for _ in range(3):
                out[i] = _run(dgrs[i], ops, *bts[i])
        ops.partition_stream(dev, 0, 1)
        _lib.use_ctx(None)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(2):
        assert out[i] is not None
        for a, b in zip(out[i], refs[i]):
            np.testing.assert_array_equal(a, b)


def test_partition_stream_refuses_bad_arguments():
    from deepglobalregistration_amd import ops
    dev = torch.device('cuda')
    for part, nparts in ((0, 3), (4, 4), (-1, 2), (0, 8), (0, 0)):
        with pytest.raises(ValueError):
            ops.partition_stream(dev, part, nparts)
    assert ops.partition_stream(dev, 0, 1) is None
