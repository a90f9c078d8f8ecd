"""Both sides of the 2-GB guard of the dense-tile kernel (net.hip, `Fwd::conv`): `sparse_conv_dense_f16x2` addresses its
input through buffer loads with 32-bit byte offsets, so a same-stride C <= 64 layer whose INPUT tensor reaches 2 GB stays
on the list-based kernel (`sparse_conv_os`, 32-bit ELEMENT offsets: 16 GB).  One FCGF forward over 8.7 M voxels puts the
guard on both sides at once: the level-0 64-channel tensors (256 B per row) are 2.2 GB, the level-0 32-channel ones
1.1 GB.

The cloud is 18 x 18 copies of one voxelised 3DMatch-shaped fragment, 256 voxels apart (a multiple of every tensor
stride, far beyond the receptive field of the net): the copies cannot interact, so every copy's features must equal the
features of the fragment alone -- which tests/test_gpu_resunet.py / test_gpu_fullsize.py hold to the oracle -- to the
few f32 ulps by which the two kernels differ (tests/test_gpu_dense_conv.py: 2e-6 of the unit-norm features)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_dense_kernel_guard_both_sides_of_2gb():
    from deepglobalregistration_amd import ops, synth
    a, _, _ = synth.synth_pair(7, n_raw=50000)
    _, c, _ = ops.voxelize(a, 0.05, 0)
    c = c.cpu().numpy()
    n = len(c)
    assert c[:, 1:].max() - c[:, 1:].min() < 200
    G = 18
    shifts = np.array([[0, 256 * i, 256 * j, 0] for i in range(G) for j in range(G)], np.int32)
    big = (c[None, :, :] + shifts[:, None, :]).reshape(-1, 4)
    N = len(big)
    assert N * 256 >= 2 ** 31 > N * 128, N                # the guard separates the 64- and the 32-channel level-0 tensors
    net = ops.NetHandle(synth.synth_state_dict(3, 1, 32, 3, 11), 3, 1, 32, 3, True)
    F_small = net.forward(torch.from_numpy(c).cuda(), torch.ones(n, 1).cuda()).cpu().numpy()
    ops.set_profiling('cuda', True)
    F_big = net.forward(torch.from_numpy(big).cuda(), torch.ones(N, 1).cuda())
    kinds = ops.conv_launch_kinds('cuda')
    ops.set_profiling('cuda', False)
    # The guard looks at the tensor's CAPACITY (the row count of a coarse map is only known on the device, its capacity
    # is the parent map's): block1 (layers 1, 2: 32-channel level-0 input, 1.1 GB) runs dense tiles; block2_tr (19, 20:
    # 64-channel level-0 input, 2.2 GB) and the 64-channel level-1 layers (4, 5, 16, 17: capacity 2.2 GB, 0.75 GB used)
    # the list-based kernel
    for li in (1, 2):
        assert kinds[li].startswith('sparse_conv_dense_f16x2<32, 32'), (li, kinds[li])
    for li in (4, 5, 16, 17, 19, 20):
        assert kinds[li].startswith('sparse_conv_os<64, 64'), (li, kinds[li])
    scale = np.abs(F_small).max()
    for t in (0, 1, G, G * G // 2 + 3, G * G - 1):        # first, neighbours, middle, last copy
        Ft = F_big[t * n:(t + 1) * n].cpu().numpy()
        err = float(np.abs(Ft.astype(np.float64) - F_small).max() / scale)
        print(f'copy {t}: max |F_copy - F_alone| / max |F| = {err:.1e}')
        assert np.isfinite(Ft).all() and err < 2e-6, (t, err)
