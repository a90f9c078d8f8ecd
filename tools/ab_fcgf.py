"""A/B: FCGF forward of one 4-pair batch (8 clouds) with per-layer times; the variant is chosen by the environment
(DGR_OS_LISTS=1: list-based kernel everywhere; DGR_HIP_LIB: another build of the library), one process per variant."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
from deepglobalregistration_amd import ops, synth
sd = synth.synth_state_dict(3, 1, 32, 7, 0)
net = ops.NetHandle(sd, 3, 1, 32, 7, True)
cs = []
for s in range(4):
    a, b, _ = synth.synth_pair(s, n_raw=50000)
    _, ca, _ = ops.voxelize(a, 0.05, 2 * s); _, cb, _ = ops.voxelize(b, 0.05, 2 * s + 1)
    cs += [ca, cb]
C = torch.cat(cs)
if os.environ.get('AB_SORT'):
    # rows in Morton order per cloud (what a spatially sorted internal row order would give)
    c = C.cpu().numpy().astype(np.int64)
    q = c[:, 1:] - c[:, 1:].min(0)
    key = np.zeros(len(c), np.int64)
    for b in range(10):
        for d in range(3):
            key |= ((q[:, d] >> b) & 1) << (3 * b + d)
    key |= c[:, 0] << 40
    C = C[torch.from_numpy(np.argsort(key, kind='stable')).cuda()].contiguous()
if os.environ.get('AB_PARITY'):
    # rows grouped by the parity class of their voxel coordinates (what the transposed convs' row blocks would look like
    # under a parity-sorted row permutation: a class uses 2^(odd dims) of the 27 offsets, all of its rows the same ones)
    c = C.cpu().numpy().astype(np.int64)
    key = (c[:, 1] & 1) | ((c[:, 2] & 1) << 1) | ((c[:, 3] & 1) << 2)
    C = C[torch.from_numpy(np.argsort(key, kind='stable')).cuda()].contiguous()
ones = torch.ones(len(C), 1, device='cuda')
F = net.forward(C, ones)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10): F = net.forward(C, ones)
torch.cuda.synchronize()
tag = os.environ.get('AB_TAG', 'lists' if os.environ.get('DGR_OS_LISTS') else 'default')
print('sorted' if os.environ.get('AB_SORT') else 'parity-sorted' if os.environ.get('AB_PARITY') else 'random-order', 'variant', tag, 'N', len(C), 'fwd ms', (time.time() - t0) * 100)
ops.set_profiling('cuda', True)
F = net.forward(C, ones)
t, g = ops.conv_launch_times('cuda'); kinds = ops.conv_launch_kinds('cuda')
st = ops.stage_times('cuda')
ops.set_profiling('cuda', False)
print('maps_3d', st['maps_3d'], 'conv', st['conv_kernels'])
for i, (a, k) in enumerate(zip(t, kinds)): print(f'  L{i:2d} {a*1e3:8.1f} us  {k}')
if os.environ.get('AB_SAVE'): np.save('gpurun_out/ab_F_%s.npy' % tag, F.cpu().numpy())
