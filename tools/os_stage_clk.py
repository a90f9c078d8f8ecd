"""Where the list-based output-stationary kernel (conv_os.hip) spends a workgroup's life: a build with -DDGR_OS_STAGE_CLK
(make EXTRA=-DDGR_OS_STAGE_CLK, library given by DGR_HIP_LIB) sums per-workgroup wall-clock spans of its stages; this
re-runs every such layer of one FCGF forward (8 clouds, the A/B workload of tools/ab_fcgf.py) and prints the averages."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, '.')
from deepglobalregistration_amd import ops, synth, _lib
sd = synth.synth_state_dict(3, 1, 32, 7, 0)
net = ops.NetHandle(sd, 3, 1, 32, 7, True)
cs = []
for s in range(4):
    a, b, _ = synth.synth_pair(s, n_raw=50000)
    _, ca, _ = ops.voxelize(a, 0.05, 2 * s); _, cb, _ = ops.voxelize(b, 0.05, 2 * s + 1)
    cs += [ca, cb]
Cc = torch.cat(cs)
if os.environ.get('AB_PARITY'):
    c = Cc.cpu().numpy().astype(np.int64)
    key = (c[:, 1] & 1) | ((c[:, 2] & 1) << 1) | ((c[:, 3] & 1) << 2)
    Cc = Cc[torch.from_numpy(np.argsort(key, kind='stable')).cuda()].contiguous()
ones = torch.ones(len(Cc), 1, device='cuda')
ops.set_profiling('cuda', True)
net.forward(Cc, ones)
kinds = ops.conv_launch_kinds('cuda')
ops.set_profiling('cuda', False)
lib = _lib.load()
buf = (C.c_ulonglong * 8)()
REPS = 5
print('layer kernel | us per launch | per workgroup (us): compaction, init+groups, phases, epilogue | workgroups, phases per workgroup, us per phase')
for li, k in enumerate(kinds):
    if not k.startswith('sparse_conv_os'): continue
    lib.dgr_debug_os_stage_clk(None, 1)
    g, r = net.rerun_layer(li, REPS)
    torch.cuda.synchronize()
    lib.dgr_debug_os_stage_clk(buf, 0)
    v = [int(x) for x in buf]
    wg = max(v[4], 1)
    st = [v[i] / wg * 0.01 for i in range(4)]
    print(f'L{li:2d} {k:44s} | {g * 1e3:7.1f} | {st[0]:6.2f} {st[1]:6.2f} {st[2]:7.2f} {st[3]:6.2f} | {wg // REPS:6d} {v[5] / wg:6.1f} {st[2] / max(v[5] / wg, 1e-9):6.2f}')
