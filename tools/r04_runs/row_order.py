"""Does the ROW ORDER of the level-0 tensor matter to the 3-D (FCGF) net?  One batch of 8 BASELINE configs[1] clouds
(4 pairs), the same voxels in four row orders; whole-forward time and the per-layer times of the library's profiling
counters.  (The wide rule-major kernels of the 6-D net do not care -- profiles/r04_wide_check_sorted_vs_shuffled.txt --
but the dense-tile kernel gathers 27 neighbours per output row, and the rows of a wave tile share neighbours only if
they are neighbours themselves.)  Output rows are compared with the first order's after un-permuting."""
import json
import os
import sys

import numpy as np
import torch


def morton3(c):
    c = (c - c.min(0)).astype(np.uint64)
    key = np.zeros(len(c), np.uint64)
    for b in range(16):
        for d in range(3):
            key |= ((c[:, d] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + d)
    return key


def orders(coords, rng):
    n = len(coords)
    b = coords[:, 0].astype(np.uint64)
    xyz = coords[:, 1:].astype(np.int64)
    out = {'first_occurrence': np.arange(n)}
    out['shuffled'] = np.lexsort((rng.random(n), b))
    out['morton'] = np.lexsort((morton3(xyz), b))
    blk = morton3(xyz >> 3)
    out['block8_then_first_occurrence'] = np.lexsort((np.arange(n), blk, b))
    return out


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepglobalregistration_amd import ops, synth
    rng = np.random.default_rng(0)
    cl = []
    for p in range(4):
        a, b, _ = synth.synth_pair(100 + p, n_raw=50000)
        cl += [a, b]
    parts = [ops.voxelize(x, 0.05, i)[1].cpu().numpy() for i, x in enumerate(cl)]
    coords = np.concatenate(parts)
    print('rows', len(coords), [len(p) for p in parts], flush=True)
    net = ops.NetHandle(synth.synth_state_dict(3, 1, 32, 7, 0), 3, 1, 32, 7, True)
    res, base = {}, None
    only = sys.argv[2].split(',') if len(sys.argv) > 2 else None     # (optional: a subset of the orders, for A/B runs of library builds)
    for name, perm in orders(coords, rng).items():
        if only and name not in only:
            continue
        c = torch.from_numpy(np.ascontiguousarray(coords[perm])).cuda()
        f = torch.ones(len(c), 1, device='cuda')
        for _ in range(3):
            F = net.forward(c, f)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            F = net.forward(c, f)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        ops.set_profiling('cuda', True)
        net.forward(c, f)
        torch.cuda.synchronize()
        t, g = ops.conv_launch_times('cuda')
        kinds = ops.conv_launch_kinds('cuda')
        st = ops.stage_times('cuda')
        ops.set_profiling('cuda', False)
        Fh = np.empty((len(c), 32), np.float32)
        Fh[perm] = F.cpu().numpy()
        if base is None:
            base = Fh
        dev = float(np.abs(Fh - base).max())
        per_kind = {}
        for k, v in zip(kinds, t):
            per_kind[k] = per_kind.get(k, 0.0) + v
        res[name] = {'forward_ms': ms, 'conv_ms': sum(t), 'maps_3d_ms': st.get('maps_3d'), 'max_dev_vs_first': dev,
                     'per_kernel_ms': {k: round(v, 4) for k, v in sorted(per_kind.items(), key=lambda kv: -kv[1])},
                     'per_layer_ms': [round(v, 4) for v in t]}
        print(name, json.dumps({k: v for k, v in res[name].items() if k != 'per_layer_ms'}), flush=True)
    json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else 'row_order.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
