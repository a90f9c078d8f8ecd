"""Thin torch-tensor wrappers over the C ABI (one function per entry point of
include/dgr_hip.h).  torch supplies device memory and the current stream only; all
arithmetic happens in libdgr_hip.so."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, get_ctx, ptr, stream_ptr, vp

F32_EPS = float(np.finfo(np.float32).eps)


def _dev(t):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise ValueError('expected a CUDA(ROCm) tensor')
    return t.device


def _as(t, dtype, device):
    if not torch.is_tensor(t):
        t = torch.as_tensor(np.asarray(t))
    return t.to(device=device, dtype=dtype).contiguous()


# ----------------------------------------------------------------------------
def voxelize(xyz, voxel_size, batch_index=0, device='cuda'):
    """ME.utils.sparse_quantize(xyz / voxel, return_index=True) + batched_coordinates
    (core/deep_global_registration.py:152-158).  Returns (xyz_sel f32 [N,3], coords i32 [N,4],
    sel i64 [N]) on the device.  float64 input is quantised in float64 like the reference."""
    lib = _lib.load()
    device = torch.device(device)
    if not torch.is_tensor(xyz):
        xyz = torch.from_numpy(np.ascontiguousarray(xyz))
    if xyz.dtype not in (torch.float32, torch.float64):
        xyz = xyz.double()
    if xyz.dim() != 2 or xyz.shape[1] != 3:
        raise ValueError(f'expected an [M,3] point array, got {tuple(xyz.shape)}')
    xyz = xyz.to(device).contiguous()
    M = xyz.shape[0]
    if M == 0:
        raise ValueError('empty point cloud')
    sel = torch.empty(M, dtype=torch.int64, device=device)
    coords = torch.empty((M, 4), dtype=torch.int32, device=device)
    out = torch.empty((M, 3), dtype=torch.float32, device=device)
    n = C.c_int64(0)
    check(lib.dgr_voxelize(get_ctx(device), ptr(xyz), int(xyz.dtype == torch.float64), M,
                           float(voxel_size), int(batch_index), ptr(sel), ptr(coords), ptr(out),
                           C.byref(n), stream_ptr(device.index)))
    n = n.value
    return out[:n], coords[:n], sel[:n]


# ----------------------------------------------------------------------------
class NetHandle:
    """Owns a dgr_net (weights resident in HBM).  The state dict must already be in the library's kernel-offset
    convention (include/dgr_hip.h at dgr_net_create; model/me_conventions.py converts a checkpoint written under
    another reading -- `ResUNet2.load_state_dict` does that, this class and the C API do not)."""

    def __init__(self, state_dict, D, in_channels, out_channels, conv1_kernel_size,
                 normalize_feature, device='cuda', share_from=None):
        lib = _lib.load()
        self.device = torch.device(device)
        self.D, self.cin, self.cout = D, in_channels, out_channels
        self.conv1_ks, self.normalize = int(conv1_kernel_size), bool(normalize_feature)
        if share_from is not None:
            # a net object for the calling thread's context over the weights `share_from` already holds (dgr_net_share)
            if (share_from.D, share_from.cin, share_from.cout, share_from.conv1_ks, share_from.normalize) != \
                    (D, in_channels, out_channels, self.conv1_ks, self.normalize):
                raise ValueError('share_from is a different network')
            h = vp()
            with torch.cuda.device(self.device):
                check(lib.dgr_net_share(get_ctx(self.device), share_from.handle, C.byref(h)))
            self.handle = h
            return
        keep, descs = [], []
        items = [(n, t) for n, t in state_dict.items() if not n.endswith('num_batches_tracked')]
        # a state dict that is ALREADY on this device (torch CUDA tensors: e.g. views of the broadcast buffer of a
        # multi-GPU start, dist.broadcast_checkpoint) stays there: dgr_net_create_device folds / splits / tiles it with
        # HIP kernels; anything else goes through the host path
        on_device = bool(items) and all(torch.is_tensor(t) and t.is_cuda and t.device.index == (self.device.index or 0)
                                        for _, t in items)
        self.created_on_device = on_device
        for name, t in items:
            if on_device:
                a = t.detach().to(torch.float32).contiguous()
                keep.append(a)
                descs.append(_lib.WeightDesc(name.encode(), a.data_ptr(), a.numel()))
            else:
                a = t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
                a = np.ascontiguousarray(a, dtype=np.float32)
                keep.append(a)
                descs.append(_lib.WeightDesc(name.encode(), a.ctypes.data, a.size))
        arr = (_lib.WeightDesc * len(descs))(*descs)
        h = vp()
        with torch.cuda.device(self.device):
            if on_device:
                torch.cuda.current_stream(self.device).synchronize()   # the tensors' producers (a broadcast) have finished
            create = lib.dgr_net_create_device if on_device else lib.dgr_net_create
            check(create(get_ctx(self.device), D, in_channels, out_channels,
                         conv1_kernel_size, int(bool(normalize_feature)), arr, len(descs),
                         C.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                _lib.load().dgr_net_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    @property
    def param_bytes(self):
        return _lib.load().dgr_net_param_bytes(self.handle)

    @property
    def sharers(self):
        """Net objects (one per context) that hold this net's weight set."""
        return _lib.load().dgr_net_sharers(self.handle)

    def forward(self, coords, feats):
        lib = _lib.load()
        dev = self.device
        coords = _as(coords, torch.int32, dev)
        feats = _as(feats, torch.float32, dev)
        if coords.dim() != 2 or coords.shape[1] != self.D + 1:
            raise ValueError(f'coords must be [N,{self.D + 1}] (batch column first), got {tuple(coords.shape)}')
        if feats.dim() != 2 or feats.shape != (coords.shape[0], self.cin):
            raise ValueError(f'feats must be [{coords.shape[0]},{self.cin}], got {tuple(feats.shape)}')
        N = coords.shape[0]
        if N == 0:
            raise ValueError('empty sparse tensor')
        out = torch.empty((N, self.cout), dtype=torch.float32, device=dev)
        check(lib.dgr_resunet_forward(get_ctx(dev), self.handle, ptr(coords), ptr(feats), N, ptr(out),
                                      stream_ptr(dev.index)))
        return out

    def intermediate(self, name):
        lib = _lib.load()
        rows, cols = C.c_int64(0), C.c_int64(0)
        ctx = get_ctx(self.device)
        check(lib.dgr_net_get_intermediate(ctx, self.handle, name.encode(), None, 0, C.byref(rows), C.byref(cols)))
        buf = np.empty((rows.value, cols.value), np.float32)
        check(lib.dgr_net_get_intermediate(ctx, self.handle, name.encode(), buf.ctypes.data, buf.size,
                                           C.byref(rows), C.byref(cols)))
        return buf

    def debug_conv_layer(self, layer, coords, feats, relu=False):
        """Conv layer `layer` (forward order) applied to `feats` [N, Cin] over the 3^D same-stride map of `coords`,
        through the kernels the forward uses for it (dgr_debug_conv_layer; test instrument)."""
        dev = self.device
        coords = _as(coords, torch.int32, dev)
        feats = _as(feats, torch.float32, dev)
        lay = _lib.load()
        st_cout = None
        # the layer's output width: the channel tables of ResUNetBN2C (model/resunet.py:664-665)
        ch, tr = [None, 32, 64, 128, 256], [None, 64, 64, 64, 128]
        widths = [ch[1]] * 3 + [ch[2]] * 3 + [ch[3]] * 3 + [ch[4]] * 3 + [tr[4]] * 3 + [tr[3]] * 3 + [tr[2]] * 3 + [tr[1], self.cout]
        st_cout = widths[layer]
        out = torch.empty((coords.shape[0], st_cout), dtype=torch.float32, device=dev)
        check(lay.dgr_debug_conv_layer(get_ctx(dev), self.handle, int(layer), ptr(coords), ptr(feats), int(bool(relu)),
                                       coords.shape[0], ptr(out), stream_ptr(dev.index)))
        return out

    def rerun_layer(self, layer, reps=5):
        """(gemm_ms, reduce_ms) of conv layer `layer` of the last forward (kernel-tuning instrument)."""
        lib = _lib.load()
        g, r = C.c_float(0), C.c_float(0)
        check(lib.dgr_net_rerun_layer(get_ctx(self.device), self.handle, layer, reps, C.byref(g), C.byref(r)))
        return g.value, r.value

    def layer_stats(self):
        """Per conv layer of the last forward: dict(pairs, nonempty, n_in, n_out, cin, cout, K)."""
        lib = _lib.load()
        out = []
        for li in range(lib.dgr_net_num_layers(self.handle)):
            st = (C.c_int64 * 8)()
            check(lib.dgr_net_layer_stats(get_ctx(self.device), self.handle, li, st))
            out.append(dict(pairs=st[0], nonempty=st[1], n_in=st[2], n_out=st[3], cin=st[4], cout=st[5], K=st[6]))
        return out


# ----------------------------------------------------------------------------
class Maps:
    """Coordinate maps + kernel maps of one sparse tensor (inspection / parity tests)."""

    def __init__(self, coords, D, conv1_kernel_size=3, device='cuda'):
        lib = _lib.load()
        self.device = torch.device(device)
        coords = _as(coords, torch.int32, self.device)
        self.D = D
        h = vp()
        check(lib.dgr_maps_create(get_ctx(self.device), ptr(coords), coords.shape[0], D, conv1_kernel_size,
                                  C.byref(h), stream_ptr(self.device.index)))
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                _lib.load().dgr_maps_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def coords(self, ts):
        lib = _lib.load()
        n = C.c_int64(0)
        check(lib.dgr_maps_get_coords(self.handle, ts, None, 0, C.byref(n)))
        out = np.empty((n.value, self.D + 1), np.int32)
        check(lib.dgr_maps_get_coords(self.handle, ts, out.ctypes.data, out.size, C.byref(n)))
        return out

    def kernel_map(self, kind, ts):
        """kind: 'same' | 'conv1' | 'down' (rule-major maps) or, D = 3 only, 'nbr_same' | 'nbr_down' | 'nbr_up'
        (the dense neighbour tables of the output-stationary conv; 'nbr_up' = transposed conv 2 ts -> ts).
        Returns (k, in, out) int64 arrays sorted by (k, out)."""
        lib = _lib.load()
        kid = {'same': 0, 'conv1': 1, 'down': 2, 'nbr_same': 3, 'nbr_down': 4, 'nbr_up': 5}[kind]
        K, P = C.c_int64(0), C.c_int64(0)
        check(lib.dgr_maps_get_kernel_map(self.handle, kid, ts, None, 0, None, None, 0, C.byref(K), C.byref(P)))
        rule = np.empty(K.value + 1, np.int32)
        pin = np.empty(P.value, np.int32)
        pout = np.empty(P.value, np.int32)
        check(lib.dgr_maps_get_kernel_map(self.handle, kid, ts, rule.ctypes.data, rule.size, pin.ctypes.data,
                                          pout.ctypes.data, pin.size, C.byref(K), C.byref(P)))
        k = np.repeat(np.arange(K.value), np.diff(rule))
        return k.astype(np.int64), pin.astype(np.int64), pout.astype(np.int64)


# ----------------------------------------------------------------------------
def knn1(F0, F1, squared=False, return_distance=False):
    lib = _lib.load()
    dev = _dev(F0)
    F0 = _as(F0, torch.float32, dev)
    F1 = _as(F1, torch.float32, dev)
    if F0.dim() != 2 or F1.dim() != 2 or F0.shape[1] != F1.shape[1]:
        raise ValueError('F0 [N0,C] and F1 [N1,C] must share the feature width')
    N0, N1 = F0.shape[0], F1.shape[0]
    idx = torch.empty(N0, dtype=torch.int64, device=dev)
    dist = torch.empty(N0, dtype=torch.float32, device=dev) if return_distance else None
    check(lib.dgr_knn1_l2(get_ctx(dev), ptr(F0), N0, ptr(F1), N1, F0.shape[1], int(squared), ptr(idx),
                          ptr(dist), stream_ptr(dev.index)))
    return (idx, dist) if return_distance else idx


def knn1_batch(F0, F1, off0, off1, squared=False, return_distance=False):
    """1-NN of every pair (rows off0[p]:off0[p+1] of F0 against rows off1[p]:off1[p+1] of F1) in one library call;
    indices address rows of the concatenated F1."""
    lib = _lib.load()
    dev = _dev(F0)
    F0 = _as(F0, torch.float32, dev)
    F1 = _as(F1, torch.float32, dev)
    if F0.dim() != 2 or F1.dim() != 2 or F0.shape[1] != F1.shape[1]:
        raise ValueError('F0 [N0,C] and F1 [N1,C] must share the feature width')
    o0 = np.ascontiguousarray(off0, dtype=np.int64)
    o1 = np.ascontiguousarray(off1, dtype=np.int64)
    if len(o0) != len(o1) or len(o0) < 2 or o0[0] != 0 or o1[0] != 0 or o0[-1] != F0.shape[0] or o1[-1] != F1.shape[0]:
        raise ValueError('off0 / off1 must be [npairs+1] row offsets covering F0 / F1')
    idx = torch.empty(F0.shape[0], dtype=torch.int64, device=dev)
    dist = torch.empty(F0.shape[0], dtype=torch.float32, device=dev) if return_distance else None
    check(lib.dgr_knn1_l2_batch(get_ctx(dev), ptr(F0), o0.ctypes.data_as(_lib.c_i64p), ptr(F1),
                                o1.ctypes.data_as(_lib.c_i64p), len(o0) - 1, F0.shape[1], int(squared), ptr(idx),
                                ptr(dist), stream_ptr(dev.index)))
    return (idx, dist) if return_distance else idx


def inlier_inputs(coords0, xyz0, coords1, xyz1, idx1, feature_type='coords'):
    lib = _lib.load()
    dev = _dev(xyz0)
    ft = {'ones': 0, 'coords': 1}.get(feature_type)
    if ft is None:
        raise TypeError('Undefined feature type')
    coords0, coords1 = _as(coords0, torch.int32, dev), _as(coords1, torch.int32, dev)
    xyz0, xyz1 = _as(xyz0, torch.float32, dev), _as(xyz1, torch.float32, dev)
    idx1 = _as(idx1, torch.int64, dev).reshape(-1)
    N0 = coords0.shape[0]
    if idx1.shape[0] != N0:
        raise ValueError('one correspondence per row of fragment 0 expected')
    coords6 = torch.empty((N0, 7), dtype=torch.int32, device=dev)
    feats = torch.empty((N0, 6 if ft == 1 else 1), dtype=torch.float32, device=dev)
    check(lib.dgr_inlier_inputs(get_ctx(dev), ptr(coords0), ptr(xyz0), N0, ptr(coords1), ptr(xyz1),
                                coords1.shape[0], ptr(idx1), ft, ptr(coords6), ptr(feats),
                                stream_ptr(dev.index)))
    return coords6, feats


def sigmoid_clip_sum(logit, clip):
    lib = _lib.load()
    dev = _dev(logit)
    logit = _as(logit, torch.float32, dev).reshape(-1)
    w = torch.empty_like(logit)
    s = C.c_double(0)
    check(lib.dgr_sigmoid_clip_sum(get_ctx(dev), ptr(logit), logit.shape[0], float(clip), ptr(w), C.byref(s),
                                   stream_ptr(dev.index)))
    return w.reshape(-1, 1), s.value


def gather_rows3(src, idx):
    lib = _lib.load()
    dev = _dev(src)
    src = _as(src, torch.float32, dev)
    idx = _as(idx, torch.int64, dev).reshape(-1)
    out = torch.empty((idx.shape[0], 3), dtype=torch.float32, device=dev)
    check(lib.dgr_gather_rows3(get_ctx(dev), ptr(src), ptr(idx), idx.shape[0], ptr(out), stream_ptr(dev.index)))
    return out


def _xyw(X, Y, w):
    dev = _dev(X)
    X, Y = _as(X, torch.float32, dev), _as(Y, torch.float32, dev)
    w = _as(w, torch.float32, dev).reshape(-1)
    if X.shape != Y.shape or X.dim() != 2 or X.shape[1] != 3 or w.shape[0] != X.shape[0]:
        raise ValueError('X, Y must be [N,3] and w [N] / [N,1]')
    return dev, X, Y, w


def weighted_procrustes(X, Y, w, eps=F32_EPS):
    lib = _lib.load()
    dev, X, Y, w = _xyw(X, Y, w)
    R = (C.c_float * 9)()
    t = (C.c_float * 3)()
    check(lib.dgr_weighted_procrustes(get_ctx(dev), ptr(X), ptr(Y), ptr(w), X.shape[0], float(eps), R, t,
                                      stream_ptr(dev.index)))
    return np.array(R, np.float32).reshape(3, 3), np.array(t, np.float32)


def se3_refine(X, Y, w, quantization_size=1.0, max_iter=1000, max_break_count=20,
               break_threshold_ratio=1e-5):
    lib = _lib.load()
    dev, X, Y, w = _xyw(X, Y, w)
    R = (C.c_float * 9)()
    t = (C.c_float * 3)()
    it, bc, loss = C.c_int32(0), C.c_int32(0), C.c_float(0)
    check(lib.dgr_se3_refine(get_ctx(dev), ptr(X), ptr(Y), ptr(w), X.shape[0], float(quantization_size),
                             int(max_iter), int(max_break_count), float(break_threshold_ratio), R, t,
                             C.byref(it), C.byref(loss), C.byref(bc), stream_ptr(dev.index)))
    return (np.array(R, np.float32).reshape(3, 3), np.array(t, np.float32),
            {'iterations': it.value, 'loss': loss.value, 'break_count': bc.value})


def se3_refine_from(X, Y, w, state, max_iter, quantization_size=1.0, max_break_count=10 ** 9,
                    break_threshold_ratio=1e-5):
    """Parity instrumentation (dgr_debug_se3_refine_from): the refinement loop resumed at iteration state['i'] from the
    optimiser state `state` = {'i', 'prm' [9], 'm' [9], 'v' [9], 'loss_prev', 'breaks'} (the layout of
    `oracle.registration.global_registration(states=...)`) and run up to iteration `max_iter`.  Returns the end state."""
    lib = _lib.load()
    dev, X, Y, w = _xyw(X, Y, w)
    si = (C.c_double * 30)(*([float(v) for v in state['prm']] + [float(v) for v in state['m']] + [float(v) for v in state['v']]
                             + [float(state['i']), float(state['loss_prev']), float(state['breaks'])]))
    so = (C.c_double * 30)()
    check(lib.dgr_debug_se3_refine_from(get_ctx(dev), ptr(X), ptr(Y), ptr(w), X.shape[0], float(quantization_size),
                                        int(max_iter), int(max_break_count), float(break_threshold_ratio), si, so,
                                        stream_ptr(dev.index)))
    so = np.array(so, np.float64)
    return {'prm': so[:9].copy(), 'm': so[9:18].copy(), 'v': so[18:27].copy(), 'i': int(so[27]), 'loss_prev': float(so[28]),
            'breaks': int(so[29])}


def _xyz_dev(a, dev=None):
    t = a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))
    if dev is None:
        dev = t.device if t.is_cuda else torch.device('cuda', torch.cuda.current_device())
    t = t.to(device=dev, dtype=torch.float32).contiguous()
    if t.dim() != 2 or t.shape[1] != 3:
        raise ValueError(f'expected [N,3] points, got {tuple(t.shape)}')
    return t


def icp_point_to_point(source, target, max_correspondence_distance, init=None, max_iter=30,
                       relative_fitness=1e-6, relative_rmse=1e-6):
    """Point-to-point ICP (dgr_icp_point_to_point).  Returns (T [4,4] float64, fitness, inlier_rmse,
    iterations)."""
    lib = _lib.load()
    src = _xyz_dev(source)
    dst = _xyz_dev(target, src.device)
    Ti = None
    if init is not None:
        init = np.ascontiguousarray(np.asarray(init, np.float64))
        if init.shape != (4, 4):
            raise ValueError('init must be a 4x4 matrix')
        Ti = (C.c_double * 16)(*init.reshape(-1))
    T = (C.c_double * 16)()
    st = (C.c_double * 3)()
    check(lib.dgr_icp_point_to_point(get_ctx(src.device), ptr(src), src.shape[0], ptr(dst), dst.shape[0],
                                     float(max_correspondence_distance), Ti, int(max_iter), float(relative_fitness),
                                     float(relative_rmse), T, st, stream_ptr(src.device.index)))
    return np.array(T, np.float64).reshape(4, 4), float(st[0]), float(st[1]), int(st[2])


def ransac_correspondence(X, Y, distance_threshold, num_hypotheses, seed=0):
    """RANSAC over corresponding points X[i] <-> Y[i] (dgr_ransac_correspondence).  Returns
    (T [4,4] float64, best hypothesis index, inlier count, inlier rmse)."""
    lib = _lib.load()
    X = _xyz_dev(X)
    Y = _xyz_dev(Y, X.device)
    if X.shape != Y.shape:
        raise ValueError('X and Y must have the same shape')
    T = (C.c_double * 16)()
    st = (C.c_double * 3)()
    check(lib.dgr_ransac_correspondence(get_ctx(X.device), ptr(X), ptr(Y), X.shape[0], float(distance_threshold),
                                        int(num_hypotheses), int(seed) & 0xffffffff, T, st, stream_ptr(X.device.index)))
    return np.array(T, np.float64).reshape(4, 4), int(st[0]), int(st[1]), float(st[2])


# ----------------------------------------------------------------------------
def register_batch(fcgf, inlier, coords0, xyz0, off0, coords1, xyz1, off1, voxel_size,
                   clip_weight_thresh=0.05, inlier_feature_type='coords', max_iter=1000,
                   max_break_count=20, break_threshold_ratio=1e-4, skip_refinement=False,
                   forced_logit=None, override_idx1=None, safeguard=False, use_icp=False, ransac_hypotheses=4000000,
                   ransac_seed=0):
    """Fused pipeline over a batch of voxelised pairs (dgr_register_batch).  Returns
    T [npairs,4,4] float64 (what the reference's register() returns; fetched through dgr_register_batch_f64: the learned
    f32 estimate widens exactly, the safeguard / ICP stages compute in float64), status [npairs] int32, stats [npairs,4] float32.
    `status` is a CODE plus FLAG bits:
    `status & _lib.STATUS_MASK` is 0 ok / 1 low confidence / 2 SVD failed / 3 safeguard (T from the RANSAC), and
    `_lib.STATUS_FLAG_ICP_SKIPPED` (0x100) is OR-ed on when `use_icp` was asked for but the final ICP could not run on the
    pair -- compare the masked code, not the raw word."""
    lib = _lib.load()
    dev = fcgf.device
    npairs = len(off0) - 1
    coords0, coords1 = _as(coords0, torch.int32, dev), _as(coords1, torch.int32, dev)
    xyz0, xyz1 = _as(xyz0, torch.float32, dev), _as(xyz1, torch.float32, dev)
    o0 = (C.c_int64 * (npairs + 1))(*[int(v) for v in off0])
    o1 = (C.c_int64 * (npairs + 1))(*[int(v) for v in off1])
    if coords0.shape[0] != off0[-1] or coords1.shape[0] != off1[-1]:
        raise ValueError('offset arrays do not match the coordinate arrays')
    prm = _lib.Params(float(clip_weight_thresh), float(voxel_size),
                      {'ones': 0, 'coords': 1}[inlier_feature_type], int(max_iter), int(max_break_count),
                      float(break_threshold_ratio), int(bool(skip_refinement)), int(bool(safeguard)),
                      int(ransac_hypotheses), int(ransac_seed) & 0xffffffff, int(bool(use_icp)))
    T = np.empty((npairs, 16), np.float32)   # (float32 at this entry point; float64 through dgr_register_batch_f64 below)
    status = np.empty(npairs, np.int32)
    stats = np.empty((npairs, 4), np.float32)
    fl = None
    if forced_logit is not None:
        fl = _as(forced_logit, torch.float32, dev).reshape(-1)
        if fl.shape[0] != coords0.shape[0]:
            raise ValueError('forced_logit must have one entry per row of fragment 0')
    ov = None
    if override_idx1 is not None:
        ov = _as(override_idx1, torch.int64, dev).reshape(-1)
        if ov.shape[0] != coords0.shape[0]:
            raise ValueError('override_idx1 must have one entry per row of fragment 0')
    check(lib.dgr_register_batch(get_ctx(dev), fcgf.handle, inlier.handle, ptr(coords0), ptr(xyz0), o0,
                                 ptr(coords1), ptr(xyz1), o1, npairs, C.byref(prm), ptr(ov), ptr(fl),
                                 T.ctypes.data_as(_lib.c_f32p), status.ctypes.data_as(_lib.c_i32p),
                                 stats.ctypes.data_as(_lib.c_f32p), stream_ptr(dev.index)))
    # always at full width, whatever the flags (the RANSAC / ICP stages compute in float64 like Open3D; without them the
    # f32 estimate widens exactly): one return dtype.  The library clears its float64 copy when a call starts, so a call
    # that failed midway cannot leave an earlier batch's transforms to be read here.
    T64 = np.empty((npairs, 16), np.float64)
    n = C.c_int64(0)
    check(lib.dgr_register_batch_f64(get_ctx(dev), T64.ctypes.data_as(_lib.c_f64p), npairs, C.byref(n)))
    if n.value != npairs:
        raise RuntimeError(f'dgr_register_batch_f64 holds {n.value} pairs, expected {npairs}')
    return T64.reshape(npairs, 4, 4), status, stats


_BATCH_OUT = {'idx1': (0, torch.int64), 'logit': (1, torch.float32), 'weights': (2, torch.float32),
              'F0': (3, torch.float32), 'F1': (4, torch.float32)}


def batch_output(device, which):
    """Copy of a device-side intermediate of the last register_batch ('idx1', 'logit', 'weights',
    'F0', 'F1') as a flat torch tensor."""
    lib = _lib.load()
    dev = torch.device(device)
    wid, dtype = _BATCH_OUT[which]
    n = C.c_int64(0)
    check(lib.dgr_register_batch_output(get_ctx(dev), wid, None, 0, C.byref(n), stream_ptr(dev.index)))
    out = torch.empty(n.value, dtype=dtype, device=dev)
    check(lib.dgr_register_batch_output(get_ctx(dev), wid, ptr(out), out.numel() * out.element_size(),
                                        C.byref(n), stream_ptr(dev.index)))
    return out


def partition_stream(device, part, nparts):
    """A stream of the calling thread's context on share `part` of `nparts` (2 or 4) equal shares of the GPU's compute
    units (dgr_ctx_create_partition_stream, include/dgr_hip.h), as a torch stream: make it the current stream
    (`with torch.cuda.stream(s):`) for every call of this context.  `nparts = 1` drops the partition and returns None.
    Pays only when several contexts (one per host thread) keep the GPU busy at once; results do not depend on it."""
    device = torch.device(device)
    h = vp()
    check(_lib.load().dgr_ctx_create_partition_stream(get_ctx(device), int(part), int(nparts), C.byref(h)))
    return torch.cuda.ExternalStream(h.value, device) if h.value else None


def set_profiling(device, enable):
    check(_lib.load().dgr_ctx_set_profiling(get_ctx(device), int(bool(enable))))


def stage_times(device):
    t = (C.c_float * 16)()
    n = C.c_int(0)
    check(_lib.load().dgr_ctx_stage_times_v2(get_ctx(device), t, 16, C.byref(n)))
    names = ['fcgf', 'knn', 'inlier_inputs', 'inlier_net', 'registration', 'maps_3d', 'maps_6d', 'conv_kernels', 'o3d_steps']
    out = dict(zip(names, [float(v) for v in t[:n.value]]))
    out['conv_launches'] = int(_lib.load().dgr_ctx_conv_launches(get_ctx(device)))
    return out


def conv_launch_times(device):
    """Per-launch durations (ms) of the sparse-conv layers of the last profiled batch in launch order:
    (MFMA phase + reduce phase, MFMA phase alone)."""
    cap = 4096
    t, g = (C.c_float * cap)(), (C.c_float * cap)()
    n = C.c_int64(0)
    check(_lib.load().dgr_ctx_conv_launch_times(get_ctx(device), t, g, cap, C.byref(n)))
    return [float(t[i]) for i in range(n.value)], [float(g[i]) for i in range(n.value)]


def conv_launch_kernel_us(device):
    """Per launch of `conv_launch_times`: the kernel's own execution span in us (stamped by the kernel on the device wall
    clock: the duration rocprofv3 --kernel-trace reports, valid under concurrent streams); 0 = kernel not instrumented."""
    cap = 4096
    t = (C.c_float * cap)()
    n = C.c_int64(0)
    check(_lib.load().dgr_ctx_conv_launch_kernel_us(get_ctx(device), t, cap, C.byref(n)))
    return [float(t[i]) for i in range(n.value)]


def conv_launch_kinds(device):
    """Kernel variant (rocprof kernel name) of every launch reported by `conv_launch_times`."""
    cap = 1 << 16
    buf = C.create_string_buffer(cap)
    n = C.c_int64(0)
    check(_lib.load().dgr_ctx_conv_launch_kinds(get_ctx(device), buf, cap, C.byref(n)))
    names = buf.value.decode().split('\n')
    return names[:n.value]


def debug_ortho2rotation(p6, grad_R=None):
    """ortho2rotation forward (and backward for `grad_R` [n,3,3]) exactly as the registration kernel computes it."""
    dev = _dev(p6)
    p6 = _as(p6, torch.float32, dev).reshape(-1, 6)
    n = p6.shape[0]
    R = torch.empty((n, 3, 3), dtype=torch.float32, device=dev)
    g = dp = None
    if grad_R is not None:
        g = _as(grad_R, torch.float32, dev).reshape(n, 9)
        dp = torch.empty((n, 6), dtype=torch.float32, device=dev)
    check(_lib.load().dgr_debug_ortho2rotation(get_ctx(dev), ptr(p6), n, ptr(g), ptr(R), ptr(dp), stream_ptr(dev.index)))
    return (R, dp) if grad_R is not None else R


def debug_smooth_l1(X, Y, quantization_size):
    """HighDimSmoothL1Loss per point, the device function of the registration kernel."""
    dev = _dev(X)
    X, Y = _as(X, torch.float32, dev), _as(Y, torch.float32, dev)
    out = torch.empty(X.shape[0], dtype=torch.float32, device=dev)
    check(_lib.load().dgr_debug_smooth_l1(get_ctx(dev), ptr(X), ptr(Y), X.shape[0], float(quantization_size), ptr(out),
                                          stream_ptr(dev.index)))
    return out
