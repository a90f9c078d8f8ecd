"""ctypes binding of libdgr_hip.so (the C ABI declared in include/dgr_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a call
fails, a RuntimeError is raised.  torch is imported first so that the process
uses ONE HIP runtime (torch's bundled libamdhip64.so.7 satisfies the library's
NEEDED entry by SONAME).
"""
import ctypes as C
import os
import threading

import torch  # noqa: F401  (must be loaded before libdgr_hip.so, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DGR_HIP_LIB') or os.path.join(_HERE, 'lib', 'libdgr_hip.so')  # override: experiments only

DGR_OK, DGR_EINVAL, DGR_EHIP, DGR_ENOMEM, DGR_ESVD, DGR_EINTERNAL = 0, -1, -2, -3, -4, -5
STATUS_OK, STATUS_LOW_CONFIDENCE, STATUS_SVD_FAILED, STATUS_SAFEGUARD = 0, 1, 2, 3
STATUS_FLAG_ICP_SKIPPED, STATUS_MASK = 0x100, 0xff   # flag OR-ed onto a code: the final ICP could not run on the pair

c_i32p, c_i64p, c_f32p, c_f64p = (C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                  C.POINTER(C.c_float), C.POINTER(C.c_double))
vp = C.c_void_p


class WeightDesc(C.Structure):
    _fields_ = [('name', C.c_char_p), ('data', vp), ('numel', C.c_int64)]


class Params(C.Structure):
    _fields_ = [('clip_weight_thresh', C.c_float), ('voxel_size', C.c_float),
                ('inlier_feature_type', C.c_int), ('max_iter', C.c_int),
                ('max_break_count', C.c_int), ('break_threshold_ratio', C.c_double),
                ('skip_refinement', C.c_int), ('safeguard', C.c_int), ('ransac_hypotheses', C.c_int64),
                ('ransac_seed', C.c_uint32), ('use_icp', C.c_int)]


# name -> (restype, argtypes); every symbol declared in include/dgr_hip.h
SIGNATURES = {
    'dgr_last_error': (C.c_char_p, []),
    'dgr_version': (C.c_char_p, []),
    'dgr_ctx_create': (C.c_int, [C.c_int, C.POINTER(vp)]),
    'dgr_ctx_destroy': (None, [vp]),
    'dgr_ctx_workspace_bytes': (C.c_int64, [vp]),
    'dgr_ctx_create_partition_stream': (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(vp)]),
    'dgr_voxelize': (C.c_int, [vp, vp, C.c_int, C.c_int64, C.c_double, C.c_int32, vp, vp, vp, c_i64p, vp]),
    'dgr_net_create': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(WeightDesc), C.c_int, C.POINTER(vp)]),
    'dgr_net_create_device': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(WeightDesc), C.c_int, C.POINTER(vp)]),
    'dgr_net_destroy': (None, [vp]),
    'dgr_net_param_bytes': (C.c_int64, [vp]),
    'dgr_net_share': (C.c_int, [vp, vp, C.POINTER(vp)]),
    'dgr_net_sharers': (C.c_int, [vp]),
    'dgr_resunet_forward': (C.c_int, [vp, vp, vp, vp, C.c_int64, vp, vp]),
    'dgr_net_get_intermediate': (C.c_int, [vp, vp, C.c_char_p, vp, C.c_int64, c_i64p, c_i64p]),
    'dgr_net_layer_stats': (C.c_int, [vp, vp, C.c_int, c_i64p]),
    'dgr_net_num_layers': (C.c_int, [vp]),
    'dgr_net_rerun_layer': (C.c_int, [vp, vp, C.c_int, C.c_int, c_f32p, c_f32p]),
    'dgr_maps_create': (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, C.POINTER(vp), vp]),
    'dgr_maps_destroy': (None, [vp]),
    'dgr_maps_get_coords': (C.c_int, [vp, C.c_int, vp, C.c_int64, c_i64p]),
    'dgr_maps_get_kernel_map': (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int64, vp, vp, C.c_int64,
                                          c_i64p, c_i64p]),
    'dgr_knn1_l2': (C.c_int, [vp, vp, C.c_int64, vp, C.c_int64, C.c_int, C.c_int, vp, vp, vp]),
    'dgr_knn1_l2_batch': (C.c_int, [vp, vp, c_i64p, vp, c_i64p, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    'dgr_inlier_inputs': (C.c_int, [vp, vp, vp, C.c_int64, vp, vp, C.c_int64, vp, C.c_int, vp, vp, vp]),
    'dgr_sigmoid_clip_sum': (C.c_int, [vp, vp, C.c_int64, C.c_float, vp, c_f64p, vp]),
    'dgr_gather_rows3': (C.c_int, [vp, vp, vp, C.c_int64, vp, vp]),
    'dgr_weighted_procrustes': (C.c_int, [vp, vp, vp, vp, C.c_int64, C.c_float, c_f32p, c_f32p, vp]),
    'dgr_se3_refine': (C.c_int, [vp, vp, vp, vp, C.c_int64, C.c_float, C.c_int, C.c_int, C.c_double,
                                 c_f32p, c_f32p, c_i32p, c_f32p, c_i32p, vp]),
    'dgr_register_batch': (C.c_int, [vp, vp, vp, vp, vp, c_i64p, vp, vp, c_i64p, C.c_int,
                                     C.POINTER(Params), vp, vp, c_f32p, c_i32p, c_f32p, vp]),
    'dgr_register_batch_f64': (C.c_int, [vp, c_f64p, C.c_int64, c_i64p]),
    'dgr_register_batch_output': (C.c_int, [vp, C.c_int, vp, C.c_int64, c_i64p, vp]),
    'dgr_ctx_set_profiling': (C.c_int, [vp, C.c_int]),
    'dgr_icp_point_to_point': (C.c_int, [vp, vp, C.c_int64, vp, C.c_int64, C.c_double, c_f64p, C.c_int, C.c_double,
                                          C.c_double, c_f64p, c_f64p, vp]),
    'dgr_ransac_correspondence': (C.c_int, [vp, vp, vp, C.c_int64, C.c_double, C.c_int64, C.c_uint32, c_f64p, c_f64p, vp]),
    'dgr_ctx_stage_times': (C.c_int, [vp, c_f32p]),
    'dgr_ctx_stage_times_v2': (C.c_int, [vp, c_f32p, C.c_int, C.POINTER(C.c_int)]),
    'dgr_ctx_conv_launches': (C.c_int64, [vp]),
    'dgr_ctx_conv_launch_times': (C.c_int, [vp, c_f32p, c_f32p, C.c_int64, C.POINTER(C.c_int64)]),
    'dgr_ctx_conv_launch_kernel_us': (C.c_int, [vp, c_f32p, C.c_int64, C.POINTER(C.c_int64)]),
    'dgr_ctx_conv_launch_kinds': (C.c_int, [vp, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]),
    'dgr_debug_ortho2rotation': (C.c_int, [vp, vp, C.c_int64, vp, vp, vp, vp]),
    'dgr_debug_se3_refine_from': (C.c_int, [vp, vp, vp, vp, C.c_int64, C.c_float, C.c_int, C.c_int, C.c_double,
                                            c_f64p, c_f64p, vp]),
    'dgr_debug_smooth_l1': (C.c_int, [vp, vp, vp, C.c_int64, C.c_float, vp, vp]),
    'dgr_debug_conv_layer': (C.c_int, [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int64, vp, vp]),
}

_lib = None


def load():
    """Load libdgr_hip.so and declare every prototype.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} not found: the HIP extension is not built.  Run '
            '`python -c "import __graft_entry__ as g; g.build()"` (or `make -C '
            'deepglobalregistration_amd/csrc`).  There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class DgrError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f'libdgr_hip error {code}: {message}')
        self.code = code


def check(rc):
    """0 -> ok; DGR_EINVAL -> ValueError; everything else -> RuntimeError (so that the
    `except RuntimeError` around the SVD at core/deep_global_registration.py:295 keeps working)."""
    if rc == DGR_OK:
        return
    msg = load().dgr_last_error().decode('utf-8', 'replace')
    if rc == DGR_EINVAL:
        raise ValueError(f'libdgr_hip: {msg}')
    raise DgrError(rc, msg)


# ----------------------------------------------------------------------------
# one context per device
# ----------------------------------------------------------------------------
_ctxs = {}
_tls = threading.local()   # optional per-thread context (one context per stream / host thread)


def new_ctx(device):
    """A fresh library context on `device` (own workspace).  A context is not thread-safe: a host
    thread that drives its own HIP stream installs its own with `use_ctx`."""
    lib = load()
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    h = vp()
    check(lib.dgr_ctx_create(idx, C.byref(h)))
    return h


def use_ctx(handle):
    """Install `handle` as the calling thread's context (None = back to the per-device default)."""
    _tls.ctx = handle


def get_ctx(device=None):
    lib = load()
    if getattr(_tls, 'ctx', None) is not None:
        return _tls.ctx
    if not torch.cuda.is_available():
        raise RuntimeError('deepglobalregistration_amd needs a ROCm GPU (torch.cuda.is_available() is '
                           'False); there is no CPU fallback')
    if device is None:
        idx = torch.cuda.current_device()
    else:
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError(f'deepglobalregistration_amd runs on the GPU only, got device {device}')
        idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _ctxs:
        h = vp()
        check(lib.dgr_ctx_create(idx, C.byref(h)))
        _ctxs[idx] = h
    return _ctxs[idx]


def stream_ptr(device_index=None):
    return vp(torch.cuda.current_stream(device_index).cuda_stream)


def ptr(t):
    """Device/host pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return vp(0)
    assert t.is_contiguous(), 'tensor must be contiguous'
    return vp(t.data_ptr())
