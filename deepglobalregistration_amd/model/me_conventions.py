"""The two MinkowskiEngine conventions a CHECKPOINT depends on, as loader options.

A `kernel` tensor of a checkpoint is [K, Cin, Cout] with K = ks^D kernel offsets in the order MinkowskiEngine 0.5.4
enumerates them (`ME.KernelGenerator(... HYPER_CUBE ...)`, model/residual_block.py:31-36), and a transposed convolution
pairs kernel index and offset the way ME's `MinkowskiConvolutionTranspose` does (`:56-80`).  ME is neither vendored
in the reference nor installable offline, so the library's reading of both -- SURVEY.md appendix A5 / A6: offset index
j = sum_d (delta_d + ks // 2) ks^d, FIRST spatial axis fastest; transposed convolution = the forward strided map with
in / out swapped and the SAME index -- is restated from ME's published behaviour and cannot be checked here (DESIGN.md
section 2, "parity unpinned").  With synthetic weights nothing depends on it; a trained checkpoint read under the
wrong convention produces garbage without any error.

This module makes the reading a property of the LOADER instead of the kernels: `convert_state_dict` re-indexes the
offset axis of the kernels from a stated checkpoint convention to the library's, on the host, before the weights are
tiled for the GPU.  The defaults are the identity.  `tools/check_me_conventions.py` runs a real checkpoint on a real
pair under all four combinations and reports the confidence gate of each: the right one stands out.
"""
import numpy as np

KERNEL_ORDERS = ('first_axis_fastest', 'last_axis_fastest')
DEFAULT = {'kernel_order': 'first_axis_fastest', 'transposed_mirrored': False}


def _is_transposed(name):
    """`conv4_tr / conv3_tr / conv2_tr` (model/resunet.py:521-566) are the MinkowskiConvolutionTranspose layers; `conv1_tr`
    is a plain 1x1 convolution (:568-575) and the `block*_tr` residual blocks hold plain convolutions."""
    parts = name.split('.')
    return len(parts) == 2 and parts[0] in ('conv4_tr', 'conv3_tr', 'conv2_tr')


def kernel_volume_axes(K, D):
    """ks with ks^D == K, or None."""
    ks = int(round(K ** (1.0 / D)))
    for c in (ks - 1, ks, ks + 1):
        if c >= 1 and c ** D == K:
            return c
    return None


def convert_kernel(w, D, kernel_order='first_axis_fastest', mirrored=False):
    """[K, Cin, Cout] in the checkpoint's convention -> the library's.  K = 1 kernels ([Cin, Cout]) pass through."""
    w = np.asarray(w)
    if w.ndim != 3 or w.shape[0] == 1:
        return w
    if kernel_order not in KERNEL_ORDERS:
        raise ValueError(f'kernel_order must be one of {KERNEL_ORDERS}, got {kernel_order!r}')
    K = w.shape[0]
    ks = kernel_volume_axes(K, D)
    if ks is None:
        raise ValueError(f'kernel volume {K} is not ks^{D}')
    if kernel_order == 'last_axis_fastest':
        # checkpoint index = sum_d o_d ks^(D-1-d): as a C-ordered grid its axes are (o_0 .. o_{D-1}); the library's grid
        # is (o_{D-1} .. o_0)
        g = w.reshape((ks,) * D + w.shape[1:])
        w = np.transpose(g, tuple(range(D - 1, -1, -1)) + (D, D + 1)).reshape(w.shape)
    if mirrored:
        w = w[::-1]            # delta -> -delta is index K - 1 - j under either enumeration
    return np.ascontiguousarray(w)


def convert_state_dict(state_dict, D, kernel_order='first_axis_fastest', transposed_mirrored=False):
    """A state dict in the stated checkpoint convention -> the library's (a new dict; tensors untouched when the
    convention is the default)."""
    if kernel_order == DEFAULT['kernel_order'] and not transposed_mirrored:
        return state_dict
    out = {}
    for name, v in state_dict.items():
        if name.endswith('.kernel'):
            v = convert_kernel(v, D, kernel_order, transposed_mirrored and _is_transposed(name))
        out[name] = v
    return out
