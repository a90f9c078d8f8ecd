"""`ResUNetBN2C` with the reference's constructor / load_state_dict / eval / call surface
(model/resunet.py:419-665), executed by libdgr_hip.so.  Only the inference path exists:
weights live in HBM in MFMA-tiled form, `forward` is one C-ABI call."""
import numpy as np
import torch

from .. import ops
from ..sparse import SparseTensor
from . import me_conventions


class ResUNet2:
    NORM_TYPE = None
    BLOCK_NORM_TYPE = 'BN'
    CHANNELS = [None, 32, 64, 128, 256]
    TR_CHANNELS = [None, 32, 64, 64, 128]

    def __init__(self, in_channels=3, out_channels=32, bn_momentum=0.1, conv1_kernel_size=3,
                 normalize_feature=False, D=3):
        if (self.NORM_TYPE, self.CHANNELS, self.TR_CHANNELS) != ('BN', [None, 32, 64, 128, 256],
                                                                   [None, 64, 64, 64, 128]):
            raise NotImplementedError('only the ResUNetBN2C configuration is on the DGR inference path')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.bn_momentum = bn_momentum            # training-only; kept for signature parity
        self.conv1_kernel_size = conv1_kernel_size
        self.normalize_feature = normalize_feature
        self.D = D
        self.device = torch.device('cuda')
        self._state = None
        self._net = None
        self.training = True
        # how the CHECKPOINT enumerates kernel offsets / pairs them in transposed convs (me_conventions.py; the default
        # is the library's own reading of MinkowskiEngine 0.5.4); set before load_state_dict
        self.me_conventions = dict(me_conventions.DEFAULT)

    # --- torch.nn.Module-like surface used by core/deep_global_registration.py:114-131 -------
    def load_state_dict(self, state_dict, strict=True):
        default_conv = (self.me_conventions['kernel_order'] == me_conventions.DEFAULT['kernel_order']
                        and not self.me_conventions['transposed_mirrored'])
        if default_conv and state_dict and all(torch.is_tensor(v) and v.is_cuda for v in state_dict.values()):
            # already in HBM and in the library's convention (e.g. out of dist.broadcast_checkpoint): it stays on the device,
            # the library prepares its operand layouts there (dgr_net_create_device)
            self._state = dict(state_dict)
        else:
            self._state = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                           for k, v in state_dict.items()}
            self._state = me_conventions.convert_state_dict(self._state, self.D, self.me_conventions['kernel_order'],
                                                            self.me_conventions['transposed_mirrored'])
        self._net = None
        self._share = None
        return self

    def to(self, device):
        self.device = torch.device(device)
        self._net = None
        return self

    def cuda(self):
        return self.to('cuda')

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError('libdgr_hip implements the inference path only (eval-mode batch norm)')
        return self.eval()

    def share_weights(self, other):
        """Use the device-resident weights of `other` (the same architecture, already loaded) instead of a copy of its
        own: for one model object per HIP stream / library context over ONE weight set (dgr_net_share)."""
        if (type(other), other.D, other.in_channels, other.out_channels, other.conv1_kernel_size) != \
                (type(self), self.D, self.in_channels, self.out_channels, self.conv1_kernel_size):
            raise ValueError('share_weights: the other model is a different network')
        if other._net is None:
            # the source's library handle belongs to the context of the thread that CREATED it: resolving it lazily from the
            # sharer's thread would bind the loader's net to the wrong context (and two sharers could race on `other._net`)
            raise RuntimeError('share_weights: the other model has no device-resident weights yet -- run it once, or call '
                               'its _handle(), on the thread that owns it before sharing')
        self.normalize_feature = other.normalize_feature
        self._state = None
        self._share = other
        self._net = None
        return self

    def _handle(self):
        if self._net is None:
            share = getattr(self, '_share', None)
            if share is not None:
                self._net = ops.NetHandle(None, self.D, self.in_channels, self.out_channels, self.conv1_kernel_size,
                                          self.normalize_feature, self.device, share_from=share._net)
                return self._net
            if self._state is None:
                raise RuntimeError('load_state_dict() must be called before the first forward')
            self._net = ops.NetHandle(self._state, self.D, self.in_channels, self.out_channels,
                                      self.conv1_kernel_size, self.normalize_feature, self.device)
        return self._net

    def forward(self, x):
        if not isinstance(x, SparseTensor):
            raise TypeError('expected a deepglobalregistration_amd.SparseTensor')
        out = self._handle().forward(x.C, x.F)
        return SparseTensor(out, coordinates=x.C, device=self.device)

    __call__ = forward


class ResUNetBN2C(ResUNet2):
    NORM_TYPE = 'BN'
    CHANNELS = [None, 32, 64, 128, 256]
    TR_CHANNELS = [None, 64, 64, 64, 128]
