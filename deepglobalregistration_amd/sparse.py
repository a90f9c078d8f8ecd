"""Minimal stand-in for `ME.SparseTensor` on the DGR inference path: a COO coordinate matrix
(int32 [N,1+D], batch column first) plus a feature matrix [N,C], both resident in HBM.
The reference builds it at core/deep_global_registration.py:167,214 and reads `.F` at :169,217."""
import numpy as np
import torch


class SparseTensor:
    def __init__(self, features, coordinates=None, device=None, **kwargs):
        if coordinates is None:
            coordinates = kwargs.get('coords')        # ME 0.4 spelling
        if coordinates is None:
            raise ValueError('coordinates are required')
        if not torch.is_tensor(features):
            features = torch.as_tensor(np.asarray(features))
        if not torch.is_tensor(coordinates):
            coordinates = torch.as_tensor(np.asarray(coordinates))
        if device is None:
            device = features.device if features.is_cuda else coordinates.device
        self.device = torch.device(device)
        self.F = features.to(self.device, torch.float32).contiguous()
        self.C = coordinates.to(self.device, torch.int32).contiguous()
        if self.F.dim() != 2 or self.C.dim() != 2 or self.F.shape[0] != self.C.shape[0]:
            raise ValueError(f'features {tuple(self.F.shape)} and coordinates {tuple(self.C.shape)} '
                             'must be row aligned 2-D matrices')

    @property
    def D(self):
        return self.C.shape[1] - 1

    def __len__(self):
        return self.F.shape[0]
