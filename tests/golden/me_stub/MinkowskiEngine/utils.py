"""`MinkowskiEngine.utils` names `core/deep_global_registration.py:152,158` calls (see the package docstring)."""
import numpy as np
import torch


def sparse_quantize(coordinates, features=None, labels=None, ignore_label=-100, return_index=False,
                    return_inverse=False, quantization_size=None):
    """Floor to the voxel lattice; one point per voxel: the first, voxels in the order of their first points."""
    assert features is None and labels is None and not return_inverse and quantization_size is None
    c = coordinates.numpy() if torch.is_tensor(coordinates) else np.asarray(coordinates)
    cells = np.floor(c).astype(np.int32)
    first = {}
    for r, cell in enumerate(map(tuple, cells.tolist())):
        first.setdefault(cell, r)
    index = np.fromiter(first.values(), np.int64, len(first))
    return (cells[index], index) if return_index else cells[index]


def batched_coordinates(coords, dtype=torch.int32, device=None):
    """[N_b, D] per batch entry -> [sum N_b, 1 + D] with the batch index in front."""
    rows = [torch.cat((torch.full((len(c), 1), b, dtype=dtype), torch.as_tensor(c).to(dtype)), dim=1) for b, c in enumerate(coords)]
    return torch.cat(rows, dim=0)
