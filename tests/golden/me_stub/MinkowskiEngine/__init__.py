"""A stand-in for the few MinkowskiEngine 0.5.4 names `model/*.py` and `core/deep_global_registration.py` of the
reference touch -- so that the reference's OWN model classes and its `register()` execute in this container
(tests/golden/make_golden_model.py, make_golden_register.py) and pin the oracle's restatement of their topology, op
order, state-dict layout and glue.  TEST INFRASTRUCTURE, imported by those generators only (never by the package, the
oracle or a test).

What it is NOT: MinkowskiEngine.  The arithmetic below restates ME's published semantics of a generalized sparse
convolution in the plainest form available -- Python dictionaries from coordinate tuples to rows, one matrix product
per kernel offset -- and shares no code with `oracle/me_semantics.py` (sorted-key searches on packed integers), so the
two are independent implementations of one reading of ME; whether that reading is ME's is still unpinned
(oracle/__init__.py).  Conventions restated here:

* a tensor at stride s lives on the lattice s*Z^D; a stride-2 convolution maps it to floor(c / 2s) * 2s;
* HYPER_CUBE kernels of odd size k: offsets {-(k-1)/2 .. (k-1)/2}^D times the INPUT tensor stride, centred on the
  output coordinate, kernel index with the first spatial axis fastest;
* a transposed convolution writes onto the existing coordinate set of the finer stride through the forward map
  swapped (in = coarse row, out = fine row, same kernel index);
* `utils.sparse_quantize` keeps the FIRST point of every voxel, voxels in the order of their first points;
* parameters: `kernel` [K, Cin, Cout] ([Cin, Cout] when K = 1), `bias` [1, Cout]; `MinkowskiBatchNorm.bn` is a
  `torch.nn.BatchNorm1d` on the feature matrix.
"""
import enum
import itertools

import numpy as np
import torch
import torch.nn as nn


class RegionType(enum.Enum):
    HYPER_CUBE = 0
    HYPER_CROSS = 1


class KernelGenerator:
    def __init__(self, kernel_size=-1, stride=1, dilation=1, is_transpose=False, region_type=RegionType.HYPER_CUBE,
                 region_offsets=None, expand_coordinates=False, axis_types=None, dimension=-1):
        assert dimension > 0 and dilation == 1 and region_type is RegionType.HYPER_CUBE
        self.kernel_size, self.stride, self.is_transpose, self.dimension = kernel_size, stride, is_transpose, dimension


class CoordinateManager:
    """Coordinate sets by tensor stride (ME keys them by (stride, string id); the model only ever has one per stride)."""

    def __init__(self, D):
        self.D = D
        self.rows = {}          # stride -> list of coordinate tuples (batch, x_1 .. x_D)
        self.index = {}         # stride -> {tuple: row}

    def insert(self, stride, tuples):
        assert stride not in self.rows
        self.rows[stride] = list(tuples)
        self.index[stride] = {c: r for r, c in enumerate(self.rows[stride])}
        assert len(self.index[stride]) == len(self.rows[stride]), 'duplicate coordinates'

    def stride_set(self, stride):
        """The coordinate set at 2 x `stride` (created on first use, rows in first-occurrence order)."""
        s2 = 2 * stride
        if s2 not in self.rows:
            seen = {}
            for c in self.rows[stride]:
                seen.setdefault((c[0],) + tuple((x // s2) * s2 for x in c[1:]), None)     # // floors, like ME
            self.insert(s2, seen.keys())
        return s2


class SparseTensor:
    def __init__(self, features, coordinates=None, tensor_stride=1, coordinate_map_key=None, coordinate_manager=None,
                 device=None):
        self.F = features
        if coordinate_manager is None:
            c = np.asarray(coordinates.cpu() if torch.is_tensor(coordinates) else coordinates)
            coordinate_manager = CoordinateManager(c.shape[1] - 1)
            coordinate_manager.insert(tensor_stride, [tuple(int(v) for v in row) for row in c])
            coordinate_map_key = tensor_stride
        self.coordinate_manager, self.coordinate_map_key = coordinate_manager, coordinate_map_key
        assert len(coordinate_manager.rows[coordinate_map_key]) == len(features)

    @property
    def C(self):
        return torch.tensor(self.coordinate_manager.rows[self.coordinate_map_key], dtype=torch.int32)

    @property
    def tensor_stride(self):
        return [self.coordinate_map_key] * self.coordinate_manager.D

    @property
    def D(self):
        return self.coordinate_manager.D

    def _like(self, F, key=None):
        return SparseTensor(F, coordinate_map_key=self.coordinate_map_key if key is None else key,
                            coordinate_manager=self.coordinate_manager)

    def __add__(self, other):
        assert other.coordinate_map_key == self.coordinate_map_key and other.coordinate_manager is self.coordinate_manager
        return self._like(self.F + other.F)

    def __iadd__(self, other):
        assert other.coordinate_map_key == self.coordinate_map_key and other.coordinate_manager is self.coordinate_manager
        self.F = self.F + other.F
        return self


def cat(*tensors):
    t0 = tensors[0]
    assert all(t.coordinate_map_key == t0.coordinate_map_key and t.coordinate_manager is t0.coordinate_manager for t in tensors)
    return t0._like(torch.cat([t.F for t in tensors], dim=1))


class MinkowskiNetwork(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.D = D


def _offsets(kernel_size, D):
    """HYPER_CUBE offsets of an odd kernel in kernel-index order: the FIRST spatial axis runs fastest."""
    assert kernel_size % 2 == 1
    h = kernel_size // 2
    return [tuple(reversed(o)) for o in itertools.product(range(-h, h + 1), repeat=D)]


class _ConvBase(nn.Module):
    transposed = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, convolution_mode=None, dimension=None):
        super().__init__()
        assert dimension is not None and dilation == 1 and stride in (1, 2)
        if kernel_generator is not None:
            assert kernel_generator.kernel_size == kernel_size and kernel_generator.dimension == dimension
            assert kernel_generator.stride == stride and kernel_generator.is_transpose == self.transposed
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dimension = kernel_size, stride, dimension
        K = kernel_size ** dimension
        self.kernel = nn.Parameter(torch.zeros((in_channels, out_channels) if K == 1 else (K, in_channels, out_channels)))
        self.bias = nn.Parameter(torch.zeros(1, out_channels)) if bias else None

    def _finish(self, x, out, key):
        if self.bias is not None:
            out = out + self.bias
        return x._like(out, key)


class MinkowskiConvolution(_ConvBase):
    def forward(self, x):
        cm, s = x.coordinate_manager, x.coordinate_map_key
        if self.kernel_size == 1:
            assert self.stride == 1
            return self._finish(x, x.F @ self.kernel, s)
        s_out = s if self.stride == 1 else cm.stride_set(s)
        src, dst = cm.index[s], cm.rows[s_out]
        out = torch.zeros(len(dst), self.out_channels, dtype=x.F.dtype)
        for k, o in enumerate(_offsets(self.kernel_size, self.dimension)):
            i_rows, o_rows = [], []
            for r, u in enumerate(dst):                      # out[u] += W_k^T in[u + o * s]   (s = INPUT stride)
                i = src.get((u[0],) + tuple(a + b * s for a, b in zip(u[1:], o)))
                if i is not None:
                    i_rows.append(i)
                    o_rows.append(r)
            if i_rows:
                out.index_add_(0, torch.tensor(o_rows), x.F[torch.tensor(i_rows)] @ self.kernel[k])
        return self._finish(x, out, s_out)


class MinkowskiConvolutionTranspose(_ConvBase):
    transposed = True

    def forward(self, x):
        cm, s = x.coordinate_manager, x.coordinate_map_key
        assert self.stride == 2 and s % 2 == 0 and (s // 2) in cm.rows, 'the finer coordinate set must exist'
        s_out = s // 2
        src, dst = cm.rows[s], cm.index[s_out]
        out = torch.zeros(len(dst), self.out_channels, dtype=x.F.dtype)
        for k, o in enumerate(_offsets(self.kernel_size, self.dimension)):
            i_rows, o_rows = [], []
            for i, u in enumerate(src):                      # the forward pair (fine u + o * s_out -> coarse u), swapped
                r = dst.get((u[0],) + tuple(a + b * s_out for a, b in zip(u[1:], o)))
                if r is not None:
                    i_rows.append(i)
                    o_rows.append(r)
            if i_rows:
                out.index_add_(0, torch.tensor(o_rows), x.F[torch.tensor(i_rows)] @ self.kernel[k])
        return self._finish(x, out, s_out)


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def forward(self, x):
        return x._like(self.bn(x.F))


class MinkowskiInstanceNorm(nn.Module):          # constructible (model/common.py:14); no pinned model runs it
    def __init__(self, num_features):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(1, num_features))
        self.bias = nn.Parameter(torch.zeros(1, num_features))

    def forward(self, x):
        raise NotImplementedError


class MinkowskiReLU(nn.Module):
    def forward(self, x):
        return x._like(torch.relu(x.F))


class MinkowskiELU(nn.Module):
    def forward(self, x):
        return x._like(torch.nn.functional.elu(x.F))


from . import MinkowskiFunctional, utils  # noqa: E402,F401
