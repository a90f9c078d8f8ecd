"""Restatement of the MinkowskiEngine 0.5.4 semantics the DGR hot path relies on.

TEST INFRASTRUCTURE (see oracle/__init__.py).  **Parity unpinned**: ME's
source is not under /root/reference (un-vendored pip dependency,
`requirements.txt:24`), so every function here restates ME's published
behaviour and is anchored on the reference's call sites:

* `sparse_quantize`      <- `core/deep_global_registration.py:152`
* `batched_coordinates`  <- `core/deep_global_registration.py:158`
* `stride_coords`        <- every `stride=2` conv, `model/resunet.py:461-507`
* `kernel_offsets`       <- `ME.KernelGenerator(... HYPER_CUBE ...)`,
                            `model/residual_block.py:31-36`
* `kernel_map`           <- `ME.MinkowskiConvolution` forward,
                            `model/residual_block.py:38-44`
* `transposed_kernel_map`<- `ME.MinkowskiConvolutionTranspose`,
                            `model/residual_block.py:72-80`

Each unverifiable convention is in exactly one function so it can be flipped
if a real MinkowskiEngine becomes available.
"""
import numpy as np


# ----------------------------------------------------------------------------
# A1  ME.utils.sparse_quantize(coords, return_index=True)
# ----------------------------------------------------------------------------
def first_occurrence_unique(rows):
    """Indices of the first occurrence of every distinct row, ascending."""
    rows = np.ascontiguousarray(rows)
    _, first = np.unique(rows, axis=0, return_index=True)
    return np.sort(first)


def sparse_quantize(coords, return_index=False):
    """floor -> int32, keep the FIRST occurrence of each voxel, in ascending
    original-row order; returns (unique_coords, index) like ME 0.5.x.

    `coords` is already divided by the voxel size by the caller
    (`deep_global_registration.py:152`); floor is taken in the input dtype."""
    disc = np.floor(np.asarray(coords)).astype(np.int32)
    sel = first_occurrence_unique(disc)
    if return_index:
        return disc[sel], sel
    return disc[sel]


# ----------------------------------------------------------------------------
# A2  ME.utils.batched_coordinates
# ----------------------------------------------------------------------------
def batched_coordinates(coords_list):
    """int32 [N, 1+D] with the batch index in column 0."""
    out = []
    for b, c in enumerate(coords_list):
        c = np.asarray(c).astype(np.int32)
        out.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], axis=1))
    return np.concatenate(out, axis=0)


# ----------------------------------------------------------------------------
# A4  strided output coordinate map
# ----------------------------------------------------------------------------
def stride_coords(coords, new_ts):
    """Output map of a stride-2 conv: unique(floor(c / new_ts) * new_ts) over
    the spatial columns (floor division also for negatives), batch column
    untouched.  Row order = first occurrence (internal in ME; results do not
    depend on it)."""
    coords = np.asarray(coords)
    out = coords.copy()
    out[:, 1:] = np.floor_divide(coords[:, 1:], new_ts) * new_ts
    sel = first_occurrence_unique(out)
    return out[sel]


# ----------------------------------------------------------------------------
# A5  kernel offsets: HYPER_CUBE, odd kernel size, dilation 1
# ----------------------------------------------------------------------------
def kernel_offsets(D, ks):
    """[K, D] integer offsets in units of the tensor stride.  Enumeration:
    FIRST spatial dimension fastest, j = sum_d (delta_d + ks//2) * ks**d."""
    assert ks % 2 == 1, "only odd kernel sizes are on the DGR path"
    K = ks ** D
    offs = np.zeros((K, D), np.int64)
    j = np.arange(K)
    for d in range(D):
        offs[:, d] = (j % ks) - ks // 2
        j = j // ks
    return offs


# ----------------------------------------------------------------------------
# coordinate -> row lookup (plain sorting based; no hashing on the CPU side)
# ----------------------------------------------------------------------------
class CoordIndex:
    """Exact row lookup over an int coordinate array [N, 1+D]."""

    def __init__(self, coords, pad):
        c = np.asarray(coords).astype(np.int64)
        self.lo = c.min(axis=0) - pad
        hi = c.max(axis=0) + pad
        ext = hi - self.lo + 1
        total = 1
        for e in ext:
            total *= int(e)
        self.packable = total < (1 << 62)
        if self.packable:
            self.mul = np.ones(c.shape[1], np.int64)
            for d in range(c.shape[1] - 2, -1, -1):
                self.mul[d] = self.mul[d + 1] * ext[d + 1]
            self.ext = ext
            keys = ((c - self.lo) * self.mul).sum(axis=1)
            self.order = np.argsort(keys, kind='stable')
            self.sorted_keys = keys[self.order]
        else:  # pragma: no cover - only for absurd coordinate ranges
            self.table = {tuple(r): i for i, r in enumerate(c.tolist())}

    def lookup(self, query):
        """Row index of each query coordinate, -1 where absent."""
        q = np.asarray(query).astype(np.int64)
        if not self.packable:  # pragma: no cover
            return np.array([self.table.get(tuple(r), -1) for r in q.tolist()], np.int64)
        rel = q - self.lo
        inside = np.all((rel >= 0) & (rel < self.ext), axis=1)
        keys = (rel * self.mul).sum(axis=1)
        pos = np.searchsorted(self.sorted_keys, keys)
        pos = np.minimum(pos, len(self.sorted_keys) - 1)
        hit = inside & (self.sorted_keys[pos] == keys)
        return np.where(hit, self.order[pos], -1)


def kernel_map(coords_in, coords_out, D, ks, ts_in):
    """Pairs (k, in, out) with coord_in[in] == coord_out[out] + delta_k*ts_in.
    Returned as three int64 arrays sorted by (k, out)."""
    offs = kernel_offsets(D, ks) * int(ts_in)
    index = CoordIndex(coords_in, pad=int(ts_in) * (ks // 2) + 1)
    cout = np.asarray(coords_out).astype(np.int64)
    ks_, ins, outs = [], [], []
    for k in range(len(offs)):
        q = cout.copy()
        q[:, 1:] += offs[k]
        r = index.lookup(q)
        o = np.nonzero(r >= 0)[0]
        if len(o):
            ks_.append(np.full(len(o), k, np.int64))
            ins.append(r[o])
            outs.append(o)
    if not ks_:
        z = np.zeros(0, np.int64)
        return z, z, z
    return np.concatenate(ks_), np.concatenate(ins), np.concatenate(outs)


# ----------------------------------------------------------------------------
# A6  transposed convolution kernel map
# ----------------------------------------------------------------------------
def transposed_kernel_map(coords_coarse, coords_fine, D, ks, ts_fine):
    """Kernel map of `MinkowskiConvolutionTranspose(kernel 3, stride 2)`: the
    map of the *forward* strided conv fine->coarse with in/out swapped and the
    same offset index: out_fine[f] += in_coarse[c] * W[k] where
    coord_fine[f] == coord_coarse[c] + delta_k * ts_fine.
    Returned sorted by (k, out=fine)."""
    k, fine, coarse = kernel_map(coords_fine, coords_coarse, D, ks, ts_fine)
    # forward map: in=fine, out=coarse.  Swap roles for the transposed conv.
    order = np.lexsort((fine, k))
    return k[order], coarse[order], fine[order]
