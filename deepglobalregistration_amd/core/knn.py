"""`find_knn_gpu` with the reference signature (core/knn.py:23-74), executed by the tiled
brute-force HIP kernel.  Only knn=1 is on the inference path
(core/deep_global_registration.py:175-179)."""
from .. import ops


def find_knn_gpu(F0, F1, nn_max_n=-1, knn=1, return_distance=False):
    """Chunked branch (`nn_max_n > 1`): L2 distances, outputs shaped [N0,1]; unchunked branch:
    squared L2, indices [N0], distances [N0,1] -- exactly the reference's two conventions.  The
    chunking itself is unnecessary here (nothing of size chunk x N1 x C is materialised)."""
    if knn != 1:
        raise NotImplementedError('only knn=1 is implemented (the only value the DGR path uses)')
    chunked = nn_max_n > 1
    idx, dist = ops.knn1(F0, F1, squared=not chunked, return_distance=True)
    if chunked:
        idx, dist = idx.unsqueeze(1), dist.unsqueeze(1)
    else:
        dist = dist.unsqueeze(1)
    return (idx, dist) if return_distance else idx
