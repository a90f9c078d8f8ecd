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


def find_knn_gpu_batch(F0, F1, len_batch, nn_max_n=-1, knn=1, return_distance=False, concat_results=False):
    """Interface of core/knn.py:106-140: one independent search per pair of a collated batch.  F0 / F1 hold
    the rows of all pairs back to back and `len_batch` lists (N0, N1) per pair
    (dataloader/base_loader.py:63-81).  Returns per-pair lists, or -- with `concat_results` -- single
    tensors whose indices address rows of the concatenated F1."""
    import itertools
    if knn != 1:
        raise NotImplementedError('only knn=1 is implemented (the only value the DGR path uses)')
    sizes = [(int(a), int(b)) for a, b in len_batch]
    first0 = [0] + list(itertools.accumulate(n0 for n0, _ in sizes))
    first1 = [0] + list(itertools.accumulate(n1 for _, n1 in sizes))
    # one library call for the whole batch (dgr_knn1_l2_batch); shapes per pair as find_knn_gpu returns them
    chunked = nn_max_n > 1
    idx_all, dist_all = ops.knn1_batch(F0, F1, first0, first1, squared=not chunked, return_distance=True)
    dist_all = dist_all.unsqueeze(1)
    if chunked:
        idx_all = idx_all.unsqueeze(1)
    if concat_results:
        return (idx_all, dist_all) if return_distance else idx_all
    idx = [idx_all[s0:s0 + n0] - s1 for (n0, _), s0, s1 in zip(sizes, first0, first1)]
    dist = [dist_all[s0:s0 + n0] for (n0, _), s0 in zip(sizes, first0)]
    return (idx, dist) if return_distance else idx


def find_knn_batch(F0, F1, len_batch, return_distance=False, nn_max_n=-1, knn=1, search_method=None,
                   concat_results=False):
    """core/knn.py:76-103.  Only the GPU method exists in this build."""
    if search_method is None or search_method == 'gpu':
        return find_knn_gpu_batch(F0, F1, len_batch=len_batch, nn_max_n=nn_max_n, knn=knn,
                                  return_distance=return_distance, concat_results=concat_results)
    if search_method == 'cpu':
        raise ValueError("Search method cpu is not available in the MI355X build (use 'gpu')")
    raise ValueError(f'Search method {search_method} not defined')
