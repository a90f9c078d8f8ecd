"""MI355X-native Deep Global Registration inference path.

Python host code mirroring the reference's operator surface over the C ABI of
libdgr_hip.so (include/dgr_hip.h).  Importing the package is cheap and does not
need a GPU; the first call into an op loads the HIP library and fails loudly
if it is missing -- there is no CPU fallback.
"""
from .sparse import SparseTensor  # noqa: F401

__all__ = ['SparseTensor', 'DeepGlobalRegistration', 'load_model', 'find_knn_gpu',
           'GlobalRegistration', 'weighted_procrustes']


def __getattr__(name):
    if name == 'DeepGlobalRegistration':
        from .core.deep_global_registration import DeepGlobalRegistration
        return DeepGlobalRegistration
    if name == 'load_model':
        from .model import load_model
        return load_model
    if name == 'find_knn_gpu':
        from .core.knn import find_knn_gpu
        return find_knn_gpu
    if name in ('GlobalRegistration', 'weighted_procrustes'):
        from .core import registration
        return getattr(registration, name)
    raise AttributeError(name)
