"""isplib_amd -- MI355X-native SpMM aggregation backend behind iSpLib's surface.

    from isplib_amd import *            # reference: `from isplib import *` (README.md:68)
    iSpLibPlugin.patch_pyg()            # torch_sparse.matmul -> HIP kernels

Only the one hot path of the reference is here: SpMM(sum|mean|max|min) forward
and its autograd backward.  Importing the package loads the two in-tree
shared objects and raises ImportError when they are missing (no CPU fallback).
"""
from . import _lib

_lib.load_ops()

from . import cabi  # noqa: E402
from .plugin import fusedmm, gcn_norm_matmul, iSpLibPlugin, isplib_autotune, matmul, spmm_autotuned  # noqa: E402
from .sparse import SparseStorage, SparseTensor  # noqa: E402

__version__ = "0.1.0"
__all__ = ["iSpLibPlugin", "isplib_autotune", "matmul", "spmm_autotuned", "gcn_norm_matmul", "fusedmm", "SparseTensor", "SparseStorage",
           "cabi"]
