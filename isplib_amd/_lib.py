"""Loads the two in-tree shared objects; fails loudly when either is missing.

``libisplib_hip.so``  -- the C ABI (include/isplib_hip.h), HIP kernels for gfx950.
``_fusedmm_hip.so``   -- ``torch.ops.isplib.*`` on top of it.

Mirrors the reference's load step (isplib/__init__.py:18-28), which raises
ImportError when no ``_fusedmm_{cuda,cpu}`` library sits next to the package.
There is no CPU fallback: without the HIP extension nothing in this package works.
"""
from __future__ import annotations

import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
CABI_PATH = os.path.join(_PKG, "libisplib_hip.so")
OPS_PATH = os.path.join(_PKG, "_fusedmm_hip.so")

_cdll = None
_ops_loaded = False


def _missing(path: str) -> ImportError:
    return ImportError(
        f"isplib_amd: native library '{os.path.basename(path)}' not found in {_PKG}. "
        "Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C isplib_amd/csrc`). There is no CPU fallback."
    )


def cdll() -> ctypes.CDLL:
    """ctypes handle of the C ABI (torch must already be imported so that the
    process-wide HIP runtime is torch's)."""
    global _cdll
    if _cdll is None:
        if not os.path.exists(CABI_PATH):
            raise _missing(CABI_PATH)
        _cdll = ctypes.CDLL(CABI_PATH, mode=ctypes.RTLD_GLOBAL)
    return _cdll


def load_ops() -> None:
    """Register torch.ops.isplib.* (idempotent)."""
    global _ops_loaded
    if _ops_loaded:
        return
    if not os.path.exists(OPS_PATH):
        raise _missing(OPS_PATH)
    cdll()
    torch.ops.load_library(OPS_PATH)
    _ops_loaded = True
