"""oracle -- CPU checker for the SpMM aggregation path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``isplib_amd``) never does and has
no CPU fallback.

Contents
--------
* ``fusedMM_csr`` (ctypes binding of ``libfusedmm_oracle.so``, the C restatement
  in ``fusedmm_oracle.c`` -- see its header for what it follows and its parity
  status: **parity unpinned** at the kernel-body level because the reference's
  kernel library is an absent, unpinned third-party dependency).
* ``spmm_fw``: NumPy restatement of the reference launcher
  ``fusedmm_spmm_fw`` (/root/reference/csrc/fusedmm.cpp:113-203): output
  initialisation, arg sentinel, op-message per reduction, the C call.
* ``spmm_sum_bw / spmm_mean_bw / spmm_minmax_bw``: NumPy restatements of the
  reference autograd backward formulas (csrc/fusedmm.cpp:258-293, 340-383,
  410-451 / 477-517).
* ``scan_spmm``: a deliberately naive pure-Python/NumPy sequential scan used as
  an independent check of the C file on small inputs.
* ``csr_transpose``: the CSR->CSC operands torch_sparse's storage would hand
  the reference (row, rowcount, colptr, csr2csc), stable order.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfusedmm_oracle.so")
REF_LIB_PATH = os.path.join(_HERE, "_ref", "_fusedmm_ref.so")

# op-message nibbles: values per /root/reference/csrc/fusedMM.h:18-74
VOP_COPY_RHS, ROP_NOOP, SOP_COPY = 0x2, 0x00, 0x100
VSC_MUL, VSC_MEAN = 0x1000, 0x3000
AOP_ADD, AOP_MAX, AOP_MIN = 0x10000, 0x20000, 0x30000

# reduction codes of fusedmm_spmm_fw (csrc/fusedmm.cpp:168-186)
REDUCE_CODE = {"sum": 0, "add": 0, "max": 1, "min": 2, "mean": 3}

FLT_LOWEST = np.float32(np.finfo(np.float32).min)
FLT_MAX = np.float32(np.finfo(np.float32).max)


def build(force: bool = False) -> str:
    """Compile the C restatement (and oracle/_ref when the reference tree is here)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "fusedmm_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, os.path.join(_HERE, "libfusedmm_oracle.so")])
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, f32, vp = ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
        L.fusedMM_csr.restype = ctypes.c_int
        L.fusedMM_csr.argtypes = [ctypes.c_int32, i64, i64, i64, f32, i64, i64, i64,
                                  vp, vp, vp, vp, vp, i64, vp, i64, f32, vp, i64, vp]
        L.performDummySpMM.restype = None
        L.performDummySpMM.argtypes = [i64]
        L.oracle_sddmm_csr.restype = ctypes.c_int
        L.oracle_sddmm_csr.argtypes = [i64, i64, vp, vp, vp, vp, i64, vp, i64, ctypes.c_int, vp]
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_fusedMM_csr_udef.restype = ctypes.c_int
        L.oracle_fusedMM_csr_udef.argtypes = [ctypes.c_int32, i64, i64, i64, vp, vp, vp, vp, vp, i64, vp, i64, vp, i64, vp,
                                              ctypes.c_int, f32]
        _lib = L
    return _lib


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def message(reduce: str) -> int:
    code = REDUCE_CODE[reduce]
    vsc = VSC_MEAN if code == 3 else VSC_MUL
    aop = {0: AOP_ADD, 1: AOP_MAX, 2: AOP_MIN, 3: AOP_ADD}[code]
    return VOP_COPY_RHS | ROP_NOOP | SOP_COPY | vsc | aop


def fusedMM_csr(imsg, m, n, k, val, indx, rowptr, y, z, z_arg=None) -> int:
    """Raw call with the reference's argument pattern (csrc/fusedmm.cpp:198):
    pntrb = rowptr, pntre = rowptr + 1, ldy = ldz = k, alpha = 1, beta = 0."""
    assert val.dtype == np.float32 and y.dtype == np.float32 and z.dtype == np.float32
    assert indx.dtype == np.int64 and rowptr.dtype == np.int64
    assert y.flags.c_contiguous and z.flags.c_contiguous
    dummy = np.zeros(1, np.float32)
    pb = rowptr.ctypes.data
    return lib().fusedMM_csr(int(imsg), m, n, k, 1.0, val.size, m, n,
                             _p(val), _p(indx), ctypes.c_void_p(pb), ctypes.c_void_p(pb + 8),
                             _p(dummy), k, _p(y), k, 0.0, _p(z), k, _p(z_arg))


def fusedmm_general(imsg, rowptr, col, value, x, y, sop_udef=0, sop_param=0.0):
    """The generic five-stage pipeline (fusedmm_oracle.c, oracle_fusedMM_csr_udef) -> (status, z, z_arg | None).
    z is written, not accumulated into; z_arg only for AOP_MAX / AOP_MIN."""
    m, k = rowptr.size - 1, y.shape[1]
    z = np.empty((m, k), np.float32)
    arg = np.empty((m, k), np.int64) if ((imsg >> 16) & 0xF) in (2, 3) else None
    pb = rowptr.ctypes.data
    st = lib().oracle_fusedMM_csr_udef(int(imsg), m, k, col.size, _p(value), _p(col), ctypes.c_void_p(pb),
                                       ctypes.c_void_p(pb + 8), _p(x), k, _p(y), k, _p(z), k, _p(arg),
                                       int(sop_udef), float(sop_param))
    return st, z, arg


def spmm_fw(rowptr, col, value, mat, reduce="sum"):
    """fusedmm_spmm_fw (csrc/fusedmm.cpp:113-203) on NumPy arrays.

    Returns (out, arg_out|None).  ``value`` must be present, as at :126,131.
    """
    rowptr = np.ascontiguousarray(rowptr, np.int64)
    col = np.ascontiguousarray(col, np.int64)
    value = np.ascontiguousarray(value, np.float32)
    mat = np.ascontiguousarray(mat, np.float32)          # :140
    m = rowptr.size - 1                                  # :120
    n, k = mat.shape[-2], mat.shape[-1]                  # :121-122
    code = REDUCE_CODE[reduce]
    if code == 1:
        out = np.full((m, k), FLT_LOWEST, np.float32)    # :148
    elif code == 2:
        out = np.full((m, k), FLT_MAX, np.float32)       # :150
    else:
        out = np.zeros((m, k), np.float32)               # :152
    arg = np.full((m, k), col.size, np.int64) if code in (1, 2) else None  # :171,177
    st = fusedMM_csr(message(reduce), m, n, k, value, col, rowptr, mat, out, arg)
    if st != 0:
        raise RuntimeError(f"fusedMM_csr returned status {st}")
    return out, arg


def set_empty_row(mode: str = "zero") -> None:
    """What an EMPTY row of max / min holds: "zero" (default; torch_sparse's CPU semantic) or "init" (the launcher's
    pre-fill, lowest() / max() of csrc/fusedmm.cpp:147-150, left untouched).  The HIP path has the same switch."""
    if mode not in ("zero", "init"):
        raise ValueError("empty-row mode: 'zero' or 'init'")
    lib().oracle_set_empty_row(1 if mode == "init" else 0)


def spmm_sum_timed(rowptr, col, value, mat, reps=5):
    """bench.py's CPU baseline: SpMM-sum with NUMA-aware placement and a static nnz-balanced row partition
    (fusedmm_oracle.c: oracle_spmm_sum_timed).  Returns (seconds per pass [reps], out)."""
    rowptr = np.ascontiguousarray(rowptr, np.int64)
    col = np.ascontiguousarray(col, np.int64)
    value = np.ascontiguousarray(value, np.float32)
    mat = np.ascontiguousarray(mat, np.float32)
    m, (n, k) = rowptr.size - 1, mat.shape
    secs = np.zeros(reps, np.float64)
    out = np.empty((m, k), np.float32)
    fn = lib().oracle_spmm_sum_timed
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int64] * 4 + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    st = fn(m, n, k, col.size, _p(value), _p(col), _p(rowptr), _p(mat), int(reps), _p(secs), _p(out))
    if st != 0:
        raise RuntimeError(f"oracle_spmm_sum_timed returned status {st}")
    return secs, out


def csr_transpose(rowptr, col, ncols):
    """row, rowcount, colptr, csr2csc as torch_sparse's SparseStorage builds them
    (stable sort of the CSR entries by column) -- the operands the reference
    wrapper reads at isplib/__init__.py:58-73."""
    rowptr = np.asarray(rowptr, np.int64)
    col = np.asarray(col, np.int64)
    m = rowptr.size - 1
    rowcount = np.diff(rowptr)
    row = np.repeat(np.arange(m, dtype=np.int64), rowcount)
    csr2csc = np.argsort(col, kind="stable").astype(np.int64)
    colptr = np.zeros(ncols + 1, np.int64)
    np.cumsum(np.bincount(col, minlength=ncols), out=colptr[1:])
    return row, rowcount, colptr, csr2csc


def spmm_sum_bw(rowptr, col, value, ncols, grad_out):
    """dX of SpMM-sum (csrc/fusedmm.cpp:285 with the operands cached at
    isplib/__init__.py:76-80): the same kernel on A^T in CSR form."""
    row, _, colptr, csr2csc = csr_transpose(rowptr, col, ncols)
    val_t = np.asarray(value, np.float32)[csr2csc]       # 'index_select'
    row_t = row[csr2csc]                                 # 'row_select'
    return spmm_fw(colptr, row_t, val_t, grad_out, "sum")[0]


def mean_bw_weights(rowptr, col, value, ncols):
    """A^T-ordered (new_row, new_rowcount) of the mean backward, in the form the
    commented C++ original states (csrc/fusedmm.cpp:357-364):
    value[csr2csc] / max(rowcount,1)[row[csr2csc]].  The Python at
    isplib/__init__.py:86-91 indexes rowcount with the un-permuted row (a
    defect); not reproduced."""
    row, rowcount, colptr, csr2csc = csr_transpose(rowptr, col, ncols)
    new_row = row[csr2csc]
    deg = np.maximum(rowcount, 1).astype(np.float32)[new_row]
    new_rowcount = (np.asarray(value, np.float32)[csr2csc] / deg).astype(np.float32)
    return colptr, new_row, new_rowcount


def spmm_mean_bw(rowptr, col, value, ncols, grad_out):
    """dX of SpMM-mean (csrc/fusedmm.cpp:375)."""
    colptr, new_row, new_rowcount = mean_bw_weights(rowptr, col, value, ncols)
    return spmm_fw(colptr, new_row, new_rowcount, grad_out, "sum")[0]


def spmm_minmax_bw(col, value, mat, arg_out, grad_out):
    """(grad_value, grad_mat) of SpMM-max/min, csrc/fusedmm.cpp:410-451 (ATen
    gather/scatter_add restated with np.add.at, CPU order = row-major)."""
    col = np.asarray(col, np.int64)
    value = np.asarray(value, np.float32)
    mat = np.asarray(mat, np.float32)
    nnz = col.size
    invalid = arg_out == nnz                              # :417
    arg = np.where(invalid, 0, arg_out)                   # :418
    ind = col[arg]                                        # :422,442
    kk = np.broadcast_to(np.arange(mat.shape[1]), arg.shape)
    g_val = mat[ind, kk] * grad_out                       # :423-424
    g_val = np.where(invalid, np.float32(0), g_val).astype(np.float32)
    grad_value = np.zeros(nnz, np.float32)
    np.add.at(grad_value, arg.ravel(), g_val.ravel())     # :427-428
    v = (value[arg] * grad_out).astype(np.float32)        # :434-437
    v = np.where(invalid, np.float32(0), v).astype(np.float32)
    grad_mat = np.zeros_like(mat)
    np.add.at(grad_mat, (ind.ravel(), kk.ravel()), v.ravel())  # :444-445
    return grad_value, grad_mat


def sddmm(rowptr, col, mat, grad_out, mean=False):
    """dA the reference leaves commented out (csrc/fusedmm.cpp:270,351)."""
    rowptr = np.ascontiguousarray(rowptr, np.int64)
    col = np.ascontiguousarray(col, np.int64)
    mat = np.ascontiguousarray(mat, np.float32)
    grad_out = np.ascontiguousarray(grad_out, np.float32)
    m, k = rowptr.size - 1, mat.shape[1]
    out = np.zeros(col.size, np.float32)
    pb = rowptr.ctypes.data
    st = lib().oracle_sddmm_csr(m, k, _p(col), ctypes.c_void_p(pb), ctypes.c_void_p(pb + 8),
                                _p(mat), k, _p(grad_out), k, int(bool(mean)), _p(out))
    assert st == 0
    return out


def scan_spmm(rowptr, col, value, mat, reduce="sum"):
    """Independent, deliberately naive sequential scan (small inputs only).
    fp32 arithmetic step by step, CSR order, strict compares."""
    rowptr = np.asarray(rowptr, np.int64)
    col = np.asarray(col, np.int64)
    value = np.asarray(value, np.float32)
    mat = np.asarray(mat, np.float32)
    m, k, nnz = rowptr.size - 1, mat.shape[1], col.size
    code = REDUCE_CODE[reduce]
    out = np.zeros((m, k), np.float32)
    arg = np.full((m, k), nnz, np.int64) if code in (1, 2) else None
    for i in range(m):
        b, e = int(rowptr[i]), int(rowptr[i + 1])
        if code in (0, 3):
            acc = np.zeros(k, np.float32)
            for j in range(b, e):
                # one fused multiply-add per element, like the -O3 -march=native C
                acc = (acc.astype(np.float64) + value[j].astype(np.float64) * mat[col[j]].astype(np.float64)).astype(np.float32)
            if code == 3:
                acc = acc / np.float32(max(e - b, 1))
            out[i] = acc
        else:
            if e <= b:
                continue                      # value 0, arg = nnz
            cur = np.full(k, FLT_LOWEST if code == 1 else FLT_MAX, np.float32)
            for j in range(b, e):
                t = value[j] * mat[col[j]]
                with np.errstate(invalid="ignore"):
                    win = (t > cur) if code == 1 else (t < cur)
                cur = np.where(win, t, cur)
                arg[i] = np.where(win, j, arg[i])
            out[i] = cur
    return out, arg
