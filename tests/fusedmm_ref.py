"""A second, independent statement of the generic FusedMM pipeline in vectorised NumPy (fp64), used to pin
oracle/fusedmm_oracle.c's generic path and, through it, the HIP kernel.  Written from the stage table in
include/isplib_hip.h, per edge instead of per row; no code shared with the C loops."""
import numpy as np


def sop_menu(kind, s, p):
    if kind == 1:
        return 1.0 / (1.0 + np.exp(-s))
    if kind == 2:
        return 1.0 - 1.0 / (1.0 + np.exp(-s))
    if kind == 3:
        return 1.0 / (1.0 + s)
    if kind == 4:
        return p * s
    if kind == 5:
        return np.exp(s)
    if kind == 6:
        return np.exp(np.where(s > 0, s, p * s))
    return s


def fusedmm(imsg, rowptr, col, val, x, y, sop_udef=0, sop_param=0.0):
    """-> (z fp64 [m,k], arg int64 [m,k] | None).  Ties: lowest CSR position; empty row under max/min: 0 / nnz."""
    vop, rop, sop, vsc, aop = imsg & 0xF, (imsg >> 4) & 0xF, (imsg >> 8) & 0xF, (imsg >> 12) & 0xF, (imsg >> 16) & 0xF
    m, k, nnz = rowptr.size - 1, y.shape[1], col.size
    row = np.repeat(np.arange(m), np.diff(rowptr))
    xe = (x[row] if x is not None else np.zeros((nnz, k))).astype(np.float64)
    ye = y[col].astype(np.float64)
    t = {1: xe, 2: ye, 3: xe + ye, 4: xe - ye, 5: ye - xe, 6: np.maximum(xe, ye), 7: np.minimum(xe, ye)}[vop]
    s = {0: np.ones(nnz), 1: (xe * t).sum(1), 2: xe.sum(1), 3: t.sum(1), 4: (xe * xe).sum(1), 5: (t * t).sum(1)}[rop]
    if sop == 1:
        s = (val if val is not None else np.ones(nnz)).astype(np.float64)
    elif sop == 0xF:
        s = sop_menu(sop_udef, s, sop_param)
    if vsc in (1, 3):
        t = s[:, None] * t
    elif vsc == 2:
        t = s[:, None] + t
    z = np.zeros((m, k))
    arg = None
    if aop == 1:
        np.add.at(z, row, t)
        if vsc == 3:
            z /= np.maximum(np.diff(rowptr), 1)[:, None]
    else:
        arg = np.full((m, k), nnz, np.int64)
        for i in range(m):
            b, e = rowptr[i], rowptr[i + 1]
            if e > b:
                seg = t[b:e]
                pick = seg.argmax(0) if aop == 2 else seg.argmin(0)      # first occurrence = lowest CSR position
                z[i] = seg[pick, np.arange(k)]
                arg[i] = b + pick
    return z, arg
