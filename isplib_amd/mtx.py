"""MatrixMarket coordinate files <-> CSR, the graph format of the reference's tuner.

The reference tunes on `.mtx` files (autotuner/findbestk.py:34-38 hands `../dataset/<name>.mtx` to the
FusedMM timer; README.md:147-168 shows how a PyG adjacency is written out with fast_matrix_market,
which is not in this image).  This module is the host-side reader/writer for that format: `coordinate`
matrices with `real | integer | pattern` fields and `general | symmetric | skew-symmetric` symmetry.
Entries are ordered by (row, column) with equal keys kept in file order -- torch_sparse's order, the one
the SpMM kernels' tie rules are defined on -- and duplicates are kept, not summed.
"""
from __future__ import annotations

import io
from typing import Optional, Tuple

import numpy as np
import torch

_FIELDS = ("real", "integer", "pattern")
_SYMMETRIES = ("general", "symmetric", "skew-symmetric")


def _body(handle, n_cols: int, count: int) -> np.ndarray:
    """The `count` data lines as a float64 [count, n_cols] array (pandas' C parser when present: a Reddit-sized
    file has 10^8 lines and np.loadtxt needs minutes for it)."""
    if count == 0:
        return np.zeros((0, n_cols), np.float64)
    try:
        import pandas as pd
        frame = pd.read_csv(handle, sep=r"\s+", header=None, comment="%", nrows=count, dtype=np.float64,
                            engine="c", usecols=range(n_cols))
        data = frame.to_numpy()
    except ImportError:                      # pragma: no cover - pandas is in the image
        data = np.loadtxt(handle, dtype=np.float64, comments="%", max_rows=count, ndmin=2)[:, :n_cols]
    if data.shape[0] != count:
        raise ValueError(f"mtx: size line promises {count} entries, file holds {data.shape[0]}")
    return data


def read_mtx(path) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], Tuple[int, int]]:
    """-> (rowptr int64 [M+1], col int64 [nnz], value fp32 [nnz] | None for `pattern`, (M, N)), CPU tensors."""
    handle = open(path, "r") if not isinstance(path, io.IOBase) else path
    try:
        banner = handle.readline().split()
        if len(banner) < 5 or banner[0] != "%%MatrixMarket" or banner[1].lower() != "matrix":
            raise ValueError("mtx: missing '%%MatrixMarket matrix ...' banner")
        layout, field, symmetry = (t.lower() for t in banner[2:5])
        if layout != "coordinate":
            raise ValueError(f"mtx: only coordinate (sparse) files are graphs; got '{layout}'")
        if field not in _FIELDS:
            raise ValueError(f"mtx: field '{field}' not supported (real, integer, pattern)")
        if symmetry not in _SYMMETRIES:
            raise ValueError(f"mtx: symmetry '{symmetry}' not supported (general, symmetric, skew-symmetric)")
        line = handle.readline()
        while line and (line.startswith("%") or not line.strip()):
            line = handle.readline()
        try:
            m, n, count = (int(t) for t in line.split())
        except ValueError:
            raise ValueError(f"mtx: bad size line {line!r}") from None
        data = _body(handle, 2 if field == "pattern" else 3, count)
    finally:
        if handle is not path:
            handle.close()
    row = data[:, 0].astype(np.int64) - 1
    col = data[:, 1].astype(np.int64) - 1
    val = None if field == "pattern" else data[:, 2].astype(np.float32)
    if count and (row.min() < 0 or row.max() >= m or col.min() < 0 or col.max() >= n):
        raise ValueError("mtx: entry outside the declared shape")
    if symmetry != "general":
        off = row != col                         # the stored triangle plus its mirror image
        if symmetry == "skew-symmetric" and val is None:
            raise ValueError("mtx: a pattern file cannot be skew-symmetric")
        row, col = np.concatenate([row, col[off]]), np.concatenate([col, row[off]])
        if val is not None:
            val = np.concatenate([val, -val[off] if symmetry == "skew-symmetric" else val[off]])
    order = np.argsort(row * max(n, 1) + col, kind="stable")
    row, col = row[order], col[order]
    rowptr = np.zeros(m + 1, np.int64)
    np.cumsum(np.bincount(row, minlength=m), out=rowptr[1:])
    return (torch.from_numpy(rowptr), torch.from_numpy(col),
            None if val is None else torch.from_numpy(val[order]), (m, n))


def write_mtx(path, rowptr: torch.Tensor, col: torch.Tensor, value: Optional[torch.Tensor], sparse_sizes,
              comment: str = "") -> None:
    """`coordinate real general` (or `pattern` when value is None), one line per stored entry, CSR order."""
    rowptr_h = rowptr.detach().cpu().numpy().astype(np.int64)
    col_h = col.detach().cpu().numpy().astype(np.int64)
    row_h = np.repeat(np.arange(rowptr_h.size - 1, dtype=np.int64), np.diff(rowptr_h))
    m, n = int(sparse_sizes[0]), int(sparse_sizes[1])
    with open(path, "w") as out:
        out.write(f"%%MatrixMarket matrix coordinate {'pattern' if value is None else 'real'} general\n")
        for text in filter(None, comment.split("\n")):
            out.write(f"% {text}\n")
        out.write(f"{m} {n} {col_h.size}\n")
        if value is None:
            np.savetxt(out, np.stack([row_h + 1, col_h + 1], 1), fmt="%d %d")
        else:
            val_h = value.detach().cpu().numpy().astype(np.float64)
            np.savetxt(out, np.stack([row_h + 1, col_h + 1, val_h], 1), fmt="%d %d %.9g")
