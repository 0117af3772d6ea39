"""A minimal CSR container with exactly the members the reference wrapper touches.

The reference's ``spmm_autotuned`` (isplib/__init__.py:48-157) receives a
``torch_sparse.SparseTensor`` and uses only

    src.csr()                                   -> (rowptr, col, value|None)   :49
    src.storage._row/_rowcount/_csr2csc/_colptr    cached-only fields          :58-61
    src.storage.row()/rowcount()/csr2csc()/colptr()  lazy getters              :67,70-73

``torch_sparse`` is not part of this image, so the package ships this
duck-typed stand-in for users and tests; ``iSpLibPlugin`` accepts either.  The
lazy getters run on the DEVICE through the C ABI (``isplib_csr2csc_hip`` /
``isplib_csr_row_ids_hip``), replacing torch_sparse's host-side sort.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import cabi


class SparseStorage:
    """CSR storage with lazily built, cached transpose operands (torch_sparse names)."""

    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, value: Optional[torch.Tensor],
                 sparse_sizes: Tuple[int, int]):
        self._rowptr = rowptr
        self._col = col
        self._value = value
        self._sparse_sizes = (int(sparse_sizes[0]), int(sparse_sizes[1]))
        self._row: Optional[torch.Tensor] = None
        self._rowcount: Optional[torch.Tensor] = None
        self._colptr: Optional[torch.Tensor] = None
        self._csr2csc: Optional[torch.Tensor] = None
        # A^T operands kept beside the CSC fields (what isplib/__init__.py caches in
        # class-level dicts keyed by data pointers, :35-40,76-106)
        self._row_t: Optional[torch.Tensor] = None
        self._val_t: Optional[torch.Tensor] = None
        self._mean_val_t: Optional[torch.Tensor] = None
        # state of `value` (data pointer, in-place version counter) the two weight vectors above were built from:
        # an optimiser stepping trainable edge weights in place must not leave the backward on stale A^T weights
        self._val_t_state = None
        self._mean_val_t_state = None
        # task plans of A and A^T per slice count ([] when the rows are not column-sorted: plain kernel)
        self._plans = {}
        self._plans_t = {}
        # stream plans of A and A^T per (transposed, streams, slices, chunk[, "minmax"]); their weight vectors in stream
        # order per (transposed, streams, slices, chunk, kind) with the state of `value` they were gathered from
        self._streams = {}
        self._stream_vals = {}
        # (dense rows, k, minmax) -> schedule choice measured by iSpLibPlugin.autotune: ("stream", streams, slices, chunk) |
        # ("tasks", slices) | ("plain",); (dense rows, k, minmax, transposed) -> the slice rule's answer, remembered
        self._tuned = {}
        self._last_schedule = None   # what the last spmm_autotuned call on this graph ran its forward on

    def sparse_sizes(self) -> Tuple[int, int]:
        return self._sparse_sizes

    def rowptr(self) -> torch.Tensor:
        return self._rowptr

    def col(self) -> torch.Tensor:
        return self._col

    def value(self) -> Optional[torch.Tensor]:
        return self._value

    def row(self) -> torch.Tensor:
        if self._row is None:
            self._row = cabi.csr_row_ids(self._rowptr, self._col.numel())
        return self._row

    def rowcount(self) -> torch.Tensor:
        if self._rowcount is None:
            self._rowcount = self._rowptr[1:] - self._rowptr[:-1]
        return self._rowcount

    def _value_state(self):
        v = self._value
        return None if v is None else (v.data_ptr(), v._version)

    def _build_transpose(self) -> None:
        colptr, perm, row_t, val_t = cabi.csr2csc(self._rowptr, self._col, self._value, self._sparse_sizes[1],
                                                  want_val=self._value is not None)
        self._colptr, self._csr2csc, self._row_t, self._val_t = colptr, perm, row_t, val_t
        self._val_t_state = self._value_state()

    def colptr(self) -> torch.Tensor:
        if self._colptr is None:
            self._build_transpose()
        return self._colptr

    def csr2csc(self) -> torch.Tensor:
        if self._csr2csc is None:
            self._build_transpose()
        return self._csr2csc

    # -- A^T operands (not torch_sparse API; used by the plug-in's per-graph cache) --
    def row_t(self) -> torch.Tensor:
        if self._row_t is None:
            self._build_transpose()
        return self._row_t

    def val_t(self) -> Optional[torch.Tensor]:
        """value[csr2csc] (isplib/__init__.py:79), rebuilt when `value` was replaced or written in place since."""
        if self._value is None:
            return None
        if self._val_t is None:
            self._build_transpose()
        elif self._val_t_state != self._value_state():
            # the structure (colptr, csr2csc, row_t) is unchanged: only permute the fresh weights, into a NEW tensor
            # (a saved-for-backward reference to the old one stays what it was)
            self._val_t = self._value.detach()[self.csr2csc()].contiguous()
            self._val_t_state = self._value_state()
        return self._val_t

    def plan(self, n_slices: int):
        """Schedule operands of A for the `_planned` operators: [] (plain kernel) or the task plan
        [task_row, task_b, task_len, seg_off, lane_off_cpu, col32]; built once per graph and slice count
        (col32, the packed column ids, once per graph)."""
        return self._plan_for(self._plans, n_slices, self._rowptr, self._col, self._sparse_sizes[1])

    def row_order(self, transposed: bool, k: int):
        """[order] -- the plain kernel's rows in a community order (isplib_amd/reorder.py) -- for a square graph whose dense
        operand is larger than the Infinity Cache (n k 4 > 256 MiB) and HAS community structure, else [] (the plain
        kernel in index order).  Looked for once per side (~0.2 s for the ogbn-products shape); ISPLIB_REORDER=0 never
        looks.  The result is the same bits either way."""
        import os
        m, n = self._sparse_sizes
        if m != n or n * k * 4 <= (256 << 20) or os.environ.get("ISPLIB_REORDER", "1") == "0" or self._col.numel() == 0:
            return []
        cache = self.__dict__.setdefault("_row_orders", {})
        if transposed not in cache:
            from . import cabi, reorder
            rp, cl = (self.colptr(), self.row_t()) if transposed else (self._rowptr, self._col)
            try:
                cache[transposed] = reorder.useful_order(rp, cl)
            except (cabi.IsplibError, torch.OutOfMemoryError) as e:
                # the search is a speed-up only (~16 nnz + 16 m bytes of workspace, 8 radix sorts): when it cannot run, the
                # plain kernel in index order serves the call -- and the failure is remembered, not retried per call
                import warnings
                warnings.warn(f"isplib_amd: the row-order search was skipped ({type(e).__name__}: {e}); rows stay in index order")
                cache[transposed] = None
                torch.cuda.empty_cache()
        return [] if cache[transposed] is None else [cache[transposed]]

    def plan_t(self, n_slices: int):
        """The same for A^T (CSC operands)."""
        if n_slices <= 0:
            return []
        return self._plan_for(self._plans_t, n_slices, self.colptr(), self.row_t(), self._sparse_sizes[0])

    @staticmethod
    def _plan_for(cache, n_slices, rowptr, col, ncols):
        if n_slices <= 0:
            return []
        if n_slices not in cache:
            from .plan import build_task_plan
            shared = next((v[5] for v in cache.values() if len(v) == 6), None)     # one packed copy per graph
            try:
                p = build_task_plan(rowptr, col, ncols, n_slices, col32=shared)
            except RuntimeError:          # e.g. a graph too large for int32 task ids: plain kernel
                p = None
            cache[n_slices] = [] if p is None else [p.task_row, p.task_b, p.task_len, p.seg_off,
                                                    torch.tensor(p.lane_off, dtype=torch.int64), p.col32]
        return cache[n_slices]

    def stream_plan(self, transposed: bool, geom, kind: str = "sum"):
        """Operands of the stream schedule for the `_planned` operators: [words, vals, wave_step_off, wave_row, wave_part,
        hub_row, hub_off, meta_cpu] for A (or A^T), `geom` = (streams, slices, chunk) from ``cabi.suggest_stream``.
        `kind`: which weights ride in the plan -- "sum": value (A) / value[csr2csc] (A^T); "mean" (A^T only): the mean
        backward's value[csr2csc] / max(deg, 1); "minmax" (A only): value, on a plan of the max / min kernel's geometry
        with the permutation appended (geom = (streams, slices, chunk) from ``cabi.suggest_stream_minmax``).  The structure is built once per graph and geometry; the weights are
        re-gathered through the plan's permutation whenever `value` was replaced or written in place."""
        from .plan import build_stream_plan_native
        minmax = kind == "minmax"       # (A only) the max / min kernel's own geometry; None when rows are not column-sorted
        key = (bool(transposed),) + tuple(int(v) for v in geom) + (("minmax",) if minmax else ())
        plan = self._streams.get(key, False)
        if plan is False:
            # the library's own builder (rocPRIM sorts: ~10 ms on the Reddit shape; the torch construction of plan.py,
            # which produces the same arrays, takes 27-30 ms) -- the one-off cache build of isplib/__init__.py:76-106
            if transposed:
                plan = build_stream_plan_native(self.colptr(), self.row_t(), self._sparse_sizes[0], geom[1], geom[0], geom[2])
            else:
                plan = build_stream_plan_native(self._rowptr, self._col, self._sparse_sizes[1], geom[1], geom[0], geom[2], minmax=minmax)
            self._streams[key] = plan
            if plan is not None:
                plan.meta = torch.tensor([plan.rows, plan.cols, plan.slices, plan.gens, plan.waves_per_gen, plan.rows_per_wave,
                                          plan.streams, plan.n_steps, plan.n_parts, plan.n_hub, plan.chunk], dtype=torch.int64)
        if plan is None:
            return None
        vals = torch.empty(0, dtype=torch.float32, device=plan.words.device)
        if self._value is not None or (transposed and kind == "mean"):
            vkey = key + ("sum" if minmax else kind,)
            state = self._value_state()
            hit = self._stream_vals.get(vkey)
            if hit is None or hit[0] != state:
                if transposed:
                    src = self.mean_val_t() if kind == "mean" else self.val_t()
                else:
                    src = self._value
                ok = plan.perm >= 0
                vals = torch.zeros(plan.perm.numel(), dtype=torch.float32, device=plan.perm.device)
                vals[ok] = src.detach().to(torch.float32)[plan.perm[ok].to(torch.int64)]
                self._stream_vals[vkey] = (state, vals)
            else:
                vals = hit[1]
        ops = [plan.words, vals, plan.wave_step_off, plan.wave_row, plan.wave_part, plan.hub_row, plan.hub_off, plan.meta]
        if minmax:                      # the winners' word indices become CSR positions through the permutation
            if plan.perm.dtype != torch.int32:
                return None
            ops.append(plan.perm)
        return ops

    def gcn_dinv(self) -> torch.Tensor:
        """(deg + 1)^-1/2 per row: the D^-1/2 of GCN's normalisation with self loops, unit weights."""
        if getattr(self, "_gcn_dinv", None) is None:
            self._gcn_dinv = (self.rowcount() + 1).to(torch.float32).pow(-0.5)
        return self._gcn_dinv

    def mean_val_t(self) -> torch.Tensor:
        """value[csr2csc] / max(rowcount,1)[row[csr2csc]] (csrc/fusedmm.cpp:357-364)."""
        if self._mean_val_t is None or self._mean_val_t_state != self._value_state():
            _, _, _, self._mean_val_t = cabi.csr2csc(self._rowptr, self._col, self._value, self._sparse_sizes[1],
                                                     mean_scale=True, want_perm=False, want_row=False)
            self._mean_val_t_state = self._value_state()
        return self._mean_val_t


class SparseTensor:
    """CSR sparse matrix; constructor arguments follow torch_sparse.SparseTensor
    (README.md:105-110: row, col, value, sparse_sizes)."""

    def __init__(self, row: Optional[torch.Tensor] = None, rowptr: Optional[torch.Tensor] = None,
                 col: Optional[torch.Tensor] = None, value: Optional[torch.Tensor] = None,
                 sparse_sizes: Optional[Tuple[int, int]] = None, is_sorted: bool = False, validate: bool = True):
        if col is None:
            raise ValueError("SparseTensor: `col` is required")
        if rowptr is None:
            if row is None:
                raise ValueError("SparseTensor: one of `row` / `rowptr` is required")
            if sparse_sizes is None:
                m = int(row.max()) + 1 if row.numel() else 0
                n = int(col.max()) + 1 if col.numel() else 0
                sparse_sizes = (m, n)
            m, n = int(sparse_sizes[0]), int(sparse_sizes[1])
            if not is_sorted and row.numel():
                # torch_sparse order: by row, then column; equal keys keep input order
                perm = torch.sort(row * max(n, 1) + col, stable=True).indices
                row, col = row[perm], col[perm]
                value = value[perm] if value is not None else None
            counts = torch.bincount(row, minlength=m) if row.numel() else torch.zeros(m, dtype=torch.int64, device=col.device)
            rowptr = torch.zeros(m + 1, dtype=torch.int64, device=col.device)
            torch.cumsum(counts, 0, out=rowptr[1:])
        elif sparse_sizes is None:
            sparse_sizes = (rowptr.numel() - 1, int(col.max()) + 1 if col.numel() else 0)
        if validate:
            m, n = int(sparse_sizes[0]), int(sparse_sizes[1])
            if rowptr.numel() != m + 1:
                raise ValueError("SparseTensor: rowptr must have M+1 entries")
            if col.numel():
                lo, hi = int(col.min()), int(col.max())
                if lo < 0 or hi >= n:
                    raise ValueError(f"SparseTensor: column index out of range [0, {n})")
                if int(rowptr[0]) != 0 or int(rowptr[-1]) != col.numel() or bool((rowptr[1:] < rowptr[:-1]).any()):
                    raise ValueError("SparseTensor: rowptr is not a monotone prefix of nnz")
            if value is not None and value.numel() != col.numel():
                raise ValueError("SparseTensor: value and col differ in length")
        self.storage = SparseStorage(rowptr.to(torch.int64).contiguous(), col.to(torch.int64).contiguous(),
                                     value.contiguous() if value is not None else None, sparse_sizes)

    @classmethod
    def from_csr(cls, rowptr, col, value=None, sparse_sizes=None, validate=True) -> "SparseTensor":
        return cls(rowptr=rowptr, col=col, value=value, sparse_sizes=sparse_sizes, validate=validate)

    @classmethod
    def from_mtx(cls, path, device=None) -> "SparseTensor":
        """Graph from a MatrixMarket file, the input format of the reference's tuner (README.md:147-168)."""
        from .mtx import read_mtx
        rowptr, col, value, sizes = read_mtx(path)
        if device is not None:
            rowptr, col, value = rowptr.to(device), col.to(device), None if value is None else value.to(device)
        return cls(rowptr=rowptr, col=col, value=value, sparse_sizes=sizes, validate=False)

    def to_mtx(self, path, comment: str = "") -> None:
        from .mtx import write_mtx
        s = self.storage
        write_mtx(path, s._rowptr, s._col, s._value, s._sparse_sizes, comment)

    def csr(self):
        s = self.storage
        return s._rowptr, s._col, s._value

    def sparse_sizes(self) -> Tuple[int, int]:
        return self.storage.sparse_sizes()

    def size(self, dim: int) -> int:
        return self.storage.sparse_sizes()[dim]

    def nnz(self) -> int:
        return self.storage._col.numel()

    @property
    def device(self):
        return self.storage._col.device

    def to(self, device) -> "SparseTensor":
        s = self.storage
        return SparseTensor(rowptr=s._rowptr.to(device), col=s._col.to(device),
                            value=None if s._value is None else s._value.to(device),
                            sparse_sizes=s._sparse_sizes, validate=False)

    def cuda(self) -> "SparseTensor":
        return self.to("cuda")

    def t(self) -> "SparseTensor":
        """Transpose (CSR of A^T), built on the device."""
        s = self.storage
        colptr, row_t, val_t = s.colptr(), s.row_t(), s.val_t()
        return SparseTensor(rowptr=colptr, col=row_t, value=val_t,
                            sparse_sizes=(s._sparse_sizes[1], s._sparse_sizes[0]), validate=False)

    def matmul(self, other: torch.Tensor, reduce: str = "sum") -> torch.Tensor:
        from .plugin import matmul
        return matmul(self, other, reduce)

    __matmul__ = matmul
