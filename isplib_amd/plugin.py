"""iSpLibPlugin -- the reference's drop-in surface (isplib/__init__.py:34-210).

``iSpLibPlugin.patch_pyg()`` swaps ``torch_sparse.matmul`` and
``torch.sparse.mm`` for ``spmm_autotuned`` so that PyG's GCNConv / SAGEConv /
GINConv, which aggregate through ``torch_sparse.matmul(adj_t, x, reduce)``, run
on the HIP kernels unchanged; ``unpatch_pyg()`` restores LIFO;
``@isplib_autotune`` wraps a function in the pair.

Differences from the reference, each a defect there:
  * per-graph operands live on the graph's storage object (or in a table that
    checks the tensors are still the same live objects), not in class-level
    dicts keyed by raw data pointers that go stale (:35-40,50);
  * a ``mean`` call no longer poisons a later ``sum`` on the same graph (:85,141);
  * the mean-backward weights use the intended pairing
    value[csr2csc] / max(rowcount,1)[row[csr2csc]] (csrc/fusedmm.cpp:357-364), not
    the mixed CSR/CSC order of :86-91;
  * unit weights are never materialised (:51-57): ``value=None`` reaches the kernel;
  * max/min return the tensor, like ``torch_sparse.matmul`` does, not the
    ``(out, arg)`` tuple of :143,145 (the tuple stays available at the op level);
  * an unknown ``reduce`` raises ValueError instead of returning None (:154-155);
  * the patched ``torch.sparse.mm`` still serves ordinary torch sparse tensors.

Environment: ISPLIB_SLICES=<n> forces the task list with n column slices (0: plain kernel); ISPLIB_STREAM=0 keeps
sum / mean off the stream schedule; ISPLIB_STREAM_GEOM=streams:slices:chunk forces it with that plan geometry;
ISPLIB_TUNE_FILE names a tuning table to load at import; ISPLIB_DEBUG=1 prints per-operator device times.
"""
from __future__ import annotations

import json
import os
import weakref
from typing import Optional

import torch

from . import _lib
from .sparse import SparseStorage, SparseTensor

_lib.load_ops()

try:  # optional third-party modules the patch targets (absent in this image)
    import torch_sparse as _torch_sparse  # type: ignore
except Exception:  # pragma: no cover
    _torch_sparse = None
try:
    import torch_geometric.typing as _pyg_typing  # type: ignore
except Exception:  # pragma: no cover
    _pyg_typing = None


class _ForeignGraphs:
    """Prepared storage for SparseTensor objects that are not ours (torch_sparse).
    Keyed by data pointers like the reference (:50) but every hit is validated
    against weak references to the very tensors it was built from."""

    def __init__(self):
        self._table = {}

    def get(self, rowptr, col, value, sizes) -> SparseStorage:
        key = (rowptr.data_ptr(), col.data_ptr(), 0 if value is None else value.data_ptr(), col.numel())
        hit = self._table.get(key)
        if hit is not None:
            refs, versions, storage = hit
            same = refs[0]() is rowptr and refs[1]() is col and (value is None or refs[2]() is value)
            if same and versions == (rowptr._version, col._version, None if value is None else value._version):
                return storage
        self._table = {k: v for k, v in self._table.items() if all(r() is not None for r in v[0])}
        storage = SparseStorage(rowptr, col, value, sizes)
        refs = (weakref.ref(rowptr), weakref.ref(col), weakref.ref(value if value is not None else col))
        self._table[key] = (refs, (rowptr._version, col._version, None if value is None else value._version), storage)
        return storage

    def clear(self):
        self._table.clear()


_foreign = _ForeignGraphs()


def _storage_of(src, other: torch.Tensor) -> SparseStorage:
    st = getattr(src, "storage", None)
    if isinstance(st, SparseStorage):
        return st
    rowptr, col, value = src.csr()                                   # :49
    try:
        sizes = tuple(src.sparse_sizes())
    except Exception:
        sizes = (rowptr.numel() - 1, other.size(-2))
    return _foreign.get(rowptr, col, value, sizes)


def suggest_slices(m: int, n: int, nnz: int, k: int, minmax: bool = False) -> int:
    """Column-slice count for an M x N, nnz-entry SpMM over K fp32 features (0 = plain kernel).

    Measured on MI355X (scripts/exp_small.py; DESIGN.md section 5, task-list schedule): (i) a slice of the
    dense operand should be about 7 MB (2x an XCD's 4 MiB L2: the hot rows stay resident); (ii) below that,
    a few slices still pay because they cut rows into more, shorter tasks -- K/20 of them (6 at K=128, 2 at
    K=32) -- as long as the graph has work for the whole chip; (iii) a row must keep ~20 edges per slice or
    per-task overhead and the partial rows eat the gain.  Reddit-shaped graph (mean degree 492): K=32 -> 4,
    K=64 -> 8, K >= 96 with rows of whole cache lines (64-column panels) -> 8, ragged K=100 -> 13, K=602 -> 16;
    a tenth of the graph at K=128 -> 6;
    ogbn-products-shaped (mean degree 50) or under a million edges -> 0 (plain kernel, no preparation)."""
    from . import cabi
    return int(cabi.lib().isplib_suggest_slices(int(m), int(n), int(nnz), int(k), int(bool(minmax))))   # one rule, in the C ABI


# Measured choices that outlive the process: {graph signature: {"rows:k:minmax": ["stream", streams, slices, chunk] |
# ["tasks", slices] | ["plain"]}} (tables from before round 5: a bare slice count).  Filled by
# iSpLibPlugin.autotune, written/merged by save_tuning/load_tuning; ISPLIB_TUNE_FILE names a file read at import.
_tuning_db: dict = {}


def _as_choice(entry):
    """A tuning-table entry as a schedule choice: ("plain",) | ("tasks", slices) | ("stream", streams, slices, chunk).
    Tables written before round 5 hold bare slice counts (0 = plain kernel)."""
    if isinstance(entry, (int, float)):
        return ("tasks", int(entry)) if int(entry) > 0 else ("plain",)
    entry = tuple(entry)
    return (str(entry[0]),) + tuple(int(v) for v in entry[1:])


def tuned_choice(storage: SparseStorage, rows: int, k: int, minmax: bool = False):
    """What `iSpLibPlugin.autotune` measured best for an SpMM of this graph whose dense operand has `rows` rows and k
    columns (this process, or a loaded tuning table keyed by the graph's signature); None = nothing measured: the rules."""
    key = (int(rows), int(k), bool(minmax))
    hit = storage._tuned.get(key)
    if hit is None and _tuning_db:
        entry = _tuning_db.get(graph_signature(storage), {}).get(f"{key[0]}:{key[1]}:{int(key[2])}")
        if entry is not None:
            hit = storage._tuned[key] = _as_choice(entry)
    return hit


def graph_signature(storage: SparseStorage) -> str:
    """Content key of a graph for the persisted tuning table: shape, nnz and the histogram of floor(log2(degree)).
    Two graphs with the same key get the same schedule, which is all the key is for (one host sync, cached)."""
    sig = getattr(storage, "_signature", None)
    if sig is None:
        deg = storage.rowcount()
        hist = torch.bincount(torch.log2(deg.clamp(min=1).to(torch.float32)).to(torch.int64)).tolist() if deg.numel() else []
        m, n = storage.sparse_sizes()
        sig = storage._signature = f"{m}x{n}:{storage._col.numel()}:" + ",".join(str(c) for c in hist)
    return sig


def degree_cv2(rowptr: torch.Tensor) -> float:
    """Squared coefficient of variation of the row degrees (one host sync)."""
    deg = (rowptr[1:] - rowptr[:-1]).to(torch.float64)
    if deg.numel() == 0 or float(deg.sum()) == 0.0:
        return 0.0
    mean = deg.mean()
    return max(0.0, float((deg * deg).mean() / (mean * mean)) - 1.0)


def skew_adjusted(storage_or_rowptr, slices: int, cap: int = 64) -> int:
    """The rule's 7 MB per slice leans on the popularity skew of real degree distributions (hot rows of the dense operand
    stay in the L2).  A graph whose degrees hardly vary (CV^2 < 0.25) has no hot rows and wants slices closer to the L2
    size: half as many columns per slice again (uniform random graph of the Reddit size, K=128: 4.09 -> 3.49 ms on the
    task list; on the stream schedule 31 -> 47 slices: 3.21 -> 3.00 ms).  The same test lives in the C handle
    (graph_runtime.hip)."""
    if slices <= 0:
        return slices
    if isinstance(storage_or_rowptr, torch.Tensor):
        cv2 = degree_cv2(storage_or_rowptr)
    else:
        cv2 = getattr(storage_or_rowptr, "_cv2", None)
        if cv2 is None:
            cv2 = storage_or_rowptr._cv2 = degree_cv2(storage_or_rowptr._rowptr)
    return slices if cv2 >= 0.25 else min(cap, int(1.5 * slices + 0.5))


def choose_slices(storage: SparseStorage, rows: int, k: int, minmax: bool = False, transposed: bool = False) -> int:
    """Slice count for a graph held in `storage` (transposed: for its A^T, the backward's operand; `rows` = rows of the
    dense operand either way): ISPLIB_SLICES=<n> (0 disables) > a count measured by `iSpLibPlugin.autotune` for this
    graph and width (this process, or a loaded tuning file) > the `suggest_slices` rule."""
    env = os.environ.get("ISPLIB_SLICES")
    if env is not None:
        n = int(env)
        return n if 1 <= n <= 4096 else 0
    tuned = tuned_choice(storage, rows, k, minmax)
    if tuned is not None and tuned[0] == "tasks":
        return tuned[1]
    if tuned is not None and tuned[0] == "plain":
        return 0
    ruled = storage._tuned.get((rows, k, minmax, transposed))          # the rule's answer for this side, from last time
    if ruled is not None:
        return ruled
    m, nnz = storage.sparse_sizes()[1 if transposed else 0], storage._col.numel()
    s = skew_adjusted(storage, suggest_slices(m, rows, nnz, k, minmax))
    if s > 0:
        # the panel rule halves the slice count to make tasks long enough; on hub-dominated graphs they are long anyway
        # (>= 120 edges per task on the panel plan) and the whole-row plan, run in one pass, is the better schedule
        from . import cabi
        whole = int(cabi.lib().isplib_suggest_slices_whole_rows(m, rows, nnz, k))
        if whole > s:
            plan_p = storage.plan_t(s) if transposed else storage.plan(s)
            if plan_p and nnz / max(plan_p[0].numel(), 1) >= 120.0:
                s = whole
        storage._tuned[(rows, k, minmax, transposed)] = s
    return s


def choose_stream(storage: SparseStorage, m: int, n: int, k: int, weighted: bool = False):
    """(streams, slices, chunk) when sum / mean of this shape should run on the stream schedule, else None; `weighted`:
    the plan will carry edge weights (the rule then prefers wider column panels: the weight stream is read once per panel).
    ISPLIB_STREAM=0 disables it, ISPLIB_SLICES (the task list's override) disables it too: explicit schedules win."""
    if os.environ.get("ISPLIB_STREAM", "1") == "0" or os.environ.get("ISPLIB_SLICES") is not None:
        return None
    if k < 4 or n >= (1 << 24):
        return None
    forced = os.environ.get("ISPLIB_STREAM_GEOM")          # "streams:slices:chunk": tests and experiments
    if forced:
        return tuple(int(v) for v in forced.split(":"))
    tuned = tuned_choice(storage, n, k, False)             # a measured choice for this graph and width beats the rule
    if tuned is not None:
        return tuple(tuned[1:]) if tuned[0] == "stream" else None
    return stream_rule(storage, m, n, k, weighted)


def stream_rule(storage: SparseStorage, m: int, n: int, k: int, weighted: bool = False):
    """The rule's own answer (isplib_suggest_stream_weighted + the degree-skew adjustment), whatever was tuned."""
    from . import cabi
    geom = cabi.suggest_stream(m, n, storage._col.numel(), k, weighted)
    return None if geom is None else (geom[0], skew_adjusted(storage, geom[1], cap=512), geom[2])


def choose_stream_minmax(storage: SparseStorage, m: int, n: int, k: int):
    """(streams, slices, chunk) when max / min of this shape should run on the stream schedule (column-sorted rows; the
    plan builder has the last word), else None.  Same switches as choose_stream."""
    if os.environ.get("ISPLIB_STREAM", "1") == "0" or os.environ.get("ISPLIB_SLICES") is not None:
        return None
    if k < 4 or n >= (1 << 24) or storage._col.numel() >= (1 << 31):
        return None
    forced = os.environ.get("ISPLIB_STREAM_MINMAX_GEOM")   # "streams:slices:chunk": tests and experiments
    if forced:
        return tuple(int(v) for v in forced.split(":"))
    tuned = tuned_choice(storage, n, k, True)
    if tuned is not None:
        return tuple(tuned[1:]) if tuned[0] == "stream" else None
    return stream_minmax_rule(storage, m, n, k)


def stream_minmax_rule(storage: SparseStorage, m: int, n: int, k: int):
    """The rule's own answer for max / min (isplib_suggest_stream_minmax + the degree-skew adjustment)."""
    from . import cabi
    geom = cabi.suggest_stream_minmax(m, n, storage._col.numel(), k)
    return None if geom is None else (geom[0], skew_adjusted(storage, geom[1], cap=512), geom[2])


def spmm_autotuned(src, other: torch.Tensor, reduce: str = "sum") -> torch.Tensor:
    """``torch_sparse.matmul(src, other, reduce)`` on the HIP path (isplib/__init__.py:48-157)."""
    if reduce not in ("sum", "add", "mean", "max", "min"):
        raise ValueError(f"isplib: unknown reduce '{reduce}' (expected sum|add|mean|max|min)")
    if not isinstance(other, torch.Tensor) or not other.is_cuda:
        raise RuntimeError("isplib_amd: `other` must be a GPU tensor -- there is no CPU path")
    if other.dtype != torch.float32:
        raise TypeError(f"isplib_amd: only float32 features are supported (csrc/fusedmm.cpp:44), got {other.dtype}")
    s = _storage_of(src, other)
    rowptr, col, value = s._rowptr, s._col, s._value
    if value is not None and value.dtype != other.dtype:
        value = value.to(other.dtype)                                # :63-64
    squeeze = other.dim() == 1
    mat = other.unsqueeze(-1) if squeeze else other
    needs_grad = torch.is_grad_enabled() and mat.requires_grad       # :69-73
    ops = torch.ops.isplib
    k = mat.size(-1)
    m_rows = rowptr.numel() - 1
    # sum / mean on graphs with work for the whole chip: the stream schedule (rows resident in LDS, the plan's own copy
    # of the edges; include/isplib_hip.h: fusedMM_csr_stream_hip) -- measured 15-18 % ahead of the task list
    if reduce in ("sum", "add", "mean"):
        geom = choose_stream(s, m_rows, mat.size(0), k, s._value is not None)
        plan = s.stream_plan(False, geom) if geom is not None else None
    else:                                                            # max / min: its own kernel geometry, sorted rows only
        geom = choose_stream_minmax(s, m_rows, mat.size(0), k)
        plan = s.stream_plan(False, geom, "minmax") if geom is not None else None
    ran = ("stream",) + tuple(int(v) for v in geom) if plan is not None else None
    if plan is None:
        n_sl = choose_slices(s, mat.size(0), k, reduce in ("max", "min"))
        plan = s.plan(n_sl)                                          # per-graph, built once on the device
        ran = ("tasks", int(n_sl)) if plan else ("plain",)
        if not plan:                                                 # the plain kernel: rows in a community order where that pays
            plan = s.row_order(False, k)
    s._last_schedule = ran                                           # what this call's forward runs on (autotune checks it)
    if reduce in ("sum", "add", "mean"):
        colptr = val_t = row_t = None
        plan_t = []
        if needs_grad:                                               # :76-80 / :83-99 (built once per graph)
            colptr, row_t = s.colptr(), s.row_t()
            # mean: the intended pairing, val[csr2csc] / max(deg, 1) -- for an unweighted graph that is
            # 1 / deg of the edge's row in A, which the operator applies to the rows of dY instead: no weights at all
            unit_mean = reduce == "mean" and s._value is None
            val_t = None if unit_mean else (s.mean_val_t() if reduce == "mean" else s.val_t())
            geom_t = choose_stream(s, mat.size(0), m_rows, k, val_t is not None)
            plan_t = s.stream_plan(True, geom_t, "mean" if reduce == "mean" and not unit_mean else "sum") if geom_t is not None else None
            if plan_t is None:
                plan_t = s.plan_t(choose_slices(s, m_rows, k, transposed=True))
                if not plan_t:
                    plan_t = s.row_order(True, k)
        if reduce == "mean":
            out = ops.fusedmm_spmm_mean_planned(rowptr, col, value, colptr, mat, row_t, val_t, plan, plan_t)
        else:
            out = ops.fusedmm_spmm_planned(rowptr, col, value, colptr, mat, val_t, row_t, plan, plan_t)
    elif not needs_grad and not (value is not None and torch.is_grad_enabled() and value.requires_grad):
        # max / min with nothing to differentiate (inference): the patched matmul returns the tensor alone (:143,145), so the
        # winners' positions would be computed and dropped -- the values-only launch leaves them out (stream plans: 12 % less)
        out = (ops.fusedmm_spmm_max_values if reduce == "max" else ops.fusedmm_spmm_min_values)(rowptr, col, value, mat, plan)
    elif reduce == "max":
        out = ops.fusedmm_spmm_max_planned(rowptr, col, value, mat, plan)[0]   # :143
    else:
        out = ops.fusedmm_spmm_min_planned(rowptr, col, value, mat, plan)[0]   # :145
    return out.squeeze(-1) if squeeze else out


matmul = spmm_autotuned


def fusedmm(src, x: Optional[torch.Tensor], y: torch.Tensor, pattern="sigmoid_embedding", sop_param: float = 0.0,
            sop_udef: Optional[str] = None) -> torch.Tensor:
    """The generic FusedMM pipeline over the stored entries of `src` (include/isplib_hip.h, fusedMM_csr_udef_hip):
    ``z[i] = AOP_j VSC(SOP(ROP(VOP(x[i], y[j]))))``.  `pattern` is a name from ``cabi.PATTERNS`` (sigmoid_embedding,
    tdist_embedding, attention_sum, spmm) or a raw message word built from ``cabi.VOP/ROP/SOP/VSC/AOP``; for a raw
    word with SOP_UDEF, `sop_udef` names the built-in function.  Forward only (the reference has no such op:
    these words are defined by csrc/fusedMM.h:18-74 but never sent by iSpLib)."""
    from . import cabi
    if isinstance(pattern, str):
        word, fn = cabi.PATTERNS[pattern]
    else:
        word, fn = int(pattern), (sop_udef or "none")
    st = _storage_of(src, y)
    k = y.size(1)
    m_rows = st._rowptr.numel() - 1
    # the two hot SDDMM-fused words on graphs with work for the whole chip: the stream front end (rows of x and z resident in
    # LDS; include/isplib_hip.h: fusedMM_csr_udef_stream_hip) -- Reddit shape K=128: 6.3 ms on the task list below
    geom = None
    if x is not None and x.is_contiguous() and y.is_contiguous() and os.environ.get("ISPLIB_STREAM", "1") != "0" and os.environ.get("ISPLIB_SLICES") is None:
        geom = cabi.suggest_fusedmm_stream(word, m_rows, y.size(0), st._col.numel(), k)
    if geom is not None:
        plans = st.__dict__.setdefault("_fusedmm_streams", {})
        if geom not in plans:
            try:
                plans[geom] = cabi.NativeStreamPlan(st._rowptr, st._col, None, y.size(0), geom[0], geom[1], geom[2], 0, fusedmm=True)
            except cabi.IsplibError:
                plans[geom] = None                        # outside the builder's domain / no room: the task list serves the call
        if plans[geom] is not None:
            kind = sop_udef or fn
            return cabi.fusedmm_stream(word, st._rowptr, st._col.numel(), plans[geom], x, y, sop_udef=kind, sop_param=sop_param)[1]
    # the reduce stage needs whole rows of y, so there are no column panels here: slices for the full width
    plan = None
    nbytes = y.size(0) * k * 4
    slices = int(cabi.lib().isplib_suggest_slices_whole_rows(st._rowptr.numel() - 1, y.size(0), st._col.numel(), k))
    if 4 <= k <= 1024 and slices > 0 and nbytes >= (14 << 20):
        held = st.plan(slices)
        if held:
            from .plan import TaskPlan
            plan = TaskPlan(slices, held[0].numel(), held[0], held[1], held[2], held[3], [int(v) for v in held[4]], 1024, 128, held[5])
    return cabi.fusedmm(word, st._rowptr, st._col, st._value, x, y, sop_udef=sop_udef or fn, sop_param=sop_param, plan=plan)[1]


def gcn_norm_matmul(src, other: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    """relu(D^-1/2 (A + I) D^-1/2 @ other + bias) for an UNWEIGHTED square graph, without materialising the
    normalised edge weights: the SpMM runs on the unit-weight fast path, the self loop, the left D^-1/2, bias
    and ReLU are applied inside the fold kernel (include/isplib_hip.h: isplib_epilogue).  This is what a
    GCNConv(normalize=True) layer aggregates (callers: tests/dist/gcn/pyg-sparse.py:61-62); differentiable in
    `other` and `bias`.  Not part of the reference's surface."""
    if not other.is_cuda or other.dtype != torch.float32 or other.dim() != 2:
        raise RuntimeError("isplib_amd: gcn_norm_matmul needs a float32 GPU matrix [N, K] -- there is no CPU path")
    s = _storage_of(src, other)
    if s._value is not None:
        raise ValueError("gcn_norm_matmul is defined for unweighted graphs (value=None)")
    if s._sparse_sizes[0] != s._sparse_sizes[1]:
        raise ValueError("gcn_norm_matmul needs a square adjacency")
    k = other.size(1)
    n = other.size(0)
    needs_grad = torch.is_grad_enabled() and other.requires_grad
    # the same schedule rule as matmul: stream plans (unit weights; the kernel applies the epilogue when it writes a
    # finished row) where isplib_suggest_stream accepts the shape, else the task list
    geom = choose_stream(s, n, n, k)
    plan = s.stream_plan(False, geom) if geom is not None else None
    n_sl = None
    if plan is None:
        n_sl = choose_slices(s, n, k)
        plan = s.plan(n_sl)
    colptr = row_t = None
    plan_t = []
    if needs_grad:
        colptr, row_t = s.colptr(), s.row_t()
        plan_t = s.stream_plan(True, geom, "sum") if geom is not None else None
        if plan_t is None:
            plan_t = s.plan_t(choose_slices(s, n, k, transposed=True) if n_sl is None else n_sl)
    return torch.ops.isplib.gcn_norm_spmm(s._rowptr, s._col, other, s.gcn_dinv(), colptr, row_t, plan, plan_t, bias, relu)


class iSpLibPlugin:
    backup = []          # LIFO of (torch_sparse.matmul | None, torch.sparse.mm, WITH_PT2, WITH_PT20)

    @classmethod
    def patch_pyg(cls) -> None:
        saved_flags = (None, None)
        if _pyg_typing is not None:                                  # :159-171
            saved_flags = (getattr(_pyg_typing, "WITH_PT2", None), getattr(_pyg_typing, "WITH_PT20", None))
            _pyg_typing.WITH_PT2 = False
            _pyg_typing.WITH_PT20 = False
        original_mm = torch.sparse.mm

        def sparse_mm(src, other, reduce: str = "sum"):
            if hasattr(src, "csr") and hasattr(src, "storage"):
                return spmm_autotuned(src, other, reduce)
            return original_mm(src, other) if reduce == "sum" else original_mm(src, other, reduce)

        cls.backup.append((None if _torch_sparse is None else _torch_sparse.matmul, original_mm) + saved_flags)  # :173-174
        if _torch_sparse is not None:
            _torch_sparse.matmul = spmm_autotuned                    # :177
        torch.sparse.mm = sparse_mm                                  # :178

    @classmethod
    def unpatch_pyg(cls) -> None:
        if not cls.backup:                                           # :190
            return
        ts_matmul, sparse_mm, pt2, pt20 = cls.backup.pop()
        torch.sparse.mm = sparse_mm                                  # :194
        if _torch_sparse is not None and ts_matmul is not None:
            _torch_sparse.matmul = ts_matmul                         # :195
        if _pyg_typing is not None:                                  # :197-201
            if pt2 is not None:
                _pyg_typing.WITH_PT2 = pt2
            if pt20 is not None:
                _pyg_typing.WITH_PT20 = pt20

    @classmethod
    def autotune(cls, src, k: int, reduce: str = "sum", candidates=(0, 2, 4, 6, 8, 12, 16, 24), reps: int = 3, other_rows=None,
                 stream_slices=(0.5, 0.75, 1.0, 1.25, 1.5), stream_chunk=(0.5, 2.0)):
        """Times the SpMM of `src` at width `k` on every candidate SCHEDULE AND GEOMETRY on the actual graph and keeps the
        fastest for every later call (the heir of the reference's tuning scripts: autotuner/findbestk.py:34-38 sweeps K and
        prints a table, gpu/kernels/codegen.py:30-41 sweeps a launch parameter).  Candidates, each actually forced:
          ("stream", streams, slices, chunk)  the stream schedule where its rule accepts the shape: the rule's column-slice
                                              count x `stream_slices`, and the rule's hub-row chunk x `stream_chunk`
          ("tasks", slices)                   the task list at every non-zero slice count of `candidates`
          ("plain",)                          the row-per-wave kernel (a 0 in `candidates`)
        A candidate only counts if the call really ran on it (a plan builder that declines falls back to another
        schedule: that measurement is dropped).  Returns {choice: milliseconds}; the winner lives on the graph's storage
        and in the table `save_tuning` writes.  Costs one plan build per candidate; the losers' stream plans are freed."""
        rowptr, col, value = src.csr()
        x = torch.zeros((other_rows or src.sparse_sizes()[1], k), dtype=torch.float32, device=col.device)
        s = _storage_of(src, x)
        minmax = reduce in ("max", "min")
        key = (x.size(0), k, minmax)
        m_rows = rowptr.numel() - 1
        before = s._tuned.pop(key, None)
        rule = stream_minmax_rule(s, m_rows, x.size(0), k) if minmax else stream_rule(s, m_rows, x.size(0), k, s._value is not None)
        stream_off = os.environ.get("ISPLIB_STREAM", "1") == "0" or os.environ.get("ISPLIB_SLICES") is not None
        cands = []
        if rule is not None and not stream_off and k >= 4 and x.size(0) < (1 << 24):
            st, sl, ch = rule
            for f in stream_slices:
                cands.append(("stream", st, max(1, min(512, int(sl * f + 0.5))), ch))
            for f in stream_chunk:
                cands.append(("stream", st, sl, max(256, int(ch * f))))
        for c in candidates:
            cands.append(("tasks", int(c)) if int(c) > 0 else ("plain",))
        cands = list(dict.fromkeys(cands))                         # the rule's own geometry appears once
        times = {}
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        built = set(s._streams)
        for cand in cands:
            s._tuned[key] = cand
            try:
                spmm_autotuned(src, x, reduce)                     # builds the plan, warms up
                if s._last_schedule != cand:                       # the builder declined: this is another schedule's time
                    continue
                start.record()
                for _ in range(reps):
                    spmm_autotuned(src, x, reduce)
                stop.record()
                torch.cuda.synchronize()
                times[cand] = start.elapsed_time(stop) / reps
            except RuntimeError:
                continue
        if times:
            best = s._tuned[key] = min(times, key=times.get)
            _tuning_db.setdefault(graph_signature(s), {})[f"{key[0]}:{k}:{int(minmax)}"] = list(best)
        elif before is not None:
            s._tuned[key] = before
        else:
            s._tuned.pop(key, None)
        keep = s._tuned.get(key)
        for pkey in [p_ for p_ in s._streams if p_ not in built]:      # stream plans of the losing geometries: 4-12 B per edge each
            if not (keep is not None and keep[0] == "stream" and pkey[1:4] == tuple(keep[1:])):
                s._streams.pop(pkey, None)
                for vkey in [v_ for v_ in s._stream_vals if v_[:len(pkey)] == pkey]:
                    s._stream_vals.pop(vkey, None)
        return times

    @classmethod
    def save_tuning(cls, path) -> None:
        """Writes every choice `autotune` has measured (merged over what the file already holds) as JSON."""
        merged = {}
        if os.path.exists(path):
            with open(path) as f:
                merged = json.load(f)
        for sig, table in _tuning_db.items():
            merged.setdefault(sig, {}).update(table)
        with open(path, "w") as f:
            json.dump(merged, f, indent=1, sort_keys=True)

    @classmethod
    def load_tuning(cls, path) -> int:
        """Merges a file written by `save_tuning`; returns the number of graphs it describes."""
        with open(path) as f:
            table = json.load(f)
        for sig, entries in table.items():
            _tuning_db.setdefault(sig, {}).update({k: list(_as_choice(v)) for k, v in entries.items()})
        return len(table)

    @classmethod
    def is_patched(cls) -> bool:
        return bool(cls.backup)

    @classmethod
    def clear_cache(cls) -> None:
        _foreign.clear()
        _tuning_db.clear()


def isplib_autotune(fn):
    """Decorator: run ``fn`` with the patch applied (isplib/__init__.py:204-210);
    unlike the reference, the patch is removed even when ``fn`` raises."""

    def wrapper(*args, **kwargs):
        iSpLibPlugin.patch_pyg()
        try:
            return fn(*args, **kwargs)
        finally:
            iSpLibPlugin.unpatch_pyg()

    wrapper.__name__ = getattr(fn, "__name__", "wrapper")
    wrapper.__doc__ = getattr(fn, "__doc__", None)
    return wrapper


if os.environ.get("ISPLIB_TUNE_FILE") and os.path.exists(os.environ["ISPLIB_TUNE_FILE"]):
    iSpLibPlugin.load_tuning(os.environ["ISPLIB_TUNE_FILE"])
