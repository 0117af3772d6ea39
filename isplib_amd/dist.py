"""1-D row partition of the adjacency over the GPUs of one node (one process per
GPU, ``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm).

The reference has no distributed path at all; this is new
design, following BASELINE.json's north_star: rank p owns a contiguous range
of rows of A chosen so that nnz is balanced, plus the matching rows of X.  One
SpMM = ONE all-gather of X (every rank contributes its shard) followed by the
local ``out[R_p] = A[R_p, :] @ X``.  Each output row is still produced by one
wave in the same edge order, so the result is bit-identical to the single-GPU
one and no reduce-scatter / atomics are needed.  The backward is the same
thing on A^T (csrc/fusedmm.cpp:285): all-gather dY, local ``A^T[R_p, :] @ dY``.

Shards have unequal row counts (nnz-balanced); the gather buffer is laid out
[world, max_rows, K] and the local column ids are remapped ONCE, at partition
time, into that padded layout, so the SpMM reads the gathered buffer in place
(no compaction copy after the collective).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


def nnz_balanced_cuts(rowptr: torch.Tensor, world: int) -> List[int]:
    """world+1 row boundaries with ~nnz/world entries per part."""
    nnz = int(rowptr[-1])
    m = rowptr.numel() - 1
    if world == 1:
        return [0, m]
    targets = torch.arange(1, world, device=rowptr.device, dtype=torch.int64) * (nnz // world)
    cuts = torch.searchsorted(rowptr, targets).clamp_(max=m).tolist()
    out = [0] + cuts + [m]
    for i in range(1, len(out)):          # keep boundaries monotone on degenerate inputs
        out[i] = max(out[i], out[i - 1])
    return out


def _p2p_is_stream_ordered(group, device) -> bool:
    """True when the group's point-to-point calls on `device` order themselves behind the current stream (RCCL)."""
    if device.type != "cuda":
        return True                                   # host tensors: nothing is queued anywhere
    try:
        name = str(dist.get_backend(group)).lower()
    except Exception:                                 # noqa: BLE001 - an unknown backend is treated as unordered
        return False
    if "cuda:" in name:                               # "cpu:gloo,cuda:nccl"
        name = name.split("cuda:", 1)[1].split(",", 1)[0]
    return "nccl" in name or "rccl" in name


class RowPartition:
    """This rank's slice of a CSR matrix whose columns index the row-sharded X."""

    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, val: Optional[torch.Tensor], ncols: int,
                 rank: int, world: int, cuts: Optional[List[int]] = None, group=None):
        m = rowptr.numel() - 1
        self.rank, self.world, self.group = rank, world, group
        self.row_cuts = cuts if cuts is not None else nnz_balanced_cuts(rowptr, world)
        # X (ncols rows) is sharded with the same boundaries when A is square, evenly otherwise
        if ncols == m:
            self.x_cuts = list(self.row_cuts)
        else:
            self.x_cuts = [min(ncols, (ncols * p + world - 1) // world) for p in range(world)] + [ncols]
        r0, r1 = self.row_cuts[rank], self.row_cuts[rank + 1]
        self.row0, self.rows = r0, r1 - r0
        lo, hi = int(rowptr[r0]), int(rowptr[r1])
        self.edge0, self.nnz, self.total_nnz = lo, hi - lo, int(rowptr[-1])
        self.rowptr = (rowptr[r0:r1 + 1] - lo).contiguous()
        self.col = col[lo:hi].contiguous()
        self.val = None if val is None else val[lo:hi].contiguous()
        sizes = [self.x_cuts[p + 1] - self.x_cuts[p] for p in range(world)]
        # shard pitch of the gather buffer; a multiple of 192 so that a shard can be cut into q = 1, 2, 3, 4,
        # 6, 8, 12, 16, ... whole column slices (spmm_overlapped runs world * q slices)
        self.max_rows = (max(max(sizes), 1) + 191) // 192 * 192
        self.x_rows = sizes[rank]
        xc = torch.tensor(self.x_cuts, dtype=torch.int64, device=col.device)
        owner = torch.searchsorted(xc[1:].contiguous(), self.col, right=True).clamp_(max=world - 1)
        self.col_padded = (owner * self.max_rows + (self.col - xc[owner])).contiguous()
        self.ncols_padded = world * self.max_rows
        self._parked = None          # first local kernel error of the exchange in progress (_kernel / _raise_parked)
        self.fail_next_kernel = None  # test hook: an exception the next _kernel call fails with

    # ---- local failures must not break the collective sequence ------------------------------------------------------
    def _kernel(self, fn, *args, **kw):
        """One local kernel call inside an exchange schedule.  A failure on THIS rank (an argument check of the C ABI, an
        allocation) must not keep it from posting the sends / receives / waits its peers are going to block in: the first
        error is parked, the schedule goes on with its communication (the remaining local kernels are skipped), and
        `_raise_parked` raises it once the exchange is complete -- every rank has then issued the same collectives, so the
        caller can tell its peers and all of them can go on (bench.py drops the candidate on every rank)."""
        if self._parked is not None:
            return
        try:
            if self.fail_next_kernel is not None:
                e, self.fail_next_kernel = self.fail_next_kernel, None
                raise e
            fn(*args, **kw)
        except Exception as e:  # noqa: BLE001 - re-raised by _raise_parked
            self._parked = e

    def _raise_parked(self):
        e, self._parked = self._parked, None
        if e is not None:
            e.collectives_complete = True      # for the caller: this rank is still in step with its peers
            raise e

    def shard(self, x_full: torch.Tensor) -> torch.Tensor:
        """This rank's rows of a replicated X, zero-padded to max_rows."""
        k = x_full.size(1)
        out = torch.zeros((self.max_rows, k), dtype=x_full.dtype, device=x_full.device)
        out[: self.x_rows] = x_full[self.x_cuts[self.rank]: self.x_cuts[self.rank + 1]]
        return out

    def gather_buffer(self, k: int, device=None, dtype=torch.float32) -> torch.Tensor:
        return torch.empty((self.ncols_padded, k), dtype=dtype, device=device or self.col.device)

    def all_gather(self, x_shard: torch.Tensor, buf: torch.Tensor):
        """ONE collective per SpMM.  x_shard: [max_rows, K] contiguous."""
        if self.world == 1:
            buf.copy_(x_shard)
            return None
        return dist.all_gather_into_tensor(buf, x_shard, group=self.group)

    def unpad(self, buf: torch.Tensor) -> torch.Tensor:
        """Gathered padded buffer -> replicated [ncols, K] (tests / debugging only)."""
        parts = [buf[p * self.max_rows: p * self.max_rows + (self.x_cuts[p + 1] - self.x_cuts[p])]
                 for p in range(self.world)]
        return torch.cat(parts, 0)

    def spmm(self, x_shard: torch.Tensor, reduce: str = "sum", buf: Optional[torch.Tensor] = None):
        """all-gather(X) + local SpMM on the HIP path; returns (out[rows, K], arg|None).
        arg holds GLOBAL CSR positions (local position + this rank's edge offset;
        the 'no winner' sentinel becomes the global nnz)."""
        from . import cabi
        k = x_shard.size(1)
        buf = self.gather_buffer(k, x_shard.device) if buf is None else buf
        self.all_gather(x_shard, buf)
        out = torch.empty((self.rows, k), dtype=torch.float32, device=x_shard.device)
        arg = torch.empty((self.rows, k), dtype=torch.int64, device=x_shard.device) if reduce in ("max", "min") else None
        cabi.fusedMM_csr_hip(cabi.MESSAGE[reduce], self.rowptr, self.col_padded, self.val, buf, out, arg)
        if arg is not None:
            arg = self.global_arg(arg)
        return out, arg

    # ---- overlapped form: local column slices run while the all-gather is in flight ----------------

    def plan(self, k: int, reduce: str = "sum", slices: Optional[int] = None):
        """Per-graph operands of the sliced path on this rank: (slices, table, workspace) or None.
        The slice count is a multiple of `world`, so each shard of X holds whole slices."""
        from . import cabi
        # the phased sliced kernel runs whole rows in one pass (no column panels): the whole-row slice rule
        s = int(cabi.lib().isplib_suggest_slices_whole_rows(self.rows, self.ncols_padded, self.nnz, k)) if slices is None else slices
        if s <= 0:
            return None
        q = (s + self.world - 1) // self.world           # slices per shard: a divisor of the 192-row pitch unit
        while 192 % q:
            q += 1
        s = q * self.world
        table, ok = cabi.spmm_slices(self.rowptr, self.col_padded, self.ncols_padded, s)
        if not ok:
            return None
        return s, table, cabi.sliced_workspace(reduce, self.rows, k, s, self.col.device)

    def spmm_overlapped(self, x_shard: torch.Tensor, buf: torch.Tensor, out: torch.Tensor, plan,
                        reduce: str = "sum", arg: Optional[torch.Tensor] = None, gather: bool = True):
        """ONE all-gather of X, overlapped with compute: the column slices that lie in this rank's own
        shard are aggregated straight from `x_shard` while the collective runs; the remaining slices and
        the fold follow once it has landed.  Same slices and fold order as the non-overlapped sliced
        call, so the result is bitwise identical to it."""
        from . import cabi
        self._parked = None          # an error an EARLIER exchange left behind (its waits raised before _raise_parked) is not this one's
        s, table, work = plan
        k = x_shard.size(1)
        msg = cabi.MESSAGE[reduce]
        q = s // self.world
        first = self.rank * q
        handle = None
        if gather and self.world > 1:
            handle = dist.all_gather_into_tensor(buf, x_shard, group=self.group, async_op=True)
        elif gather:
            buf.copy_(x_shard)
        # local slices: column ids are in the padded layout, so shift the base onto the shard
        y_local = x_shard.data_ptr() - self.rank * self.max_rows * x_shard.stride(0) * 4
        common = (msg, self.rowptr, self.col_padded, self.val, table, s)
        self._kernel(cabi.fusedMM_csr_sliced_phase_hip, *common, first, q, False, y_local, self.ncols_padded, k, x_shard.stride(0),
                     out, arg, work)
        if handle is not None:
            handle.wait()
        # every other slice (the range wraps around modulo s), evenly over the 8 XCDs, then the fold
        self._kernel(cabi.fusedMM_csr_sliced_phase_hip, *common, (first + q) % s, s - q, True, buf.data_ptr(), self.ncols_padded, k,
                     buf.stride(0), out, arg, work)
        self._raise_parked()
        return out


    # ---- direct form: per-peer send / receive instead of one all-gather; shards are consumed as they land ----

    def post_direct(self, x_shard: torch.Tensor, buf: torch.Tensor, batches: int = 2):
        """Posts the P-1 send / receive pairs of one exchange of X in `batches` groups by ring distance (group b:
        distances [bounds[b], bounds[b+1]): send to rank + d, receive the shard of rank - d into its place in `buf`).
        Returns [(d0, d1, requests)] in the order the groups complete.  The own shard is not copied."""
        P, r = self.world, self.rank
        works = []
        if P > 1:
            # RCCL's send / receive are enqueued on a stream that waits for the caller's current stream.  Gloo's are not:
            # its SendWork / RecvWork (ProcessGroupGloo.hpp:200-245 of the installed torch) hold a tensor and an unbound
            # buffer over its data pointer and "are entirely completed by the device thread" -- no stream, no event,
            # unlike the collectives (initializeStreamsEvents, ProcessGroupGlooDetail.hpp:100-108) -- so the transport
            # reads `x_shard` and writes `buf` from the host the moment the call is posted, whatever the device still
            # has queued.  Kernels that produce x_shard, or that still write buf (a fill, the previous step's readers),
            # must therefore have finished before anything is posted.
            if not _p2p_is_stream_ordered(self.group, x_shard.device):
                torch.cuda.current_stream(x_shard.device).synchronize()
            batches = max(1, min(int(batches), P - 1))
            bounds = [1 + (P - 1) * b // batches for b in range(batches + 1)]
            for b in range(batches):
                ops = []
                for d in range(bounds[b], bounds[b + 1]):
                    src = (r - d) % P
                    ops.append(dist.P2POp(dist.isend, x_shard, (r + d) % P, group=self.group))
                    ops.append(dist.P2POp(dist.irecv, buf[src * self.max_rows:(src + 1) * self.max_rows], src, group=self.group))
                works.append((bounds[b], bounds[b + 1], dist.batch_isend_irecv(ops)))
        return works

    def spmm_direct(self, x_shard: torch.Tensor, buf: torch.Tensor, out: torch.Tensor, plan, reduce: str = "sum",
                    arg: Optional[torch.Tensor] = None, batches: int = 2):
        """The exchange of X as P-1 point-to-point transfers (over the xGMI full mesh every peer's shard
        has its own link, so nothing has to travel a ring), issued in `batches` groups by ring distance: the transfers
        of one group run concurrently (one RCCL group call: all its links busy), the groups one after the other, and
        the column slices that lie in a group's shards are aggregated as soon as that group has landed -- own shard
        first, straight from `x_shard`, while the first group is still on the links.  Same slices and the same fold
        order as `fusedMM_csr_sliced_hip` over the gathered buffer: bitwise equal to it for every reduction.
        `plan` = self.plan(k, reduce) (slice count a multiple of world).  batches = P-1 gives per-peer completion."""
        from . import cabi
        self._parked = None          # an error an EARLIER exchange left behind (its waits raised before _raise_parked) is not this one's
        s, table, work = plan
        k = x_shard.size(1)
        msg = cabi.MESSAGE[reduce]
        P, r = self.world, self.rank
        q = s // P
        common = (msg, self.rowptr, self.col_padded, self.val, table, s)
        works = self.post_direct(x_shard, buf, batches)
        # own shard: column ids are in the padded layout, so shift the base onto the shard
        y_local = x_shard.data_ptr() - r * self.max_rows * x_shard.stride(0) * 4
        self._kernel(cabi.fusedMM_csr_sliced_phase_hip, *common, r * q, q, P == 1, y_local, self.ncols_padded, k, x_shard.stride(0), out, arg, work)
        for i, (d0, d1, reqs) in enumerate(works):
            for req in reqs:
                req.wait()
            first = ((r - (d1 - 1)) % P) * q                 # shards r-d1+1 .. r-d0, ascending modulo P
            self._kernel(cabi.fusedMM_csr_sliced_phase_hip, *common, first, (d1 - d0) * q, i == len(works) - 1, buf.data_ptr(),
                         self.ncols_padded, k, buf.stride(0), out, arg, work)
        self._raise_parked()
        return out

    # ---- pipelined form: K is cut into panels, panel c+1 travels while panel c is aggregated -------

    def task_plan(self, slices: int, chunk: int = 1024, short_row: int = 128):
        """Task plan (isplib_amd.plan) of this rank's rows over the padded gather layout; None if unsorted."""
        from . import cabi
        from .plan import build_task_plan
        if getattr(self, "_col32", None) is None:
            self._col32 = cabi.pack_indices(self.col_padded)
        return build_task_plan(self.rowptr, self.col_padded, self.ncols_padded, slices, chunk, short_row, col32=self._col32)

    def _stream_plan(self, geom, minmax: bool = False):
        """Stream plan of this rank's rows over the padded gather layout, geom = (streams, slices, chunk): the library's own
        builder (rocPRIM sorts, ~10 ms at Reddit size; no torch kernel has to be loaded for it), weights gathered through
        the plan's permutation.  minmax: a plan of the max / min kernel's geometry (column-sorted rows; the padded column ids
        are monotone in the original ones, so a sorted row stays sorted).  None where the builder declines."""
        from .plan import build_stream_plan_native
        plan = build_stream_plan_native(self.rowptr, self.col_padded, self.ncols_padded, geom[1], geom[0], geom[2], minmax=minmax)
        if plan is not None and self.val is not None:
            plan.set_values(self.val)
        return plan

    def pipeline_state(self, k: int, panels: int, reduce: str = "sum", tplan=None, stream: bool = False):
        """Operands of `spmm_pipelined` for width k: panel bounds (multiples of 4 columns), one send and one
        gather buffer per panel, a task plan whose slice count suits the PANEL width (built here unless given;
        None -> no plan possible) and the task workspace of the widest panel.  stream=True / a (streams, slices,
        chunk) triple (sum / mean): a stream plan
        for the panel width instead (isplib_suggest_stream decides; None when it declines) -- the stream kernel works
        in column panels anyway, so cutting K = 128 into two 64-column collectives costs it nothing."""
        from .plugin import suggest_slices
        dev = self.col.device
        w = ((k + panels - 1) // panels + 3) // 4 * 4
        if stream:
            from . import cabi
            if reduce not in ("sum", "mean"):
                return None
            # stream may also be the plan parameters themselves, (streams, slices, chunk): tests, experiments
            geom = tuple(stream) if isinstance(stream, (tuple, list)) else cabi.suggest_stream(self.rows, self.ncols_padded, self.nnz, w, self.val is not None)
            if geom is None:
                return None
            plans = self.__dict__.setdefault("_stream_plans", {})
            if geom not in plans:
                plans[geom] = self._stream_plan(geom)
            tplan = plans[geom]
            if tplan is None:
                return None
        elif tplan is None:
            s = max(1, suggest_slices(self.rows, self.ncols_padded, self.nnz, w, reduce in ("max", "min")))
            plans = self.__dict__.setdefault("_task_plans", {})
            if s not in plans:
                plans[s] = self.task_plan(s)
            tplan = plans[s]
            if tplan is None:
                return None
        bounds = [(c0, min(k, c0 + w)) for c0 in range(0, k, w)]
        if bounds[-1][1] - bounds[-1][0] < 4 and len(bounds) > 1:       # the kernels need >= 4 columns
            last = bounds.pop()
            bounds[-1] = (bounds[-1][0], last[1])
        send = [torch.zeros((self.max_rows, c1 - c0), dtype=torch.float32, device=dev) for c0, c1 in bounds]
        recv = [torch.empty((self.ncols_padded, c1 - c0), dtype=torch.float32, device=dev) for c0, c1 in bounds]
        work = tplan.workspace() if stream else tplan.workspace(reduce, max(c1 - c0 for c0, c1 in bounds))
        return bounds, send, recv, tplan, work

    def spmm_pipelined(self, x_shard: torch.Tensor, out: torch.Tensor, state, reduce: str = "sum",
                       arg: Optional[torch.Tensor] = None):
        """The all-gather of X cut into column panels: every panel is its own collective, issued up front, and
        the task-list SpMM of panel c runs while panels c+1.. are still on the links.  Columns are independent
        in an SpMM: the result is bitwise that of the task-list SpMM run panel by panel (max/min: also bitwise the
        unpanelled call; sums can differ from it in the last bits because a wave's edge-slot count follows the
        panel width).  A K = 128 aggregation costs the same as two K = 64 ones on this hardware (DESIGN.md
        section 5), so the panels are free on the compute side and hide all but the first panel's transfer."""
        from . import cabi
        self._parked = None          # an error an EARLIER exchange left behind (its waits raised before _raise_parked) is not this one's
        bounds, send, recv, tplan, work = state
        msg = cabi.MESSAGE[reduce]
        handles = []
        for (c0, c1), s_buf, r_buf in zip(bounds, send, recv):
            s_buf[: x_shard.size(0)].copy_(x_shard[:, c0:c1])
            if self.world > 1:
                handles.append(dist.all_gather_into_tensor(r_buf, s_buf, group=self.group, async_op=True))
            else:
                r_buf.copy_(s_buf)
                handles.append(None)
        for (c0, c1), r_buf, handle in zip(bounds, recv, handles):
            if handle is not None:
                handle.wait()
            if hasattr(tplan, "words"):                     # a stream plan (sum / mean)
                self._kernel(cabi.fusedMM_csr_stream_hip, msg, self.rowptr, self.nnz, tplan, r_buf, out[:, c0:c1], work)
            else:
                self._kernel(cabi.fusedMM_csr_tasks_hip, msg, self.rowptr, self.col_padded, self.val, tplan, r_buf, out[:, c0:c1],
                             None if arg is None else arg[:, c0:c1], work)
        self._raise_parked()
        return out


class _DistSpMM(torch.autograd.Function):
    """out[R_p] = A[R_p, :] @ allgather(X);  dX[R_p] = A^T[R_p, :] @ allgather(dY)  (csrc/fusedmm.cpp:285)."""

    @staticmethod
    def forward(ctx, x_local, graph):
        ctx.graph = graph
        return graph.fwd.spmm_auto(x_local)

    @staticmethod
    def backward(ctx, grad_out):
        return ctx.graph.bwd.spmm_auto(grad_out.contiguous()), None


class _DistSpMMMean(torch.autograd.Function):
    """mean forward; dX[R_p] = A^T[R_p, :] @ allgather(dY) with the weights value[csr2csc] / max(deg, 1)[row[csr2csc]]
    (csrc/fusedmm.cpp:357-375).  For an unweighted graph that weight is 1 / deg of the edge's row in A, i.e. a scale of
    the rows of dY: applied to this rank's rows before the exchange, and A^T stays on the unit-weight path."""

    @staticmethod
    def forward(ctx, x_local, graph):
        ctx.graph = graph
        return graph.fwd.spmm_auto(x_local, "mean")

    @staticmethod
    def backward(ctx, grad_out):
        g = ctx.graph
        if g.fwd.val is None:
            return g.bwd.spmm_auto((grad_out * g.inv_deg().unsqueeze(1)).contiguous()), None
        return g.bwd_mean().spmm_auto(grad_out.contiguous()), None


class _DistSpMMMinMax(torch.autograd.Function):
    """max / min forward (values; the winners' GLOBAL CSR positions are kept for the backward).  Backward
    (csrc/fusedmm.cpp:410-451): dX[col[arg[i,c]], c] += val[arg[i,c]] * dY[i,c] -- the destination row may live on any
    rank, so the exchange carries, for every (i, c), the destination's global row and the weighted gradient (ONE
    all-gather of each, in global row order), and every rank adds up what lands in its own rows in ascending i
    (isplib_scatter_rows_det_hip: sort-based, no atomics)."""

    @staticmethod
    def forward(ctx, x_local, graph, reduce):
        out, arg = graph.fwd.spmm_auto(x_local, reduce)
        ctx.graph = graph
        ctx.save_for_backward(arg)
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, grad_out, _grad_arg):
        (arg,) = ctx.saved_tensors
        return ctx.graph.fwd.minmax_backward(arg, grad_out.contiguous()), None, None


class DistGraph:
    """This rank's rows of A and of A^T (same row boundaries) for full-batch GNN training on one node:
    `matmul(x_local, reduce)` is the aggregation of the 1-D row-partitioned graph with autograd; every call
    is ONE all-gather + the local SpMM, forward and backward alike (max / min backward: one all-gather of the
    destinations and one of the weighted gradients).  Dense layer weights are replicated by the caller and their
    gradients all-reduced (outside the SpMM path)."""

    def __init__(self, rowptr, col, val, n, rank, world, group=None):
        from . import cabi
        self.fwd = RowPartition(rowptr, col, val, n, rank, world, group=group)
        colptr, _, row_t, val_t = cabi.csr2csc(rowptr, col, val, n, want_perm=False, want_val=val is not None)
        self.bwd = RowPartition(colptr, row_t, val_t, n, rank, world, cuts=self.fwd.row_cuts, group=group)
        self.row0, self.rows = self.fwd.row0, self.fwd.rows
        self._bwd_mean = None
        self._inv_deg = None
        self._graph = (rowptr, col, val, n, rank, world, group) if val is not None else None     # for the mean backward's weights

    def inv_deg(self) -> torch.Tensor:
        """1 / max(deg, 1) of this rank's rows of A."""
        if self._inv_deg is None:
            rp = self.fwd.rowptr
            self._inv_deg = 1.0 / (rp[1:] - rp[:-1]).clamp(min=1).to(torch.float32)
        return self._inv_deg

    def bwd_mean(self) -> "RowPartition":
        """A^T with the mean backward's weights (weighted graphs only; built on first use)."""
        if self._bwd_mean is None:
            from . import cabi
            rowptr, col, val, n, rank, world, group = self._graph
            colptr, _, row_t, val_t = cabi.csr2csc(rowptr, col, val, n, mean_scale=True, want_perm=False)
            self._bwd_mean = RowPartition(colptr, row_t, val_t, n, rank, world, cuts=self.fwd.row_cuts, group=group)
        return self._bwd_mean

    def matmul(self, x_local: torch.Tensor, reduce: str = "sum") -> torch.Tensor:
        if reduce in ("sum", "add"):
            return _DistSpMM.apply(x_local, self)
        if reduce == "mean":
            return _DistSpMMMean.apply(x_local, self)
        if reduce in ("max", "min"):
            return _DistSpMMMinMax.apply(x_local, self, reduce)[0]
        raise ValueError(f"isplib: unknown reduce '{reduce}' (expected sum|add|mean|max|min)")


def _local_ops(self, k: int, reduce: str = "sum"):
    """The schedule of this rank's local SpMM once the dense operand is complete in the padded gather buffer -- the
    single-GPU rules applied to the rank's shard: ("stream", plan, workspace) where isplib_suggest_stream (sum / mean) or
    isplib_suggest_stream_minmax (max / min; column-sorted rows -- the padded column ids keep the order) accepts it, else
    ("tasks", plan, workspace) where isplib_suggest_slices gives a slice count, else ("plain", None, None).  Cached per
    (k, reduce).  ISPLIB_STREAM=0 / ISPLIB_SLICES as in the plug-in."""
    import os
    from . import cabi
    from .plugin import suggest_slices
    kind = "minmax" if reduce in ("max", "min") else "sum"
    cache = self.__dict__.setdefault("_local", {})
    key = (k, kind)
    if key in cache:
        return cache[key]
    ops = None
    stream_off = os.environ.get("ISPLIB_STREAM") == "0" or os.environ.get("ISPLIB_SLICES") is not None
    if not stream_off and self.rows > 0 and self.nnz > 0:
        if kind == "minmax":
            geom = cabi.suggest_stream_minmax(self.rows, self.ncols_padded, self.nnz, k)
        else:
            geom = cabi.suggest_stream(self.rows, self.ncols_padded, self.nnz, k, self.val is not None)
        if geom is not None:
            plans = self.__dict__.setdefault("_stream_plans", {})
            pkey = tuple(geom) + ((kind,) if kind == "minmax" else ())
            if pkey not in plans:
                plans[pkey] = self._stream_plan(geom, minmax=kind == "minmax")
            sp = plans[pkey]
            if sp is not None and not (kind == "minmax" and sp.perm.dtype != torch.int32):
                ops = ("stream", sp, sp.workspace(minmax=kind == "minmax"))
    if ops is None:
        forced = os.environ.get("ISPLIB_SLICES")
        s = int(forced) if forced is not None else suggest_slices(self.rows, self.ncols_padded, self.nnz, k, kind == "minmax")
        if s > 0 and k >= 4 and self.nnz > 0:
            plans = self.__dict__.setdefault("_task_plans", {})
            if s not in plans:
                plans[s] = self.task_plan(s)
            if plans[s] is not None:
                ops = ("tasks", plans[s], plans[s].workspace("max" if kind == "minmax" else "sum", k))
    if ops is None:
        ops = ("plain", None, None)
    cache[key] = ops
    return ops


def _local_spmm(self, ops, buf: torch.Tensor, out: torch.Tensor, reduce: str = "sum", arg: Optional[torch.Tensor] = None):
    """out[rows, K] (and arg: LOCAL CSR positions, sentinel = this rank's nnz) from the gathered buffer on schedule `ops`."""
    from . import cabi
    msg = cabi.MESSAGE[reduce]
    kind, plan, work = ops
    if kind == "stream" and reduce in ("max", "min"):
        cabi.fusedMM_csr_stream_minmax_hip(msg, self.rowptr, self.nnz, plan, buf, out, arg, work)
    elif kind == "stream":
        cabi.fusedMM_csr_stream_hip(msg, self.rowptr, self.nnz, plan, buf, out, work)
    elif kind == "tasks":
        cabi.fusedMM_csr_tasks_hip(msg, self.rowptr, self.col_padded, self.val, plan, buf, out, arg, work)
    else:
        cabi.fusedMM_csr_hip(msg, self.rowptr, self.col_padded, self.val, buf, out, arg)
    return out


def _global_arg(self, arg: torch.Tensor) -> torch.Tensor:
    """LOCAL CSR positions -> GLOBAL ones (+ this rank's edge offset; 'no winner' becomes the global nnz)."""
    return torch.where(arg == self.nnz, arg.new_full((), self.total_nnz), arg + self.edge0)


def _spmm_auto(self, x_local: torch.Tensor, reduce: str = "sum"):
    """SpMM of this partition on unpadded local rows [x_rows, K]: pads into the shard pitch, then
    ISPLIB_DIST_SCHEDULE = tasks (default: ONE all-gather, then the schedule the single-GPU rules pick for this rank's
    shard -- `local_ops`: stream schedule / task list / plain kernel, every reduction) | overlap (local column slices
    during the all-gather) | pipelined (two column panels, panel 2 travels while panel 1 is aggregated; on the stream
    schedule where isplib_suggest_stream accepts the panel (sum / mean), else on the task list) | direct
    (per-peer send / receive in ISPLIB_DIRECT_BATCHES groups, shards aggregated as they land).
    Whatever the schedule, a graph for which the slice rule says 0 runs gather + the plain kernel.
    Returns out for sum / mean, (out, arg) for max / min with arg = GLOBAL CSR positions (global nnz = no winner)."""
    import os
    from .plugin import suggest_slices
    if reduce == "add":
        reduce = "sum"
    if reduce not in ("sum", "mean", "max", "min"):
        raise ValueError(f"isplib: unknown reduce '{reduce}' (expected sum|add|mean|max|min)")
    minmax = reduce in ("max", "min")
    k = x_local.size(1)
    mode = os.environ.get("ISPLIB_DIST_SCHEDULE", "tasks")
    if mode not in ("tasks", "overlap", "pipelined", "direct"):
        raise ValueError(f"ISPLIB_DIST_SCHEDULE={mode!r}: expected tasks | overlap | pipelined | direct")
    cache = self.__dict__.setdefault("_auto", {})
    key = (k, mode, reduce)
    if key not in cache:
        ops = None
        if mode in ("overlap", "direct"):
            if suggest_slices(self.rows, self.ncols_padded, self.nnz, k, minmax) > 0:
                forced = os.environ.get("ISPLIB_SLICES")        # one-pass sliced kernel: its own (whole-row) slice rule
                ops = self.plan(k, reduce, slices=int(forced) if forced else None)
        elif mode == "pipelined" and k >= 32:
            ops = None if minmax else self.pipeline_state(k, 2, reduce, stream=True)
            if ops is None and suggest_slices(self.rows, self.ncols_padded, self.nnz, k, minmax) > 0:
                ops = self.pipeline_state(k, 2, reduce)
        if self.world > 1 and mode != "tasks":
            # these exchange X with other collectives than the one all-gather of the fallback: every rank must take
            # the same branch, and whether a plan exists is decided from the rank's own shard
            ok = torch.tensor([0 if ops is None else 1], dtype=torch.int32, device=x_local.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if not int(ok):
                ops = None
        if ops is None:
            # ONE all-gather, then the single-GPU rule on this rank's shard (the collective is the same whatever a rank
            # decides, so ranks may decide differently)
            mode_run, ops = "tasks", self.local_ops(k, reduce)
        else:
            mode_run = mode
        cache[key] = (mode_run, ops, self.gather_buffer(k, x_local.device),
                      torch.zeros((self.max_rows, k), dtype=torch.float32, device=x_local.device))
    mode_run, ops, buf, shard = cache[key]
    shard[: self.x_rows].copy_(x_local)
    out = torch.empty((self.rows, k), dtype=torch.float32, device=x_local.device)
    arg = torch.empty((self.rows, k), dtype=torch.int64, device=x_local.device) if minmax else None
    if mode_run == "overlap":
        self.spmm_overlapped(shard, buf, out, ops, reduce, arg)
    elif mode_run == "direct":
        self.spmm_direct(shard, buf, out, ops, reduce, arg, batches=int(os.environ.get("ISPLIB_DIRECT_BATCHES", "2")))
    elif mode_run == "pipelined":
        self.spmm_pipelined(shard, out, ops, reduce, arg)
    else:
        self.all_gather(shard, buf)
        self.local_spmm(ops, buf, out, reduce, arg)
    return (out, self.global_arg(arg)) if minmax else out


def _minmax_backward(self, arg_global: torch.Tensor, grad_out: torch.Tensor) -> torch.Tensor:
    """dX of this rank's rows of the dense operand for a max / min forward of this partition: arg_global [rows, K] as
    `spmm_auto` returned it, grad_out [rows, K].  Two all-gathers (int32 destinations, fp32 weighted gradients; padded to
    the longest shard), then the deterministic scatter into the rank's own rows."""
    from . import cabi
    k = grad_out.size(1)
    ok = arg_global != self.total_nnz
    a = (arg_global - self.edge0).clamp_(0, max(self.nnz - 1, 0))
    if self.nnz == 0:
        dest = torch.full(arg_global.shape, -1, dtype=torch.int32, device=grad_out.device)
        gval = torch.zeros_like(grad_out)
    else:
        dest = torch.where(ok, self.col[a], self.col.new_full((), -1)).to(torch.int32)
        gval = torch.where(ok, grad_out if self.val is None else self.val[a] * grad_out, grad_out.new_zeros(()))
    pad = max(self.row_cuts[p + 1] - self.row_cuts[p] for p in range(self.world))
    if self.world > 1:
        d_send = torch.full((pad, k), -1, dtype=torch.int32, device=grad_out.device)
        g_send = torch.zeros((pad, k), dtype=torch.float32, device=grad_out.device)
        d_send[: self.rows] = dest
        g_send[: self.rows] = gval
        d_all = torch.empty((self.world * pad, k), dtype=torch.int32, device=grad_out.device)
        g_all = torch.empty((self.world * pad, k), dtype=torch.float32, device=grad_out.device)
        dist.all_gather_into_tensor(d_all, d_send, group=self.group)
        dist.all_gather_into_tensor(g_all, g_send, group=self.group)
    else:
        d_all, g_all = dest.contiguous(), gval.contiguous()
    return self.scatter_rows(d_all, g_all, self.x_cuts[self.rank], self.x_rows)


RowPartition.spmm_auto = _spmm_auto
RowPartition.local_ops = _local_ops
RowPartition.local_spmm = _local_spmm
RowPartition.global_arg = _global_arg
RowPartition.minmax_backward = _minmax_backward


def _scatter_rows(dest: torch.Tensor, gval: torch.Tensor, lo: int, n: int) -> torch.Tensor:
    """The local kernel of `minmax_backward` (isplib_scatter_rows_det_hip); a hook, so that the CPU / gloo tests of the
    exchange can put a NumPy statement of it in its place."""
    from . import cabi
    return cabi.scatter_rows_det(dest, gval, lo, n)


RowPartition.scatter_rows = staticmethod(_scatter_rows)
