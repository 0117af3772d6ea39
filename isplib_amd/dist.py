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
            arg = torch.where(arg == self.nnz, arg.new_full((), self.total_nnz), arg + self.edge0)
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

    def _stream_plan(self, geom):
        """Stream plan of this rank's rows over the padded gather layout, geom = (streams, slices, chunk): the library's own
        builder (rocPRIM sorts, ~10 ms at Reddit size; no torch kernel has to be loaded for it), weights gathered through
        the plan's permutation.  None where the builder declines."""
        from .plan import build_stream_plan_native
        plan = build_stream_plan_native(self.rowptr, self.col_padded, self.ncols_padded, geom[1], geom[0], geom[2])
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


class DistGraph:
    """This rank's rows of A and of A^T (same row boundaries) for full-batch GNN training on one node:
    `matmul(x_local)` is the sum-aggregation of the 1-D row-partitioned graph with autograd; every call
    is ONE all-gather + the local SpMM, forward and backward alike.  Dense layer weights are replicated
    by the caller and their gradients all-reduced (outside the SpMM path)."""

    def __init__(self, rowptr, col, val, n, rank, world, group=None):
        from . import cabi
        self.fwd = RowPartition(rowptr, col, val, n, rank, world, group=group)
        colptr, _, row_t, val_t = cabi.csr2csc(rowptr, col, val, n, want_perm=False, want_val=val is not None)
        self.bwd = RowPartition(colptr, row_t, val_t, n, rank, world, cuts=self.fwd.row_cuts, group=group)
        self.row0, self.rows = self.fwd.row0, self.fwd.rows

    def matmul(self, x_local: torch.Tensor) -> torch.Tensor:
        return _DistSpMM.apply(x_local, self)


def _spmm_auto(self, x_local: torch.Tensor) -> torch.Tensor:
    """Sum-SpMM of this partition on unpadded local rows [x_rows, K]: pads into the shard pitch, then
    ISPLIB_DIST_SCHEDULE = tasks (default: one all-gather, then the stream schedule / task list / plain kernel by the
    single-GPU rules applied to this rank's shard) | overlap (local column slices
    during the all-gather) | pipelined (two column panels, panel 2 travels while panel 1 is aggregated; on the stream
    schedule where isplib_suggest_stream accepts the panel, else on the task list) | direct
    (per-peer send / receive in ISPLIB_DIRECT_BATCHES groups, shards aggregated as they land).
    Whatever the schedule, a graph for which the slice rule says 0 runs gather + the plain kernel."""
    import os
    from . import cabi
    from .plugin import suggest_slices
    k = x_local.size(1)
    mode = os.environ.get("ISPLIB_DIST_SCHEDULE", "tasks")
    if mode not in ("tasks", "overlap", "pipelined", "direct"):
        raise ValueError(f"ISPLIB_DIST_SCHEDULE={mode!r}: expected tasks | overlap | pipelined | direct")
    cache = self.__dict__.setdefault("_auto", {})
    key = (k, mode)
    if key not in cache:
        s = suggest_slices(self.rows, self.ncols_padded, self.nnz, k)
        ops = None
        if s > 0 and mode in ("overlap", "direct"):
            forced = os.environ.get("ISPLIB_SLICES")        # one-pass sliced kernel: its own (whole-row) slice rule
            ops = self.plan(k, "sum", slices=int(forced) if forced else None)
        elif mode == "pipelined" and k >= 32:
            ops = self.pipeline_state(k, 2, "sum", stream=True)
            if ops is None and s > 0:
                ops = self.pipeline_state(k, 2, "sum")
        else:
            # one all-gather, then the single-GPU rule on this rank's shard: stream schedule where isplib_suggest_stream
            # accepts it, else the task list, else the plain kernel (the collective is the same in all three: ranks
            # may decide differently)
            geom = None if os.environ.get("ISPLIB_STREAM") == "0" else cabi.suggest_stream(self.rows, self.ncols_padded, self.nnz, k, self.val is not None)
            if geom is not None:
                sp = self._stream_plan(geom)
                ops = None if sp is None else (sp, sp.workspace())
            if ops is None and s > 0:
                plans = self.__dict__.setdefault("_task_plans", {})
                if s not in plans:
                    plans[s] = self.task_plan(s)
                ops = None if plans[s] is None else (plans[s], plans[s].workspace("sum", k))
        if self.world > 1 and mode in ("pipelined", "direct"):
            # these two exchange X with other collectives than the one all-gather of the fallback: every rank must take
            # the same branch, and whether a plan exists is decided from the rank's own shard
            ok = torch.tensor([0 if ops is None else 1], dtype=torch.int32, device=x_local.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if not int(ok):
                ops = None
        cache[key] = (ops, self.gather_buffer(k, x_local.device),
                      torch.zeros((self.max_rows, k), dtype=torch.float32, device=x_local.device))
    ops, buf, shard = cache[key]
    shard[: self.x_rows].copy_(x_local)
    out = torch.empty((self.rows, k), dtype=torch.float32, device=x_local.device)
    if ops is not None and mode == "overlap":
        return self.spmm_overlapped(shard, buf, out, ops, "sum")
    if ops is not None and mode == "direct":
        return self.spmm_direct(shard, buf, out, ops, "sum", batches=int(os.environ.get("ISPLIB_DIRECT_BATCHES", "2")))
    if ops is not None and mode == "pipelined" and k >= 32:
        return self.spmm_pipelined(shard, out, ops, "sum")
    self.all_gather(shard, buf)
    if ops is not None and hasattr(ops[0], "words"):
        cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, self.rowptr, self.nnz, ops[0], buf, out, ops[1])
    elif ops is not None:
        cabi.fusedMM_csr_tasks_hip(cabi.MSG_SPMM_SUM, self.rowptr, self.col_padded, self.val, ops[0], buf, out, None, ops[1])
    else:
        cabi.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, self.rowptr, self.col_padded, self.val, buf, out)
    return out


RowPartition.spmm_auto = _spmm_auto
