"""Per-graph task plan of the task-list SpMM schedule (``fusedMM_csr_tasks_hip``).

Built ONCE per graph and slice count, on the device, independent of K.  Every non-empty
(row, column-slice) segment is cut into chunks of at most ``chunk`` edges; a chunk is a task
(= one wavefront).  Rows shorter than ``short_row`` are not sliced: the whole row is one segment,
homed on slice ``row % slices`` so short rows spread over the eight XCD lanes.  Tasks are stored
slice-major; ``lane_off`` cuts the list into eight contiguous runs of equal edge mass (one per XCD lane)
and ``seg_off`` maps (slice, row) to the segment's first task so the combine kernel can fold a row's
partials in ascending CSR order.  Any slice count from 1 (pure load balancing: hub rows become many
tasks) to 4096 is accepted.

Everything is built on the device through the C ABI (``isplib_spmm_slices_build_hip``,
``isplib_spmm_tasks_count_hip``, ``isplib_spmm_tasks_fill_hip``); this module only owns the buffers.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import cabi


@dataclass
class TaskPlan:
    slices: int
    n_tasks: int
    task_row: torch.Tensor     # int32 [n_tasks]
    task_b: torch.Tensor       # int64 [n_tasks]
    task_len: torch.Tensor     # int32 [n_tasks]
    seg_off: torch.Tensor      # int32 [slices*m + 1], slice-major
    lane_off: list             # 9 host ints
    chunk: int
    short_row: int
    col32: Optional[torch.Tensor] = None   # int32 [nnz]: the column ids packed once per graph; the task kernels
    #                                        stream these 4 bytes per edge instead of the 8 of the int64 array

    def workspace(self, reduce: str, k: int) -> torch.Tensor:
        nbytes = cabi.lib().isplib_spmm_tasks_workspace_bytes(cabi.MESSAGE[reduce], self.n_tasks, k)
        return torch.empty(nbytes, dtype=torch.uint8, device=self.task_row.device)


def build_task_plan(rowptr: torch.Tensor, col: torch.Tensor, ncols: int, slices: int, chunk: int = 1024,
                    short_row: int = 128, col32: Optional[torch.Tensor] = None) -> Optional[TaskPlan]:
    """Slice table -> task counts + prefix (one host round trip for the task count) -> task arrays, all
    through the C ABI.  None when the rows are not column-sorted (the slice table would be meaningless).
    `col32`: a packed copy of `col` shared between the plans of one graph (made here when not given)."""
    import ctypes
    assert 1 <= slices <= 4096
    m = rowptr.numel() - 1
    dev = col.device
    if m * slices + col.numel() // chunk + 1 >= 2 ** 31:        # task ids and seg_off are int32
        return None
    table, ok = cabi.spmm_slices(rowptr, col, ncols, slices)
    if not ok:
        return None
    L = cabi.lib()
    rp = rowptr.data_ptr()
    pb, pe = ctypes.c_void_p(rp), ctypes.c_void_p(rp + 8)
    seg_off = torch.empty(slices * m + 1, dtype=torch.int32, device=dev)
    info = cabi.TaskPlanInfo()
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ws = L.isplib_spmm_tasks_plan_workspace_bytes(m, slices)
        work = torch.empty(max(ws, 256), dtype=torch.uint8, device=dev)
        st = L.isplib_spmm_tasks_count_hip(m, pb, pe, ctypes.c_void_p(table.data_ptr()), slices, chunk, short_row,
                                           ctypes.c_void_p(seg_off.data_ptr()), ctypes.c_void_p(work.data_ptr()),
                                           work.numel(), ctypes.byref(info), stream)
        cabi._check(st, "isplib_spmm_tasks_count_hip")
        n_tasks = int(info.n_tasks)
        task_row = torch.empty(max(n_tasks, 1), dtype=torch.int32, device=dev)
        task_b = torch.empty(max(n_tasks, 1), dtype=torch.int64, device=dev)
        task_len = torch.empty(max(n_tasks, 1), dtype=torch.int32, device=dev)
        st = L.isplib_spmm_tasks_fill_hip(m, pb, pe, ctypes.c_void_p(table.data_ptr()), ctypes.byref(info),
                                          ctypes.c_void_p(seg_off.data_ptr()), ctypes.c_void_p(task_row.data_ptr()),
                                          ctypes.c_void_p(task_b.data_ptr()), ctypes.c_void_p(task_len.data_ptr()), stream)
        cabi._check(st, "isplib_spmm_tasks_fill_hip")
    if col32 is None:
        col32 = cabi.pack_indices(col)
    return TaskPlan(slices, n_tasks, task_row[:n_tasks], task_b[:n_tasks], task_len[:n_tasks], seg_off,
                    [int(v) for v in info.lane_off], chunk, short_row, col32)


@dataclass
class SweepPlan:
    """Plan of the sweep schedule (``fusedMM_csr_sweep_hip``, include/isplib_hip.h: isplib_sweep_plan)."""
    rows: int
    slices: int
    gens: int
    waves_per_gen: int
    rows_per_wave: int
    n_tasks: int
    n_parts: int
    n_hub: int
    wave_row: torch.Tensor       # int32 [gens*waves_per_gen*rows_per_wave]
    wave_part: torch.Tensor      # int32, same shape
    wave_task_off: torch.Tensor  # int64 [gens*waves_per_gen + 1]
    task_b: torch.Tensor         # int64 [n_tasks]
    task_meta: torch.Tensor      # int32 [n_tasks]  (slot << 24) | edges
    hub_row: torch.Tensor        # int32 [n_hub]
    hub_off: torch.Tensor        # int32 [n_hub + 1]
    chunk: int
    min_seg: int
    col32: Optional[torch.Tensor] = None

    def struct(self) -> "cabi.SweepPlanStruct":
        p = lambda t: t.data_ptr() if t is not None and t.numel() else None  # noqa: E731
        return cabi.SweepPlanStruct(self.rows, self.slices, self.gens, self.waves_per_gen, self.rows_per_wave, self.n_tasks,
                                    self.n_parts, self.n_hub, p(self.wave_row), p(self.wave_part), p(self.wave_task_off),
                                    p(self.task_b), p(self.task_meta), p(self.hub_row), p(self.hub_off))

    def workspace(self, reduce: str, k: int) -> torch.Tensor:
        import ctypes
        ps = self.struct()
        nbytes = cabi.exp_lib().isplib_spmm_sweep_workspace_bytes(cabi.MESSAGE[reduce], ctypes.byref(ps), k)
        return torch.empty(nbytes, dtype=torch.uint8, device=self.wave_row.device)


def sweep_plan_arrays(rowptr: torch.Tensor, table: torch.Tensor, slices: int, waves_per_gen: int, rows_per_wave: int = 16,
                      chunk: int = 2048, min_seg: int = 16) -> dict:
    """The arrays of a sweep plan from the slice table (``table[i, s]`` = first CSR position of row i with column
    >= s * ceil(n / slices), ``isplib_spmm_slices_build_hip``); plain torch ops on the table's device.

      * rows over `chunk` edges are cut evenly into virtual rows (so no wave is longer than the others by a hub row);
      * virtual rows are dealt to the gens*waves_per_gen waves longest first, back and forth: every wave gets the
        same number of rows (<= rows_per_wave) and, within a fraction of a percent, the same number of edges;
      * a virtual row of e edges is cut at the boundaries of min(slices, e // min_seg) groups of adjacent column
        slices (short rows are not shredded into segments of two or three edges); a segment is a task, met by its
        wave in the phase of its first slice;
      * a wave's tasks are stored in (slice, slot) order."""
    assert 1 <= slices <= 4096 and rows_per_wave in (8, 16, 32) and waves_per_gen >= 1 and 1 <= chunk < (1 << 24)
    m = rowptr.numel() - 1
    dev = table.device
    table = table.view(m, slices + 1)
    i64 = dict(dtype=torch.int64, device=dev)
    deg = rowptr[1:] - rowptr[:-1]
    nchunk = ((deg + chunk - 1) // chunk).clamp(min=1)
    csize = (deg + nchunk - 1) // nchunk
    nv = int(nchunk.sum())
    vrow = torch.repeat_interleave(torch.arange(m, **i64), nchunk)
    first = torch.cumsum(nchunk, 0) - nchunk
    ci = torch.arange(nv, **i64) - first[vrow]
    vb = rowptr[vrow] + ci * csize[vrow]
    ve = torch.minimum(vb + csize[vrow], rowptr[vrow + 1])
    vb = torch.minimum(vb, ve)
    vlen = ve - vb
    # deal the virtual rows to the waves, longest first, boustrophedon
    gens = max(1, -(-nv // (waves_per_gen * rows_per_wave)))
    nw = gens * waves_per_gen
    order = torch.sort(vlen, descending=True, stable=True).indices
    rank = torch.empty(nv, **i64)
    rank[order] = torch.arange(nv, **i64)
    rnd, pos = rank // nw, rank % nw
    wave = torch.where(rnd % 2 == 0, pos, nw - 1 - pos)
    slot = rnd
    is_hub = nchunk[vrow] > 1
    part = torch.where(is_hub, torch.cumsum(is_hub.to(torch.int64), 0) - 1, torch.full((nv,), -1, **i64))
    wave_row = torch.full((nw * rows_per_wave,), -1, dtype=torch.int32, device=dev)
    wave_part = torch.full((nw * rows_per_wave,), -1, dtype=torch.int32, device=dev)
    at = wave * rows_per_wave + slot
    wave_row[at] = vrow.to(torch.int32)
    wave_part[at] = part.to(torch.int32)
    hub_rows = torch.nonzero(nchunk > 1).flatten()
    hub_off = torch.zeros(hub_rows.numel() + 1, dtype=torch.int32, device=dev)
    if hub_rows.numel():
        hub_off[1:] = torch.cumsum(nchunk[hub_rows], 0).to(torch.int32)
    # segments: groups of adjacent slices per virtual row
    s_r = (vlen // max(min_seg, 1)).clamp(min=1, max=slices)                       # groups of this virtual row
    sl = torch.arange(slices, **i64)
    grp = (sl[None, :] * s_r[:, None]) // slices                                   # [nv, S] group of slice s
    is_start = torch.ones((nv, slices), dtype=torch.bool, device=dev)
    is_start[:, 1:] = grp[:, 1:] != grp[:, :-1]
    e_slice = ((grp + 1) * slices + s_r[:, None] - 1) // s_r[:, None]              # first slice of the next group
    tb = table[vrow]                                                               # [nv, S+1]
    seg_b = torch.minimum(torch.maximum(tb[:, :-1], vb[:, None]), ve[:, None])
    seg_e = torch.minimum(torch.maximum(torch.gather(tb, 1, e_slice.clamp(max=slices)), vb[:, None]), ve[:, None])
    seg_len = seg_e - seg_b
    keep = is_start & (seg_len > 0)
    del grp, e_slice, tb, is_start
    vi, si = torch.nonzero(keep, as_tuple=True)
    t_b, t_len = seg_b[vi, si], seg_len[vi, si]
    del seg_b, seg_e, seg_len, keep
    t_wave, t_slot = wave[vi], slot[vi]
    key = (t_wave * slices + si) * rows_per_wave + t_slot
    perm = torch.sort(key).indices
    t_b, t_len, t_wave, t_slot = t_b[perm], t_len[perm], t_wave[perm], t_slot[perm]
    wave_task_off = torch.zeros(nw + 1, **i64)
    if t_wave.numel():
        wave_task_off[1:] = torch.cumsum(torch.bincount(t_wave, minlength=nw), 0)
    task_meta = ((t_slot << 24) | t_len).to(torch.int32)
    return dict(rows=m, slices=slices, gens=gens, waves_per_gen=waves_per_gen, rows_per_wave=rows_per_wave,
                n_tasks=int(t_b.numel()), n_parts=int(is_hub.sum()), n_hub=int(hub_rows.numel()), wave_row=wave_row,
                wave_part=wave_part, wave_task_off=wave_task_off, task_b=t_b.contiguous(), task_meta=task_meta.contiguous(),
                hub_row=hub_rows.to(torch.int32), hub_off=hub_off, chunk=chunk, min_seg=min_seg)


def build_sweep_plan(rowptr: torch.Tensor, col: torch.Tensor, ncols: int, slices: int, waves_per_gen: int,
                     rows_per_wave: int = 16, chunk: int = 2048, min_seg: int = 16,
                     col32: Optional[torch.Tensor] = None) -> Optional[SweepPlan]:
    """Plan of the sweep schedule, built on the device (once per graph and geometry, independent of K): the slice
    table through the C ABI, the rest with torch ops (`sweep_plan_arrays`).  None when the rows are not column-sorted."""
    table, ok = cabi.spmm_slices(rowptr, col, ncols, slices)
    if not ok:
        return None
    arrays = sweep_plan_arrays(rowptr, table, slices, waves_per_gen, rows_per_wave, chunk, min_seg)
    del table
    if col32 is None:
        col32 = cabi.pack_indices(col)
    return SweepPlan(col32=col32, **arrays)


@dataclass
class StreamPlan:
    """Plan of the stream form of the sweep schedule (``fusedMM_csr_stream_hip``, include/isplib_hip.h:
    isplib_stream_plan).  Owns a copy of the edges in the order the waves walk them."""
    rows: int
    cols: int
    slices: int
    gens: int
    waves_per_gen: int
    rows_per_wave: int
    streams: int
    n_steps: int
    n_parts: int
    n_hub: int
    words: torch.Tensor          # int32 [n_steps*streams]
    vals: Optional[torch.Tensor]  # float32, same shape, or None (unit weights)
    perm: torch.Tensor           # int32/int64 [n_steps*streams]: CSR position of the word, -1 = padding
    wave_step_off: torch.Tensor  # int64 [gens*waves_per_gen + 1]
    wave_row: torch.Tensor       # int32 [gens*waves_per_gen*rows_per_wave]
    wave_part: torch.Tensor      # int32, same shape
    hub_row: torch.Tensor        # int32 [n_hub]
    hub_off: torch.Tensor        # int32 [n_hub + 1]
    chunk: int

    def struct(self) -> "cabi.StreamPlanStruct":
        p = lambda t: t.data_ptr() if t is not None and t.numel() else None  # noqa: E731
        return cabi.StreamPlanStruct(self.rows, self.cols, self.slices, self.gens, self.waves_per_gen, self.rows_per_wave,
                                     self.streams, int(self.chunk), self.n_steps, self.n_parts, self.n_hub, p(self.words), p(self.vals),
                                     p(self.wave_step_off), p(self.wave_row), p(self.wave_part), p(self.hub_row), p(self.hub_off),
                                     p(self.perm) if self.perm.dtype == torch.int32 else None)

    def workspace(self, minmax: bool = False) -> torch.Tensor:
        import ctypes
        ps = self.struct()
        L = cabi.lib()
        nbytes = (L.isplib_spmm_stream_minmax_workspace_bytes if minmax else L.isplib_spmm_stream_workspace_bytes)(ctypes.byref(ps))
        return torch.empty(nbytes, dtype=torch.uint8, device=self.words.device)

    def set_values(self, val: Optional[torch.Tensor]) -> None:
        """Weights in stream order (None = unit weights): one gather through `perm`, for callers whose weights change."""
        if val is None:
            self.vals = None
            return
        ok = self.perm >= 0
        out = torch.zeros(self.perm.numel(), dtype=torch.float32, device=self.perm.device)
        out[ok] = val.detach().to(torch.float32)[self.perm[ok].to(torch.int64)]
        self.vals = out


def stream_plan_arrays(rowptr: torch.Tensor, col: torch.Tensor, ncols: int, slices: int, waves_per_gen: int,
                       rows_per_wave: int = 16, streams: int = 4, chunk: int = 2048, pad_row: Optional[int] = None) -> dict:
    """The arrays of a stream plan; plain torch ops on the device of `col` (once per graph and geometry).

      * rows over `chunk` edges are dealt edge by edge, round robin, to ceil(deg / chunk) virtual rows: every
        virtual row then spans ALL column slices like an ordinary row does (contiguous pieces of a column-sorted
        row would each sit in a few slices, and a wave holding such pieces falls out of step with the others);
      * virtual rows are dealt longest first, back and forth, to the gens*waves_per_gen*streams STREAMS (a stream =
        one slot of one wave, rows_per_wave / streams rows): equal rows and, within a fraction of a percent, equal
        edges per stream; `streams` neighbouring streams form a wave and advance together;
      * a stream lists its rows' edges slice by slice (then row by row, then in CSR order) as words
        (local row << 24) | column; streams of a wave are interleaved step by step and padded to the longest."""
    assert 1 <= slices <= 4096 and streams in (2, 4, 8) and waves_per_gen >= 1
    assert rows_per_wave % streams == 0 and rows_per_wave <= 128
    assert ncols < (1 << 24), "column ids share a 32-bit word with the local row and are multiplied in 24 bits"
    m = rowptr.numel() - 1
    dev = col.device
    i64 = dict(dtype=torch.int64, device=dev)
    nnz = col.numel()
    deg = rowptr[1:] - rowptr[:-1]
    nchunk = ((deg + chunk - 1) // chunk).clamp(min=1)
    nv = int(nchunk.sum())
    vrow = torch.repeat_interleave(torch.arange(m, **i64), nchunk)              # virtual row -> row
    first = torch.cumsum(nchunk, 0) - nchunk                                    # first virtual row of a row
    ci = torch.arange(nv, **i64) - first[vrow]                                  # chunk number inside the row
    vlen = (deg[vrow] - ci + nchunk[vrow] - 1) // nchunk[vrow]                  # edges ci, ci + nchunk, ci + 2 nchunk, ...
    erow = torch.repeat_interleave(torch.arange(m, **i64), deg)                 # edge -> row
    ev = first[erow] + (torch.arange(nnz, **i64) - rowptr[erow]) % nchunk[erow]  # edge -> virtual row
    del erow
    per = rows_per_wave // streams                              # rows of a stream
    gens = max(1, -(-nv // (waves_per_gen * rows_per_wave)))
    nw = gens * waves_per_gen
    ns = nw * streams
    # deal the virtual rows to the streams in rounds of one row per stream, longest rows first; in every round the
    # longest row goes to the stream that holds the fewest edges so far (a plain back-and-forth deal leaves 5 % between
    # the longest and the mean stream when hub pieces of 2048 edges meet streams of 7000)
    order = torch.sort(vlen, descending=True, stable=True).indices
    sid = torch.empty(nv, **i64)
    rnd = torch.empty(nv, **i64)
    loads = torch.zeros(ns, **i64)
    # experiment (scripts/exp_xcd_bias.py): workgroups on even XCDs run 1-2 % slower than the mean:
    # count an edge dealt to one of their streams as 1 + bias edges, so that they get fewer
    bias = float(os.environ.get("ISPLIB_EXP_XCD_BIAS", "0"))
    cost = None
    if bias != 0.0:
        blk = (torch.arange(ns, **i64) // streams % waves_per_gen) // 4          # workgroup of the stream inside its launch
        cost = torch.where(blk % 2 == 0, 1024 + int(round(1024 * bias)), 1024 - int(round(1024 * bias))).to(torch.int64)
    for r in range(-(-nv // ns)):
        items = order[r * ns:(r + 1) * ns]
        to = torch.sort(loads, stable=True).indices[:items.numel()]
        sid[items] = to
        rnd[items] = r
        loads[to] += vlen[items] if cost is None else vlen[items] * cost[to]
    wave, slot = sid // streams, sid % streams
    lrow = slot * per + rnd                                      # local row inside the wave
    is_hub = nchunk[vrow] > 1
    part = torch.where(is_hub, torch.cumsum(is_hub.to(torch.int64), 0) - 1, torch.full((nv,), -1, **i64))
    wave_row = torch.full((nw * rows_per_wave,), -1, dtype=torch.int32, device=dev)
    wave_part = torch.full((nw * rows_per_wave,), -1, dtype=torch.int32, device=dev)
    at = wave * rows_per_wave + lrow
    wave_row[at] = vrow.to(torch.int32)
    wave_part[at] = part.to(torch.int32)
    hub_rows = torch.nonzero(nchunk > 1).flatten()
    hub_off = torch.zeros(hub_rows.numel() + 1, dtype=torch.int32, device=dev)
    if hub_rows.numel():
        hub_off[1:] = torch.cumsum(nchunk[hub_rows], 0).to(torch.int32)
    # the edges, stream by stream: (stream, slice, row of the stream) then CSR order (stable sort)
    width = -(-ncols // slices)
    if os.environ.get("ISPLIB_EXP_STREAM_ORDER") == "col":
        # experiment (scripts/exp_colorder.py): inside a (stream, slice) group the words in COLUMN order instead of row by row --
        # the 32 streams of a CU then sweep a slice's columns together and popular rows of y are gathered by several of them
        # within a few steps of each other (L1 reuse?); the price is a change of accumulator row at nearly every step
        key = (sid[ev] * slices + col // width) * width + col % width
    elif pad_row is not None or os.environ.get("ISPLIB_EXP_SNAKE") == "1":
        # max / min plans (round 4; ISPLIB_EXP_SNAKE=1 forces it for an experiment on sum plans, scripts/exp_snake.py): a stream
        # walks its rows forwards in even slices and backwards in odd ones, so the last row of slice s is the first row of
        # slice s + 1 and that change of row -- an LDS swap in the max / min kernel -- disappears (one of `per` per slice);
        # every row's own word order is unchanged, so max / min results are bit-identical (K=64 1.794 -> 1.786 ms, K=32
        # 0.858 -> 0.845).  Sum / mean plans keep the plain order (0.5 % slower there, and the sums re-associate).
        sl = col // width
        key = (sid[ev] * slices + sl) * per + torch.where(sl % 2 == 1, per - 1 - rnd[ev], rnd[ev])
    else:
        key = (sid[ev] * slices + col // width) * per + rnd[ev]
    perm = torch.sort(key, stable=True).indices
    del key
    e_sid = sid[ev][perm]
    lens = torch.bincount(e_sid, minlength=ns)
    start = torch.cumsum(lens, 0) - lens
    p = torch.arange(nnz, **i64) - start[e_sid]                                  # position inside the stream
    steps = lens.view(nw, streams).max(dim=1).values
    wave_step_off = torch.zeros(nw + 1, **i64)
    wave_step_off[1:] = torch.cumsum(steps, 0)
    n_steps = int(wave_step_off[-1])
    idx = (wave_step_off[e_sid // streams] + p) * streams + e_sid % streams
    del p, start
    # padding: column n (reads 0 through the range check), row = the first row of the word's own stream (sum / mean) or,
    # for max / min plans -- where a 0 could win -- `pad_row`, the kernel's spare row
    if pad_row is None:
        pad = ((torch.arange(streams, **i64) * per) << 24) | int(ncols)
    else:
        pad = torch.full((streams,), (int(pad_row) << 24) | int(ncols), **i64)
    pad = torch.where(pad >= 2 ** 31, pad - 2 ** 32, pad)                        # the top byte may reach the sign bit
    words = pad.to(torch.int32).repeat(n_steps)
    words[idx] = ((lrow[ev][perm] << 24) | col[perm]).to(torch.int32)
    perm_out = torch.full((n_steps * streams,), -1, dtype=torch.int32 if nnz < 2 ** 31 else torch.int64, device=dev)
    perm_out[idx] = perm.to(perm_out.dtype)
    return dict(rows=m, cols=int(ncols), slices=slices, gens=gens, waves_per_gen=waves_per_gen, rows_per_wave=rows_per_wave,
                streams=streams, n_steps=n_steps, n_parts=int(is_hub.sum()), n_hub=int(hub_rows.numel()), words=words, vals=None,
                perm=perm_out, wave_step_off=wave_step_off, wave_row=wave_row, wave_part=wave_part,
                hub_row=hub_rows.to(torch.int32), hub_off=hub_off, chunk=chunk)


def build_stream_plan(rowptr: torch.Tensor, col: torch.Tensor, val: Optional[torch.Tensor], ncols: int, slices: int,
                      waves_per_gen: Optional[int] = None, rows_per_wave: Optional[int] = None, streams: int = 4,
                      chunk: int = 512, minmax: bool = False) -> Optional[StreamPlan]:
    """Stream plan of a graph on the device.  rows_per_wave / waves_per_gen default to what the kernel of this slot
    width is built for (isplib_spmm_stream_geometry; minmax: isplib_spmm_stream_minmax_geometry, whose plans are their
    own).  None when n >= 2^24.  (Rows need not be column-sorted: the stream order -- slice, row, CSR position -- is the
    plan's own.)"""
    if ncols >= (1 << 24):
        return None
    if minmax:
        # the kernel's tie rule (first strictly better candidate in stream order = lowest CSR position) holds for rows
        # whose columns ascend; anything else stays on the task list
        if col.numel() > 1:
            starts = torch.zeros(col.numel(), dtype=torch.bool, device=col.device)
            starts[rowptr[:-1][rowptr[:-1] < col.numel()]] = True
            if bool(((col[1:] < col[:-1]) & ~starts[1:]).any()):
                return None
        rpw, resident = cabi.stream_minmax_geometry(streams)
    else:
        rpw, resident = cabi.stream_geometry(streams)
    rows_per_wave = rpw if rows_per_wave is None else rows_per_wave
    waves_per_gen = resident if waves_per_gen is None else waves_per_gen
    plan = StreamPlan(**stream_plan_arrays(rowptr, col, ncols, slices, waves_per_gen, rows_per_wave, streams, chunk,
                                           pad_row=rows_per_wave if minmax else None))
    plan.set_values(val)
    return plan


def build_stream_plan_native(rowptr: torch.Tensor, col: torch.Tensor, ncols: int, slices: int, streams: int = 4,
                             chunk: int = 512, minmax: bool = False) -> Optional[StreamPlan]:
    """The same plan through the library's own builder (``isplib_stream_plan_build_hip`` / ``_minmax_hip``: rocPRIM sorts
    and HIP kernels, ~10 ms on the Reddit shape against 27-30 ms for the torch ops above; identical arrays,
    tests/test_gpu_sweep.py::test_native_stream_plan_equals_the_torch_built_one) -- what the plug-in uses.  The arrays
    stay in the library's allocation; the returned StreamPlan holds zero-copy views of them and the owning object.
    Unit weights (`set_values` gathers through `perm` as for a torch-built plan).  None where the builder declines
    (n >= 2^24, nnz >= 2^31, max / min on rows that are not column-sorted)."""
    if ncols >= (1 << 24) or col.numel() >= (1 << 31):
        return None
    try:
        native = cabi.NativeStreamPlan(rowptr, col, None, ncols, streams, slices, chunk, 0, minmax)
    except cabi.IsplibError as e:
        if e.status == cabi.ISPLIB_FAIL:          # outside the builder's domain / unsorted rows: the caller's other schedules
            return None
        if e.status == cabi.ISPLIB_NOT_ENOUGH_MEM:    # no room for the plan (3 x 4 B per edge): the task list / plain kernel
            torch.cuda.empty_cache()                  # serve the call, as the C handle's side_stream_plan does
            return None
        raise
    i32, i64 = torch.int32, torch.int64
    plan = StreamPlan(rows=native.rows, cols=native.cols, slices=native.slices, gens=native.gens, waves_per_gen=native.waves_per_gen,
                      rows_per_wave=native.rows_per_wave, streams=native.streams, n_steps=native.n_steps, n_parts=native.n_parts,
                      n_hub=native.n_hub, words=native.view("words", i32), vals=None, perm=native.view("perm", i32),
                      wave_step_off=native.view("wave_step_off", i64), wave_row=native.view("wave_row", i32),
                      wave_part=native.view("wave_part", i32), hub_row=native.view("hub_row", i32),
                      hub_off=native.view("hub_off", i32), chunk=chunk)
    plan._native = native            # owns the device arrays the tensors above look at
    return plan


@dataclass
class HybridPlan:
    """Plan of the hybrid form of the stream schedule (``fusedMM_csr_hybrid_hip``, include/isplib_hip.h:
    isplib_hybrid_plan): a stream plan of the COLD edges plus, per column slice, the table of the hottest rows of y and
    the hot edges as (local row, table row) words, chunked by (wave, slice)."""
    cold: StreamPlan
    table_rows: int
    hot_cap: int
    n_hot_steps: int
    hot_rows: torch.Tensor       # int32 [slices*table_rows]
    hot_words: torch.Tensor      # int32 [n_hot_steps*streams]
    hot_step_off: torch.Tensor   # int64 [gens*waves_per_gen*slices + 1]
    hot_perm: torch.Tensor       # int32 [n_hot_steps*streams], -1 = padding
    hot_edges: int = 0           # how many edges are served from the tables

    def struct(self) -> "cabi.HybridPlanStruct":
        p = lambda t: t.data_ptr() if t is not None and t.numel() else None  # noqa: E731
        return cabi.HybridPlanStruct(self.cold.struct(), self.table_rows, self.hot_cap, self.n_hot_steps, p(self.hot_rows),
                                     p(self.hot_words), p(self.hot_step_off), p(self.hot_perm))

    def workspace(self) -> torch.Tensor:
        return self.cold.workspace()


def hybrid_plan_arrays(rowptr: torch.Tensor, col: torch.Tensor, ncols: int, slices: int, waves_per_gen: int, rows_per_wave: int,
                       streams: int, chunk: int, table_rows: int, hot_cap: int, min_refs: int = 2) -> dict:
    """The arrays of a hybrid plan; plain torch ops on the device of `col` (once per graph and geometry).

      * Hot rows: per column slice, the table_rows - 1 columns with the most references (in-degree; at least
        `min_refs`), the same table for every workgroup; table row table_rows - 1 stays all zero (padding words point
        at it).  An edge is HOT when its column is in the table of its slice, else COLD.
      * Rows are dealt to streams exactly as in `stream_plan_arrays` (virtual rows for hub rows, rounds of one row per
        stream, longest first to the least loaded stream) but by their COLD length: the gather pipeline is what bounds the
        kernel, so that is what is balanced.
      * Cold edges: a stream plan like any other (`cold`), whose `slices` is also the number of phases.
      * Hot edges: per (wave, slice) a chunk of steps; step j holds word j of each of the wave's slots, (local row << 24)
        | table row, rows ascending then CSR order; a slot with fewer hot edges in the slice than the chunk's longest is
        padded with (its first row << 24) | table_rows - 1.  At most `hot_cap` steps per chunk (the kernel keeps a
        chunk's words in registers): hot edges beyond it stay cold."""
    assert 1 <= slices <= 4096 and streams in (4, 8) and waves_per_gen >= 8 and waves_per_gen % 8 == 0
    assert rows_per_wave % streams == 0 and rows_per_wave <= 256 and 2 <= table_rows <= 65536 and hot_cap >= 1
    assert ncols < (1 << 24)
    m = rowptr.numel() - 1
    dev = col.device
    i64 = dict(dtype=torch.int64, device=dev)
    nnz = col.numel()
    assert nnz < 2 ** 31
    per = rows_per_wave // streams
    width = -(-ncols // slices)
    # ---- the tables ----
    indeg = torch.bincount(col, minlength=ncols)
    cslice = torch.arange(ncols, **i64) // width
    top = int(indeg.max()) + 1 if ncols else 1
    order = torch.sort(cslice * top + (top - 1 - indeg), stable=True).indices        # slice ascending, in-degree descending
    s_sorted = cslice[order]
    counts = torch.bincount(cslice, minlength=slices)
    starts = torch.cumsum(counts, 0) - counts
    rank = torch.arange(ncols, **i64) - starts[s_sorted]
    is_top = (rank < table_rows - 1) & (indeg[order] >= min_refs)
    tidx = torch.full((ncols,), -1, **i64)
    tidx[order[is_top]] = rank[is_top]
    hot_rows = torch.full((slices * table_rows,), int(ncols), dtype=torch.int32, device=dev)
    hot_rows[s_sorted[is_top] * table_rows + rank[is_top]] = order[is_top].to(torch.int32)
    del order, s_sorted, rank, is_top, counts, starts, indeg, cslice
    e_hot = tidx[col] >= 0
    # ---- virtual rows and their streams (stream_plan_arrays, on cold lengths) ----
    deg = rowptr[1:] - rowptr[:-1]
    nchunk = ((deg + chunk - 1) // chunk).clamp(min=1)
    nv = int(nchunk.sum())
    vrow = torch.repeat_interleave(torch.arange(m, **i64), nchunk)
    first = torch.cumsum(nchunk, 0) - nchunk
    erow = torch.repeat_interleave(torch.arange(m, **i64), deg)
    ev = first[erow] + (torch.arange(nnz, **i64) - rowptr[erow]) % nchunk[erow]
    del erow
    vlen = torch.bincount(ev[~e_hot], minlength=nv)
    gens = max(1, -(-nv // (waves_per_gen * rows_per_wave)))
    nw = gens * waves_per_gen
    ns = nw * streams
    order = torch.sort(vlen, descending=True, stable=True).indices
    sid = torch.empty(nv, **i64)
    rnd = torch.empty(nv, **i64)
    loads = torch.zeros(ns, **i64)
    for r in range(-(-nv // ns)):
        items = order[r * ns:(r + 1) * ns]
        to = torch.sort(loads, stable=True).indices[:items.numel()]
        sid[items] = to
        rnd[items] = r
        loads[to] += vlen[items]
    del order, loads
    wave, slot = sid // streams, sid % streams
    lrow = slot * per + rnd
    is_hub = nchunk[vrow] > 1
    part = torch.where(is_hub, torch.cumsum(is_hub.to(torch.int64), 0) - 1, torch.full((nv,), -1, **i64))
    wave_row = torch.full((nw * rows_per_wave,), -1, dtype=torch.int32, device=dev)
    wave_part = torch.full((nw * rows_per_wave,), -1, dtype=torch.int32, device=dev)
    at = wave * rows_per_wave + lrow
    wave_row[at] = vrow.to(torch.int32)
    wave_part[at] = part.to(torch.int32)
    hub_rows = torch.nonzero(nchunk > 1).flatten()
    hub_off = torch.zeros(hub_rows.numel() + 1, dtype=torch.int32, device=dev)
    if hub_rows.numel():
        hub_off[1:] = torch.cumsum(nchunk[hub_rows], 0).to(torch.int32)
    e_sid_all = sid[ev]
    e_rnd_all = rnd[ev]
    e_lrow_all = lrow[ev]
    del ev
    # ---- hot edges: order inside (stream, slice) groups, the cap, the chunks ----
    h_e = torch.nonzero(e_hot).flatten()
    hkey = (e_sid_all[h_e] * slices + col[h_e] // width) * per + e_rnd_all[h_e]
    hperm = torch.sort(hkey, stable=True).indices
    h_e = h_e[hperm]
    group = hkey[hperm] // per                                   # stream * slices + slice
    del hkey, hperm
    gcount = torch.bincount(group, minlength=ns * slices)
    gstart = torch.cumsum(gcount, 0) - gcount
    hrank = torch.arange(h_e.numel(), **i64) - gstart[group]
    over = hrank >= hot_cap
    if bool(over.any()):
        e_hot[h_e[over]] = False                                 # beyond the cap: served by the gather path
        h_e, group, hrank = h_e[~over], group[~over], hrank[~over]
    hsteps = gcount.clamp(max=hot_cap).view(nw, streams, slices).max(dim=1).values      # [nw, slices]
    hot_step_off = torch.zeros(nw * slices + 1, **i64)
    hot_step_off[1:] = torch.cumsum(hsteps.flatten(), 0)
    n_hot_steps = int(hot_step_off[-1])
    h_sid = group // slices
    h_slice = group % slices
    widx = (hot_step_off[(h_sid // streams) * slices + h_slice] + hrank) * streams + h_sid % streams
    pad = ((torch.arange(streams, **i64) * per) << 24) | int(table_rows - 1)
    hot_words = pad.to(torch.int32).repeat(n_hot_steps)
    hot_words[widx] = ((e_lrow_all[h_e] << 24) | tidx[col[h_e]]).to(torch.int32)
    hot_perm = torch.full((n_hot_steps * streams,), -1, dtype=torch.int32, device=dev)
    hot_perm[widx] = h_e.to(torch.int32)
    n_hot = int(h_e.numel())
    del h_e, group, hrank, gcount, gstart, widx, h_sid, h_slice, tidx
    # ---- cold edges: the stream plan's construction on what is left ----
    c_e = torch.nonzero(~e_hot).flatten()
    c_sid = e_sid_all[c_e]
    key = (c_sid * slices + col[c_e] // width) * per + e_rnd_all[c_e]
    perm = torch.sort(key, stable=True).indices
    del key
    c_e, c_sid = c_e[perm], c_sid[perm]
    del perm
    lens = torch.bincount(c_sid, minlength=ns)
    start = torch.cumsum(lens, 0) - lens
    p = torch.arange(c_e.numel(), **i64) - start[c_sid]
    steps = lens.view(nw, streams).max(dim=1).values
    wave_step_off = torch.zeros(nw + 1, **i64)
    wave_step_off[1:] = torch.cumsum(steps, 0)
    n_steps = int(wave_step_off[-1])
    idx = (wave_step_off[c_sid // streams] + p) * streams + c_sid % streams
    del p, start
    pad = ((torch.arange(streams, **i64) * per) << 24) | int(ncols)
    words = pad.to(torch.int32).repeat(n_steps)
    words[idx] = ((e_lrow_all[c_e] << 24) | col[c_e]).to(torch.int32)
    perm_out = torch.full((n_steps * streams,), -1, dtype=torch.int32, device=dev)
    perm_out[idx] = c_e.to(torch.int32)
    cold = dict(rows=m, cols=int(ncols), slices=slices, gens=gens, waves_per_gen=waves_per_gen, rows_per_wave=rows_per_wave,
                streams=streams, n_steps=n_steps, n_parts=int(is_hub.sum()), n_hub=int(hub_rows.numel()), words=words, vals=None,
                perm=perm_out, wave_step_off=wave_step_off, wave_row=wave_row, wave_part=wave_part,
                hub_row=hub_rows.to(torch.int32), hub_off=hub_off, chunk=chunk)
    return dict(cold=cold, table_rows=table_rows, hot_cap=hot_cap, n_hot_steps=n_hot_steps, hot_rows=hot_rows, hot_words=hot_words,
                hot_step_off=hot_step_off, hot_perm=hot_perm, hot_edges=n_hot)


def build_hybrid_plan(rowptr: torch.Tensor, col: torch.Tensor, ncols: int, slices: int, streams: int = 4, chunk: int = 512,
                      waves_per_gen: Optional[int] = None, rows_per_wave: Optional[int] = None, table_rows: Optional[int] = None,
                      hot_cap: Optional[int] = None, min_refs: int = 2) -> Optional[HybridPlan]:
    """Hybrid plan of a graph on the device (unit weights); geometry defaults to what the kernel of this slot width is built
    for (isplib_spmm_hybrid_geometry).  None when n >= 2^24 or nnz >= 2^31."""
    if ncols >= (1 << 24) or col.numel() >= (1 << 31):
        return None
    rpw, resident, ht, cap = cabi.hybrid_geometry(streams)
    arrays = hybrid_plan_arrays(rowptr, col, ncols, slices, resident if waves_per_gen is None else waves_per_gen,
                                rpw if rows_per_wave is None else rows_per_wave, streams, chunk,
                                ht if table_rows is None else table_rows, cap if hot_cap is None else hot_cap, min_refs)
    arrays["cold"] = StreamPlan(**arrays["cold"])
    return HybridPlan(**arrays)
