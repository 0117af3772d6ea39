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
