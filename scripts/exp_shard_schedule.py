"""Experiment: which schedule serves ONE rank's shard of the row-partitioned Reddit-shaped graph best?  At 8 ranks a shard has
~29 K rows and 14 M edges over all 233 K rows of X: the stream rule's reuse test (edges per generation and XCD >= 8 x rows of
X) says 7.7 and declines by a hair.  Times the local SpMM (no exchange) on the task list, the plain kernel and stream plans
of a few geometries for shards of world = 8 and 4.  usage: exp_shard_schedule.py [k]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.dist import RowPartition
from isplib_amd.plan import build_stream_plan, build_task_plan
from isplib_amd.plugin import suggest_slices

dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rowptr, col, n = synth.dataset_like("reddit", device=dev)
x = synth.features(n, k, device=dev)


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


for world in tuple(int(v) for v in os.environ.get("WORLDS", "8,4").split(",")):
    for rank in ((0, world // 2, world - 1) if world <= 8 else (0, world // 2)):
        part = RowPartition(rowptr, col, None, n, rank, world)
        buf = part.gather_buffer(k)
        for p in range(world):                                  # what the all-gather would deliver
            r0, r1 = part.x_cuts[p], part.x_cuts[p + 1]
            buf[p * part.max_rows: p * part.max_rows + (r1 - r0)] = x[r0:r1]
        out = torch.empty((part.rows, k), device=dev)
        res = [f"world {world} rank {rank}: {part.rows} rows, {part.nnz} edges, rule says {cabi.suggest_stream(part.rows, part.ncols_padded, part.nnz, k)}"]
        res.append(f"plain {timeit(lambda: cabi.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, part.rowptr, part.col_padded, None, buf, out)):.3f}")
        s = suggest_slices(part.rows, part.ncols_padded, part.nnz, k)
        if s > 0:
            tp = build_task_plan(part.rowptr, part.col_padded, part.ncols_padded, s)
            tw = tp.workspace("sum", k)
            res.append(f"tasks(S={s}) {timeit(lambda: cabi.fusedMM_csr_tasks_hip(cabi.MSG_SPMM_SUM, part.rowptr, part.col_padded, None, tp, buf, out, None, tw)):.3f}")
        for slices in (16, 32, 48):
            rpw, wpg = cabi.stream_geometry(4)
            chunk = max(256, int(part.nnz / (wpg * 4) / 3.4))
            sp = build_stream_plan(part.rowptr, part.col_padded, None, part.ncols_padded, slices, None, None, 4, chunk)
            ws = sp.workspace()
            res.append(f"stream(S={slices}, gens={sp.gens}) {timeit(lambda: cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, part.rowptr, part.nnz, sp, buf, out, ws)):.3f}")
        print(" | ".join(res), flush=True)
