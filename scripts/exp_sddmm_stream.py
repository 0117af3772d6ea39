"""Experiment: dA = SDDMM(X, dY) on the Reddit-shaped graph through the stream plan (isplib_sddmm_stream_hip) against the task
list (isplib_sddmm_csr_tasks_hip, the round-1/2 path) and against the SpMM on the same plan.  usage: exp_sddmm_stream.py [k]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_stream_plan, build_task_plan

dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rowptr, col, n = synth.dataset_like("reddit", device=dev)
nnz = col.numel()
x = synth.features(n, k, device=dev)
g = synth.features(n, k, seed=5, device=dev)


def clock(fn, reps=10):
    for _ in range(3):
        fn()
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(reps):
        fn()
    e_.record()
    torch.cuda.synchronize()
    return s_.elapsed_time(e_) / reps


streams, slices, chunk = cabi.suggest_stream(n, n, nnz, k)
plan = build_stream_plan(rowptr, col, None, n, slices, None, None, streams, chunk)
ws = plan.workspace()
out = torch.empty((n, k), device=dev)
print(f"K={k} SpMM-sum on the stream plan ({streams} streams, {slices} slices): {clock(lambda: cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, x, out, ws)):.3f} ms", flush=True)
got = cabi.sddmm_stream(rowptr, nnz, plan, x, g)
again = cabi.sddmm_stream(rowptr, nnz, plan, x, g)
print("   SDDMM on the stream plan bitwise repeatable:", bool(torch.equal(got, again)), flush=True)
print(f"K={k} SDDMM on the stream plan: {clock(lambda: cabi.sddmm_stream(rowptr, nnz, plan, x, g)):.3f} ms", flush=True)
s_whole = int(cabi.lib().isplib_suggest_slices_whole_rows(n, n, nnz, k))
tplan = build_task_plan(rowptr, col, n, max(s_whole, 1))
ref = cabi.sddmm_tasks(rowptr, col, tplan, x, g)
print(f"K={k} SDDMM on the task list ({s_whole} slices): {clock(lambda: cabi.sddmm_tasks(rowptr, col, tplan, x, g)):.3f} ms", flush=True)
row = cabi.csr_row_ids(rowptr, nnz)
err = (got - ref).abs()
mag = torch.zeros(nnz, device=dev)
step = 1 << 22
for b in range(0, nnz, step):
    mag[b:b + step] = (x[col[b:b + step]].abs() * g[row[b:b + step]].abs()).sum(1)
print(f"   stream vs task list: max |diff| / (1e-5 sum|x||g|) = {float((err / (1e-5 * mag + 1e-30)).max()):.3f}", flush=True)
h = cabi.GraphHandle(rowptr, col, None, n)
print(f"   isplib_graph_sddmm (default: task list): {clock(lambda: h.sddmm(x, g)):.3f} ms", flush=True)
# (round 4 also timed isplib_graph_sddmm on the stream plan through tuning knob 11; the knob went with the code path in round 5)
h.close()
