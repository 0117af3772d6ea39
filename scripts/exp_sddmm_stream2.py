"""Experiment: dA = SDDMM(X, dY), K=128, Reddit shape, on stream plans of 128-column slots (2 streams: ONE pass, no accumulating
second panel) against 64-column slots (two passes) and the task list.  usage: exp_sddmm_stream2.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_stream_plan, build_task_plan

dev = torch.device("cuda:0")
k = 128
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


s_whole = int(cabi.lib().isplib_suggest_slices_whole_rows(n, n, nnz, k))
tplan = build_task_plan(rowptr, col, n, max(s_whole, 1))
ref = cabi.sddmm_tasks(rowptr, col, tplan, x, g)
print(f"K={k} SDDMM on the task list ({s_whole} slices): {clock(lambda: cabi.sddmm_tasks(rowptr, col, tplan, x, g)):.3f} ms", flush=True)
del tplan
mag = cabi.sddmm_tasks(rowptr, col, build_task_plan(rowptr, col, n, max(s_whole, 1)), x.abs(), g.abs())
for streams, slices in ((4, 31), (2, 31), (2, 48), (2, 63), (2, 80)):
    plan = build_stream_plan(rowptr, col, None, n, slices, None, None, streams, 2057)
    got = cabi.sddmm_stream(rowptr, nnz, plan, x, g)
    err = float(((got - ref).abs() / (1e-5 * mag + 1e-30)).max())
    print(f"K={k} SDDMM on a stream plan of {streams} streams, {slices} slices ({plan.gens} generation(s)): "
          f"{clock(lambda: cabi.sddmm_stream(rowptr, nnz, plan, x, g)):.3f} ms   max |diff| / (1e-5 sum|x||g|) = {err:.3f}", flush=True)
    del plan
