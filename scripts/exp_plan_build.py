"""Experiment: one-off cost of the per-graph operands on the Reddit-shaped graph -- stream plan by the torch builder
(isplib_amd/plan.py) and by the native one (isplib_stream_plan_build_hip), task plan, transpose.  usage: exp_plan_build.py [k]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_stream_plan, build_task_plan

dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rowptr, col, n = synth.dataset_like("reddit", device=dev)
nnz = col.numel()
val = synth.edge_weights(nnz, device=dev)
streams, slices, chunk = cabi.suggest_stream(n, n, nnz, k)


def clock(fn, reps=3):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
        del r
    return out


print("torch builder, unit weights :", [f"{t:.1f}" for t in clock(lambda: build_stream_plan(rowptr, col, None, n, slices, None, None, streams, chunk))], "ms", flush=True)
print("torch builder, weights      :", [f"{t:.1f}" for t in clock(lambda: build_stream_plan(rowptr, col, val, n, slices, None, None, streams, chunk))], "ms", flush=True)
print("native builder, unit weights:", [f"{t:.1f}" for t in clock(lambda: cabi.NativeStreamPlan(rowptr, col, None, n, streams, slices, chunk))], "ms", flush=True)
print("native builder, weights     :", [f"{t:.1f}" for t in clock(lambda: cabi.NativeStreamPlan(rowptr, col, val, n, streams, slices, chunk))], "ms", flush=True)
nat = cabi.NativeStreamPlan(rowptr, col, val, n, streams, slices, chunk)
print("native set_values           :", [f"{t:.2f}" for t in clock(lambda: nat.set_values(val))], "ms", flush=True)
print("task plan (8 slices)        :", [f"{t:.1f}" for t in clock(lambda: build_task_plan(rowptr, col, n, 8))], "ms", flush=True)
print("csr2csc (weights)           :", [f"{t:.1f}" for t in clock(lambda: cabi.csr2csc(rowptr, col, val, n, want_perm=False))], "ms", flush=True)
# what the plug-in pays on the first call per graph: SparseStorage.stream_plan (round 3: the native builder behind zero-copy views)
import isplib_amd  # noqa: E402
from isplib_amd.plan import build_stream_plan_native  # noqa: E402
print("plug-in builder (native, views):", [f"{t:.1f}" for t in clock(lambda: build_stream_plan_native(rowptr, col, n, slices, streams, chunk))], "ms", flush=True)


def first_call():
    adj = isplib_amd.SparseTensor.from_csr(rowptr, col, None, (n, n), validate=False)
    return adj.storage.stream_plan(False, (streams, slices, chunk))


print("SparseStorage.stream_plan, first call on a fresh graph object:", [f"{t:.1f}" for t in clock(first_call)], "ms", flush=True)
x = synth.features(n, k, device=dev)
out = torch.empty((n, k), device=dev)
ws = nat.workspace()
print("one SpMM launch             :", [f"{t:.2f}" for t in clock(lambda: cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, nat, x, out, ws), 5)], "ms", flush=True)
