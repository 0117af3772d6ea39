"""Experiment: the stream plan's words in column order inside a (stream, slice) group (ISPLIB_EXP_STREAM_ORDER=col) against
row by row.  usage: exp_colorder.py [k]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_stream_plan

dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rowptr, col, n = synth.dataset_like("reddit", device=dev)
nnz = col.numel()
x = synth.features(n, k, device=dev)
streams, slices, chunk = cabi.suggest_stream(n, n, nnz, k)
outs = {}
for order in ("row", "col"):
    os.environ["ISPLIB_EXP_STREAM_ORDER"] = order
    for sl in (slices, slices * 2):
        plan = build_stream_plan(rowptr, col, None, n, sl, None, None, streams, chunk)
        ws = plan.workspace()
        out = torch.empty((n, k), device=dev)
        for _ in range(3):
            cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, x, out, ws)
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(10):
            cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, x, out, ws)
        e_.record()
        torch.cuda.synchronize()
        outs[(order, sl)] = out
        print(f"K={k} words in {order} order inside a (stream, slice) group, {sl} slices: {s_.elapsed_time(e_) / 10:.3f} ms", flush=True)
        del plan, ws
ref = outs[("row", slices)]
print("max |col order - row order| :", float((outs[("col", slices)] - ref).abs().max()), flush=True)
