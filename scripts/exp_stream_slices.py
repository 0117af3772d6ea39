"""Experiment: slice count (and hub-row chunk) of the stream schedule at width k on the Reddit-shaped graph.
usage: exp_stream_slices.py [k] ; SLICES=8,12,16,24,31"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_stream_plan

dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rowptr, col, n = synth.dataset_like("reddit", device=dev)
nnz = col.numel()
x = synth.features(n, k, device=dev)
out = torch.empty((n, k), device=dev)
streams, slices, chunk = cabi.suggest_stream(n, n, nnz, k)
print(f"K={k}: the rule says {streams} streams, {slices} slices, chunk {chunk}; geometry {cabi.stream_geometry(streams)}", flush=True)
for sl in [int(v) for v in os.environ.get("SLICES", "8,12,16,24,31,40").split(",")]:
    for ch in (chunk, chunk // 2):
        plan = build_stream_plan(rowptr, col, None, n, sl, None, None, streams, ch)
        ws = plan.workspace()
        for _ in range(3):
            cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, x, out, ws)
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(20):
            cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, x, out, ws)
        e_.record()
        torch.cuda.synchronize()
        print(f"K={k} {sl} slices, chunk {ch}, {plan.gens} generation(s), {plan.n_parts} virtual rows of hub rows: {s_.elapsed_time(e_) / 20:.3f} ms", flush=True)
        del plan, ws
