"""Experiment: all generations of a stream pass in one launch (isplib_hip_tune(10, 1)) against one launch per generation.
usage: exp_merge_gens.py [k]"""
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
for sl in (slices, 24, 40):
    plan = build_stream_plan(rowptr, col, None, n, sl, None, None, streams, chunk)
    ws = plan.workspace()
    outs = []
    for merged in (0, 1, 0, 1):
        cabi.lib().isplib_hip_tune(10, merged)
        out = torch.empty((n, k), device=dev)
        for _ in range(3):
            cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, x, out, ws)
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(20):
            cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, x, out, ws)
        e_.record()
        torch.cuda.synchronize()
        outs.append(out)
        print(f"K={k} {sl} slices, generations merged into one launch: {merged}: {s_.elapsed_time(e_) / 20:.3f} ms", flush=True)
    print("   bitwise equal:", bool(torch.equal(outs[0], outs[1])), flush=True)
    cabi.lib().isplib_hip_tune(10, 0)
