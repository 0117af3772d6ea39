"""Experiment: fewer edges for the waves of workgroups on even XCDs (ISPLIB_EXP_XCD_BIAS, torch plan builder only).
usage: exp_xcd_bias.py [k]"""
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
for bias in ("0", "0.01", "0.02", "-0.01", "-0.02", "0"):
    os.environ["ISPLIB_EXP_XCD_BIAS"] = bias
    plan = build_stream_plan(rowptr, col, None, n, slices, None, None, streams, chunk)
    ws = plan.workspace()
    out = torch.empty((n, k), device=dev)
    for _ in range(3):
        cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, x, out, ws)
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(30):
        cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, x, out, ws)
    e_.record()
    torch.cuda.synchronize()
    print(f"K={k} edges of even-XCD workgroups weighted 1 + {bias}: {s_.elapsed_time(e_) / 30:.3f} ms", flush=True)
    del plan, ws
