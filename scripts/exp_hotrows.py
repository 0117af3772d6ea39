"""Experiment (timing only, results are WRONG by construction): how much gather time and how many L2 requests would the
stream schedule save if the H most-referenced rows of y of every column slice were served from LDS instead of through the
address pipeline?  The plan is built with those edges simply dropped, so the kernel issues exactly the gathers a hybrid
kernel would still issue; what the hybrid would add (LDS reads, re-staging per slice) is not in the number: this is the
upper bound of the gain.  usage: exp_hotrows.py [k] ; HOT=0,128,384,640"""
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
out = torch.empty((n, k), device=dev)
streams, slices, chunk = cabi.suggest_stream(n, n, nnz, k)
width = -(-n // slices)
indeg = torch.bincount(col, minlength=n)                       # how often a row of y is gathered
row = cabi.csr_row_ids(rowptr, nnz)
for hot in [int(v) for v in os.environ.get("HOT", "0,128,384,640").split(",")]:
    keep = torch.ones(nnz, dtype=torch.bool, device=dev)
    if hot > 0:
        is_hot = torch.zeros(n, dtype=torch.bool, device=dev)
        for s in range(slices):
            lo, hi = s * width, min(n, (s + 1) * width)
            top = torch.topk(indeg[lo:hi], min(hot, hi - lo)).indices + lo
            is_hot[top] = True
        keep = ~is_hot[col]
    kcol = col[keep].contiguous()
    krp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.bincount(row[keep], minlength=n), 0, out=krp[1:])
    plan = build_stream_plan(krp, kcol, None, n, slices, None, None, streams, chunk)
    ws = plan.workspace()
    msg = cabi.MSG_SPMM_SUM
    for _ in range(3):
        cabi.fusedMM_csr_stream_hip(msg, krp, kcol.numel(), plan, x, out, ws)
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(10):
        cabi.fusedMM_csr_stream_hip(msg, krp, kcol.numel(), plan, x, out, ws)
    e_.record()
    torch.cuda.synchronize()
    print(f"K={k} {slices} slices, {hot} hottest rows per slice out of the gather stream: {1 - kcol.numel() / nnz:.1%} of the edges, "
          f"{s_.elapsed_time(e_) / 10:.3f} ms for the remaining gathers", flush=True)
    del plan, ws, kcol, krp
