"""Headline-loop experiments (one MI355X, Reddit shape): the sum / mean stream kernel at the rule's plan, K = 128 / 64 / 41 / 32,
unit and U(0,1) weights; prints the time and a checksum of the result's bits (variants of the loop must not change a bit).
usage: exp_sum_loop.py [cases, e.g. 128:u,128:w,64:u,41:u,32:u]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_stream_plan

dev = torch.device("cuda:0")
cases = (sys.argv[1] if len(sys.argv) > 1 else "128:u,128:w,64:u,64:w,41:u,32:u").split(",")
rowptr, col, n = synth.dataset_like("reddit", device=dev)
nnz = col.numel()
w = synth.edge_weights(nnz, device=dev)


def timeit(fn, it=20, warm=5):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


msg = cabi.MESSAGE["sum"]
for c in cases:
    k, weighted = int(c.split(":")[0]), c.endswith("w")
    x = synth.features(n, k, device=dev)
    z = torch.empty((n, k), device=dev)
    st, sl, ch = cabi.suggest_stream(n, n, nnz, k, weighted)
    plan = build_stream_plan(rowptr, col, w if weighted else None, n, sl, None, None, st, ch)
    ws = plan.workspace()
    cabi.fusedMM_csr_stream_hip(msg, rowptr, nnz, plan, x, z, ws)
    chk = int(z.view(torch.int32).to(torch.int64).sum())
    ts = [timeit(lambda: cabi.fusedMM_csr_stream_hip(msg, rowptr, nnz, plan, x, z, ws)) for _ in range(3)]
    print(f"[sum] K={k} {'w' if weighted else 'u'} plan {st}:{sl}:{ch}: " + " ".join(f"{t:.3f}" for t in ts) + f" ms  checksum {chk}", flush=True)
    del plan, ws, x, z
