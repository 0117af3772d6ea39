"""How much of the stream kernel's time is the L2 misses?  The same edge count and degree structure, but every column id
folded into the first R rows of y (col % R): with R x 256 B inside one XCD's L2 every gather hits, and what is left is the
address-pipeline time of the instruction stream itself.  R = 4096 (1 MB of a 64-column panel: all hits), 16384 (4 MB: the
L2's size), 65536 (16 MB: Infinity Cache), and the real graph.  Results are wrong by construction for the folded graphs
(only the time counts).  usage: exp_stream_ceiling.py [k=128]"""
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
z = torch.empty((n, k), device=dev)
streams, slices, chunk = cabi.suggest_stream(n, n, nnz, k)
msg = cabi.MESSAGE["sum"]
for fold in (4096, 16384, 65536, 0):
    c = col if fold == 0 else (col % fold)
    for sl in ((slices,) if fold == 0 else (1, slices)):
        plan = build_stream_plan(rowptr, c, None, n, sl, None, None, streams, chunk)
        ws = plan.workspace()
        for _ in range(3):
            cabi.fusedMM_csr_stream_hip(msg, rowptr, nnz, plan, x, z, ws)
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(20):
            cabi.fusedMM_csr_stream_hip(msg, rowptr, nnz, plan, x, z, ws)
        e_.record()
        torch.cuda.synchronize()
        what = "the real graph" if fold == 0 else f"columns folded into the first {fold} rows of y ({fold * min(k, 256 // streams) * 4 / 2**20:.1f} MB of a panel)"
        print(f"K={k} {streams}:{sl}:{chunk} {what}: {s_.elapsed_time(e_) / 20:.3f} ms ({plan.gens} generations, {plan.n_steps} steps)", flush=True)
        del plan, ws
