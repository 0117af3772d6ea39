"""SDDMM dA timing on the Reddit-shaped graph: plain kernel vs task plans of several slice counts (experiment helper).
The dot product needs whole rows of y, so -- unlike the SpMM -- it cannot run in column panels and wants the slice
count of the FULL width."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_task_plan
dev = torch.device("cuda:0")
rowptr, col, n = synth.dataset_like("reddit", device=dev)
nnz = col.numel()
for k, slist in ((64, (4, 8, 12)), (128, (8, 12, 16, 20)), (256, (16, 24, 32))):
    x = synth.features(n, k, device=dev); g = synth.features(n, k, seed=5, device=dev)
    runs = [("plain", lambda: cabi.sddmm(rowptr, col, x, g))]
    for S in slist:
        plan = build_task_plan(rowptr, col, n, S)
        runs.append((f"tasks S={S}", (lambda p: lambda: cabi.sddmm_tasks(rowptr, col, p, x, g))(plan)))
    for name, fn in runs:
        fn(); fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): fn()
        e.record(); torch.cuda.synchronize()
        print(f"K={k} sddmm {name}: {s.elapsed_time(e)/5:.3f} ms", flush=True)
