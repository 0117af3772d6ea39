"""SDDMM dA timing on the Reddit-shaped graph: plain kernel vs task plan (experiment helper)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_task_plan
dev = torch.device("cuda:0")
rowptr, col, n = synth.dataset_like("reddit", device=dev)
nnz = col.numel()
for k, S in ((64, 8), (128, 16)):
    x = synth.features(n, k, device=dev); g = synth.features(n, k, seed=5, device=dev)
    plan = build_task_plan(rowptr, col, n, S)
    for name, fn in (("plain", lambda: cabi.sddmm(rowptr, col, x, g)), ("tasks", lambda: cabi.sddmm_tasks(rowptr, col, plan, x, g))):
        fn(); fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): fn()
        e.record(); torch.cuda.synchronize()
        print(f"K={k} S={S} sddmm {name}: {s.elapsed_time(e)/5:.3f} ms", flush=True)
