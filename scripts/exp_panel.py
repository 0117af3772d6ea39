"""Does a 64-column pass over the upper half of 512-byte rows cost more than one over the lower half? (experiment helper)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_task_plan
dev = torch.device("cuda:0")
rowptr, col, n = synth.dataset_like("reddit", device=dev)
plan = build_task_plan(rowptr, col, n, 8)
work = plan.workspace("sum", 64)
for ld in (128, 160, 192, 256):
    x = synth.features(n, ld, device=dev)
    out = torch.empty((n, ld), device=dev)
    for c0 in range(0, min(ld, 256), 32):
        if c0 + 64 > ld: break
        xv, ov = x[:, c0:c0 + 64], out[:, c0:c0 + 64]
        fn = lambda: cabi.fusedMM_csr_tasks_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, plan, xv, ov, None, work)
        fn(); fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        print(f"ld={ld} columns [{c0},{c0+64}): {s.elapsed_time(e)/10:.3f} ms", flush=True)
