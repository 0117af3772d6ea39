"""Effect of the leading dimension of X on ragged widths (experiment helper): K=41 rows packed (164 B) vs padded to 48 / 64 floats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_task_plan
dev = torch.device("cuda:0")
rowptr, col, n = synth.dataset_like("reddit", device=dev)
nnz = col.numel()
for k, S in ((41, 5), (41, 8), (100, 12), (100, 16)):
    plan = build_task_plan(rowptr, col, n, S)
    work = plan.workspace("sum", k)
    out = torch.empty((n, k), device=dev)
    for ld in (k, (k + 15) // 16 * 16, (k + 31) // 32 * 32):
        xp = torch.zeros((n, ld), device=dev)
        xp[:, :k] = synth.features(n, k, device=dev)
        x = xp[:, :k]
        fn = lambda: cabi.fusedMM_csr_tasks_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, plan, x, out, None, work)
        fn(); fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): fn()
        e.record(); torch.cuda.synchronize()
        print(f"K={k} S={S} ldy={ld}: {s.elapsed_time(e)/5:.3f} ms", flush=True)
