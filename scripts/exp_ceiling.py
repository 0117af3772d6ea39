"""How fast is the task kernel when y fits the L2 entirely?  Same edge count and degrees as the Reddit shape, but the
columns index a y of only n rows (n*256 B at K=64): the L2-resident ceiling of this gather loop (experiment helper)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi
from isplib_amd.plan import build_task_plan
dev = torch.device("cuda:0")
m, deg, k = 232965, 492, 64
rowptr = torch.arange(0, (m + 1) * deg, deg, dtype=torch.int64, device=dev)
nnz = m * deg
for n in (2048, 8192, 16384, 65536, 232965):
    col = torch.randint(0, n, (m, deg), device=dev).sort(dim=1).values.reshape(-1).contiguous()
    x = torch.rand((n, k), device=dev)
    for S in ((1, 8) if n < 232965 else (1, 8, 12, 16, 24, 32)):
        plan = build_task_plan(rowptr, col, n, S, 1024, 128)
        work = plan.workspace("sum", k)
        out = torch.empty((m, k), device=dev)
        fn = lambda: cabi.fusedMM_csr_tasks_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, plan, x, out, None, work)
        fn(); fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): fn()
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) / 5
        print(f"y rows {n:7d} ({n * k * 4 / 2**20:6.1f} MB) S={S}: {t:.3f} ms  {nnz / t / 1e6:6.1f} Gedges/s  gather {nnz * k * 4 / t / 1e9:5.1f} TB/s  tasks {plan.n_tasks}", flush=True)
    del col, x
