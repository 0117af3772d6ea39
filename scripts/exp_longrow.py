"""Rounding of long rows of REPEATED terms (few distinct columns): error of each schedule and of the fp32 oracle against
the exact sum, in units of the 1e-5 * sum|a||x| tolerance (experiment helper)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from isplib_amd import cabi
from isplib_amd.plan import build_task_plan
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for deg, n, k in ((2903, 3, 100), (2903, 3, 128), (20000, 3, 64), (20000, 5000, 64)):
    col = np.sort(rng.integers(0, n, deg)).astype(np.int64)
    rowptr = np.array([0, deg], np.int64)
    x = (rng.random((n, k), dtype=np.float32) * 2 - 1)
    ref64 = x.astype(np.float64)[col].sum(0)
    tol = 1e-5 * np.abs(x).astype(np.float64)[col].sum(0)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_rowptr, d_col, d_x = t(rowptr), t(col), t(x)
    res = {}
    res["plain"] = cabi.spmm(d_rowptr, d_col, None, d_x, "sum")[0]
    table = cabi.spmm_slices(d_rowptr, d_col, n, 2)[0]
    res["sliced S=2"] = cabi.spmm_sliced(d_rowptr, d_col, None, table, 2, d_x, "sum")[0]
    for chunk in (1024, 256):
        plan = build_task_plan(d_rowptr, d_col, n, 2, chunk, 128)
        res[f"tasks S=2 chunk={chunk}"] = cabi.spmm_tasks(d_rowptr, d_col, None, plan, d_x, "sum")[0]
    orc, _ = oracle.spmm_fw(rowptr, col, np.ones(deg, np.float32), x, "sum")
    line = [f"oracle(fp32 sequential) {np.max(np.abs(orc[0] - ref64) / tol):.2f}"]
    for name, out in res.items():
        line.append(f"{name} {np.max(np.abs(out.cpu().numpy()[0] - ref64) / tol):.2f}")
    print(f"deg={deg} n={n} k={k}: err/tol  " + "  ".join(line), flush=True)
