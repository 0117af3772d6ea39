"""Per-call cost of the plug-in on a Cora-shaped graph (config 1: N=2,708, nnz=10,556, K=16): where do the microseconds go?
A launch this small is all overhead: Python dispatch in spmm_autotuned, the operator layer, the C ABI, the kernel launch."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isplib_amd
from isplib_amd import cabi, synth

dev = torch.device("cuda:0")
rowptr, col, n = synth.dataset_like("cora", device=dev)
nnz = col.numel()
adj = isplib_amd.SparseTensor.from_csr(rowptr, col, None, (n, n), validate=False)


def per_call(fn, reps=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for k in (16, 64):
    x = synth.features(n, k, device=dev)
    xg = x.clone().requires_grad_(True)
    out = torch.empty((n, k), device=dev)
    print(f"K={k}", flush=True)
    print(f"   torch elementwise (x + 1), for scale:          {per_call(lambda: x + 1.0):7.1f} us / call")
    print(f"   C ABI fusedMM_csr_hip through ctypes:           {per_call(lambda: cabi.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, x, out)):7.1f} us / call")
    for red in ("sum", "mean", "max"):
        print(f"   isplib_amd.matmul(adj, x, '{red}'), no grad:       {per_call(lambda: isplib_amd.matmul(adj, x, red)):7.1f} us / call")
    print(f"   isplib_amd.matmul(adj, x, 'sum'), requires_grad: {per_call(lambda: isplib_amd.matmul(adj, xg, 'sum')):7.1f} us / call")

    def fb():
        xg.grad = None
        isplib_amd.matmul(adj, xg, "sum").sum().backward()
    print(f"   forward + backward (sum):                       {per_call(fb, 500):7.1f} us / call")
    csr = torch.sparse_csr_tensor(rowptr, col, torch.ones(nnz, device=dev), size=(n, n))
    print(f"   torch.sparse.mm (rocSPARSE), for comparison:    {per_call(lambda: torch.sparse.mm(csr, x)):7.1f} us / call")
