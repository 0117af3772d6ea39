"""Experiment: column-slice A so that each XCD's L2 caches only its slice of X.
Emulated with the plain kernel: virtual rows = (slice, row), slice-major; the kernel's
XCD remap then hands each XCD a contiguous range of slices."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth

dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rowptr, col, n = synth.dataset_like("reddit", device=dev)
nnz = col.numel()
x = synth.features(n, k, device=dev)
row = cabi.csr_row_ids(rowptr, nnz)
ref, _ = cabi.spmm(rowptr, col, None, x, "sum")

def timeit(fn, it=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it

out = torch.empty((n, k), device=dev)
t = timeit(lambda: cabi.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, x, out))
print(f"baseline: {t:.3f} ms  {nnz/t/1e6:.2f} Gedges/s", flush=True)
import ctypes
for S, lpr in ((8,0),(8,16),(8,8),(16,0),(16,16),(16,8),(1,16),(1,8)):
    cabi.lib().isplib_hip_tune(0, lpr)
    width = (n + S - 1) // S
    sl = col // width
    key = sl * n + row
    order = torch.sort(key, stable=True).indices
    colv = col[order].contiguous()
    rpv = torch.zeros(S * n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.bincount(key, minlength=S * n), 0, out=rpv[1:])
    outv = torch.empty((S * n, k), device=dev)
    t = timeit(lambda: cabi.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, rpv, colv, None, x, outv))
    t2 = timeit(lambda: torch.sum(outv.view(S, n, k), 0, out=out))
    err = (out - ref).abs().max().item()
    print(f"S={S:4d} lpr={lpr}: spmm {t:.3f} ms + reduce {t2:.3f} ms = {t+t2:.3f} ms  {nnz/(t+t2)/1e6:.2f} Gedges/s  maxerr {err:.2e}", flush=True)
    del outv, rpv, colv, order, key, sl
