"""Reference-schema operators (what iSpLib's own Python calls) at Reddit scale: forward + backward time per call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isplib_amd  # noqa: F401
from isplib_amd import cabi, synth
dev = torch.device("cuda:0")
rowptr, col, n = synth.dataset_like("reddit", device=dev)
val = torch.ones(col.numel(), device=dev)                      # isplib/__init__.py:51-57 materialises unit weights
colptr, perm, row_t, val_t = cabi.csr2csc(rowptr, col, val, n)
row = cabi.csr_row_ids(rowptr, col.numel())
ops = torch.ops.isplib
for k in (32, 128):
    x = synth.features(n, k, device=dev).requires_grad_(True)
    g = synth.features(n, k, seed=5, device=dev)
    def step():
        out = ops.fusedmm_spmm(row, rowptr, col, val, colptr, perm, x, val_t, row_t)
        out.backward(g)
    step(); step()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): step()
    e.record(); torch.cuda.synchronize()
    print(f"K={k}: fusedmm_spmm forward+backward {s.elapsed_time(e)/5:.3f} ms  (handles cached: {ops.graph_cache_size()})", flush=True)
