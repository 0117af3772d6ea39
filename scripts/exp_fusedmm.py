"""Generic FusedMM pipeline timing at Reddit scale (experiment helper)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isplib_amd
from isplib_amd import synth
dev = torch.device("cuda:0")
rowptr, col, n = synth.dataset_like("reddit", device=dev)
adj = isplib_amd.SparseTensor.from_csr(rowptr, col, None, (n, n), validate=False)
nnz = col.numel()
for k in (32, 128):
    x = synth.features(n, k, device=dev) / k ** 0.5
    y = synth.features(n, k, seed=5, device=dev) / k ** 0.5
    for pat in ("spmm", "sigmoid_embedding", "tdist_embedding"):
        fn = lambda: isplib_amd.fusedmm(adj, x, y, pat)
        fn(); fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): fn()
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) / 3
        print(f"K={k} {pat}: {t:.3f} ms  {nnz / t / 1e6:.2f} Gedges/s", flush=True)
