"""One configuration of the slice experiment, for profiling: exp_one.py S lpr iters [k]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
S, lpr, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
k = int(sys.argv[4]) if len(sys.argv) > 4 else 128
dev = torch.device("cuda:0")
rowptr, col, n = synth.dataset_like("reddit", device=dev)
nnz = col.numel()
x = synth.features(n, k, device=dev)
cabi.lib().isplib_hip_tune(0, lpr)
if S > 1:
    row = cabi.csr_row_ids(rowptr, nnz)
    width = (n + S - 1) // S
    key = (col // width) * n + row
    order = torch.sort(key, stable=True).indices
    col = col[order].contiguous()
    rowptr = torch.zeros(S * n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.bincount(key, minlength=S * n), 0, out=rowptr[1:])
    del key, order, row
out = torch.empty((S * n, k), device=dev)
torch.cuda.synchronize()
for _ in range(iters):
    cabi.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, x, out)
torch.cuda.synchronize()
