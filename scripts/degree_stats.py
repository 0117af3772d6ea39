import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from isplib_amd import synth
dev = torch.device("cuda:0")
n, nnz = 232965, 114615892
for name, fn in (("rmat", synth.rmat_csr), ("uniform", synth.uniform_csr)):
    rp, col = fn(n, nnz, device=dev)
    d = (rp[1:] - rp[:-1])
    ds = torch.sort(d, descending=True).values
    tot = d.sum().item()
    print(name, "nnz", col.numel(), "max", ds[0].item(), "top10", ds[:10].tolist(), "rows>2048", (d > 2048).sum().item(), "rows>16384", (d > 16384).sum().item(),
          "edge share of rows>16384", (d[d > 16384].sum().item() / tot), "share rows>2048", d[d > 2048].sum().item() / tot, "empty", (d == 0).sum().item(), "median", d.median().item())
rp, col, _ = synth.dataset_like("reddit", device=dev)
d = rp[1:] - rp[:-1]
print("chunglu max", d.max().item(), "rows>2048", (d > 2048).sum().item(), "share", d[d > 2048].sum().item() / d.sum().item(), "median", d.median().item())
