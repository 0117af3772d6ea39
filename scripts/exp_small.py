import sys, torch
sys.path.insert(0, "/root/repo")
import isplib_amd
from isplib_amd import synth
dev = torch.device("cuda:0")
for scale in (0.02, 0.05, 0.1, 0.3, 1.0):
    rp, cl, n = synth.dataset_like("reddit", device=dev, scale=scale)
    adj = isplib_amd.SparseTensor.from_csr(rp, cl, None, (n, n), validate=False)
    for k in (32, 128):
        t = isplib_amd.iSpLibPlugin.autotune(adj, k, "sum", candidates=(0, 1, 2, 4, 6, 8, 12, 16, 20, 24), reps=5)
        print(f"scale {scale} N={n} nnz={cl.numel()} K={k} X={n*k*4/2**20:.1f}MB avgdeg={cl.numel()/n:.0f}: " + " ".join(f"S{s}={v:.3f}" for s, v in t.items()), flush=True)
