"""How much of a rank's partitioned GCN epoch at P = 8 is launch overhead?  The emulated rank epoch of bench.py (DistGraph with the
all-gathers replaced by local fills, no RCCL) run eagerly and replayed from ONE hipGraph (forward, loss, backward, Adam, second
forward captured together).  usage: exp_epoch_emulated_graph.py [ranks=8] [rank=0]"""
import importlib.util
import os
import statistics
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from isplib_amd import synth
from isplib_amd.dist import DistGraph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gcn_epoch", os.path.join(ROOT, "scripts", "gcn_epoch.py"))
ge = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ge)
dev = torch.device("cuda:0")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rowptr, col, n = synth.dataset_like("reddit", device=dev)
torch.manual_seed(0)
feats, hidden, classes = 602, 32, 41
x = synth.features(n, feats, device=dev)
y = torch.randint(0, classes, (n,), device=dev)
mask = torch.rand(n, device=dev) < 0.66
n_train = int(mask.sum())


def fill(x_shard, buf):
    buf.view(-1, x_shard.size(0), x_shard.size(1))[:] = x_shard.unsqueeze(0)
    return None


g = DistGraph(rowptr, col, None, n, rank, world)
g.fwd.all_gather = fill
g.bwd.all_gather = fill
r0, r1 = g.row0, g.row0 + g.rows
xr, yr, mr = x[r0:r1].contiguous(), y[r0:r1], mask[r0:r1]
idx = mr.nonzero().squeeze(1)                      # index lists: boolean-mask indexing would synchronise inside a capture
y_train = yr[idx]
mm = lambda gg, m_, red: gg.matmul(m_, red)  # noqa: E731


def make(capturable):
    torch.manual_seed(0)
    model = ge.Net(feats, hidden, classes).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4, capturable=capturable)

    def body():
        model.train()
        opt.zero_grad(set_to_none=False)
        out = model(xr, g, mm)
        loss = F.nll_loss(out[idx], y_train, reduction="sum") / n_train
        loss.backward()
        opt.step()
        pred = model(xr, g, mm).argmax(1)
        return loss.detach(), pred
    return body


eager = make(False)
times = []
for epoch in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eager()
    torch.cuda.synchronize()
    if epoch > 1:
        times.append(time.perf_counter() - t0)
print(f"P={world} rank {rank} ({g.rows} rows, {g.fwd.nnz} edges): eager epoch {statistics.mean(times) * 1e3:.3f} ms (std {statistics.pstdev(times) * 1e3:.3f})", flush=True)

body = make(True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    g_loss, g_pred = body()
times = []
for epoch in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    graph.replay()
    torch.cuda.synchronize()
    if epoch > 1:
        times.append(time.perf_counter() - t0)
print(f"P={world} rank {rank}: the same epoch replayed from one hipGraph {statistics.mean(times) * 1e3:.3f} ms (std {statistics.pstdev(times) * 1e3:.3f}); loss {float(g_loss):.4f}", flush=True)
