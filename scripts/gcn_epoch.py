#!/usr/bin/env python3
"""Config 5 of BASELINE.json: 2-layer GCN full-batch training on a Reddit-shaped graph through the
plug-in surface, epoch time mean/std -- the caller of the hot path, restated from the reference's
benchmark script tests/cpu/gcn-sparse.py:55-129 (model :58-68, Adam :79, epoch structure :84-92,
report :114-125).  PyG is not in this image, so GCNConv(cached=True, normalize=False) is restated as
    out = matmul(adj_t, x @ W) + b                       (aggregate AFTER the linear layer: K = 32, then C)
with `matmul` = isplib_amd.matmul, i.e. what torch_sparse.matmul becomes after iSpLibPlugin.patch_pyg().

    python scripts/gcn_epoch.py [--epochs 10] [--scale 1.0] [--hidden 32] [--model gcn|sage|gin] [--aggr sum|mean|max]

--model sage also runs on the 1-D row partition (N > 1 ranks: DistGraph.matmul(x, aggr) with autograd for sum / mean / max / min).
--model sage / gin restate the other two callers (tests/cpu/graphSAGE-sparse.py:65-78, tests/cpu/gin-sparse.py:59-78):
they aggregate at the INPUT width (use --features 608, the padded Reddit width), 5 SpMM per epoch (the
input features need no gradient, so layer 1 has no backward SpMM).
"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL peer mappings)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


class GCNConv(torch.nn.Module):
    def __init__(self, fin, fout, narrow=False):
        super().__init__()
        self.lin = torch.nn.Linear(fin, fout, bias=False)
        self.bias = torch.nn.Parameter(torch.zeros(fout))
        # --narrow: A (X W) = (A X) W, so aggregate on whichever side of the linear layer is narrower (PyG always
        # aggregates after it).  Layer 2 of the Reddit model then gathers 128-byte rows (K=32: 0.97 ms per SpMM)
        # instead of 164-byte ones that straddle two cache lines (K=41: 1.85 ms).
        self.aggregate_first = narrow and fin < fout

    def forward(self, x, adj_t, matmul):
        if self.aggregate_first:
            return self.lin(matmul(adj_t, x, "sum")) + self.bias
        return matmul(adj_t, self.lin(x), "sum") + self.bias


class Net(torch.nn.Module):
    def __init__(self, fin, hidden, classes, normalize=False, narrow=False):
        super().__init__()
        self.conv1, self.conv2 = GCNConv(fin, hidden, narrow), GCNConv(hidden, classes, narrow)
        self.normalize = normalize      # GCNConv(normalize=True) of tests/dist/gcn/pyg-sparse.py:61-62

    def forward(self, x, adj_t, matmul):
        if self.normalize:               # D^-1/2 (A+I) D^-1/2, bias and ReLU fused into the aggregation's fold kernel
            import isplib_amd
            x = isplib_amd.gcn_norm_matmul(adj_t, self.conv1.lin(x), self.conv1.bias, relu=True)
            x = F.dropout(x, training=self.training)
            return F.log_softmax(isplib_amd.gcn_norm_matmul(adj_t, self.conv2.lin(x), self.conv2.bias), dim=1)
        x = F.relu(self.conv1(x, adj_t, matmul))
        x = F.dropout(x, training=self.training)
        return F.log_softmax(self.conv2(x, adj_t, matmul), dim=1)


class SAGEConv(torch.nn.Module):
    """PyG SAGEConv(aggr=sum|mean, normalize=False): lin_l(aggregate(x)) + lin_r(x); aggregation at the INPUT width
    (608 on padded Reddit, tests/cpu/graphSAGE-sparse.py:65-78, dataset_loader.py:145-160)."""

    def __init__(self, fin, fout, aggr):
        super().__init__()
        self.lin_l, self.lin_r, self.aggr = torch.nn.Linear(fin, fout), torch.nn.Linear(fin, fout, bias=False), aggr

    def forward(self, x, adj_t, matmul):
        return self.lin_l(matmul(adj_t, x, self.aggr)) + self.lin_r(x)


class SAGENet(torch.nn.Module):
    def __init__(self, fin, hidden, classes, aggr):
        super().__init__()
        self.conv1, self.conv2 = SAGEConv(fin, hidden, aggr), SAGEConv(hidden, classes, aggr)

    def forward(self, x, adj_t, matmul):
        x = F.dropout(F.relu(self.conv1(x, adj_t, matmul)), training=self.training)
        return F.log_softmax(self.conv2(x, adj_t, matmul), dim=1)


class GINConv(torch.nn.Module):
    """PyG GINConv(nn), eps = 0: nn(x + sum_j x_j) (tests/cpu/gin-sparse.py:59-78)."""

    def __init__(self, fin, hidden):
        super().__init__()
        self.nn = torch.nn.Sequential(torch.nn.Linear(fin, hidden), torch.nn.ReLU(), torch.nn.Linear(hidden, hidden))

    def forward(self, x, adj_t, matmul):
        return self.nn(x + matmul(adj_t, x, "sum"))


class GINNet(torch.nn.Module):
    def __init__(self, fin, hidden, classes):
        super().__init__()
        self.conv1, self.bn1 = GINConv(fin, hidden), torch.nn.BatchNorm1d(hidden)
        self.conv2, self.bn2 = GINConv(hidden, hidden), torch.nn.BatchNorm1d(hidden)
        self.fc1, self.fc2 = torch.nn.Linear(hidden, hidden), torch.nn.Linear(hidden, classes)

    def forward(self, x, adj_t, matmul):
        x = self.bn1(self.conv1(x, adj_t, matmul))
        x = self.bn2(self.conv2(x, adj_t, matmul))
        return F.log_softmax(self.fc2(self.fc1(x).relu()), dim=1)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--model", choices=("gcn", "sage", "gin"), default="gcn")
    p.add_argument("--aggr", choices=("sum", "mean", "max", "min"), default="sum", help="SAGE aggregation")
    p.add_argument("--epochs", type=int, default=10)
    p.add_argument("--scale", type=float, default=1.0)
    p.add_argument("--hidden", type=int, default=32)        # EMBEDDING_SIZE, tests/cpu/gcn-sparse.py:4
    p.add_argument("--features", type=int, default=602)     # Reddit, tests/cpu/dataset_tester.ipynb:496
    p.add_argument("--classes", type=int, default=41)
    p.add_argument("--normalize", action="store_true", help="GCN symmetric normalisation with self loops, fused (1 GPU)")
    p.add_argument("--narrow", action="store_true", help="GCN: aggregate on the narrower side of each linear layer")
    p.add_argument("--workload", choices=("reddit", "cora"), default="reddit",
                   help="graph shape; cora (N=2,708, config 1) is launch-bound: use --hipgraph; pass --features 1433 --classes 7")
    p.add_argument("--hipgraph", action="store_true",
                   help="capture the whole epoch (2 forwards, backward, Adam) in one hipGraph and replay it (1 GPU)")
    a = p.parse_args()
    torch.manual_seed(0)                                    # tests/cpu/gcn-sparse.py:10-12
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("ISPLIB_BENCH_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local if backend == "nccl" else local % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    if world > 1:       # torchrun --nproc-per-node N scripts/gcn_epoch.py: 1-D row partition, one process per GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    import isplib_amd
    from isplib_amd import synth
    rowptr, col, n = synth.dataset_like(a.workload, device=dev, scale=a.scale)
    nnz = col.numel()
    x = synth.features(n, a.features, device=dev)
    y = torch.randint(0, a.classes, (n,), device=dev)
    train_mask = torch.rand(n, device=dev) < 0.66
    n_train = int(train_mask.sum())
    if a.model == "gin" and world > 1:
        raise SystemExit("--model gin: single GPU only (BatchNorm statistics are not synchronised here)")
    if a.model == "sage":
        model = SAGENet(a.features, a.hidden, a.classes, a.aggr).to(dev)
    elif a.model == "gin":
        model = GINNet(a.features, a.hidden, a.classes).to(dev)
    else:
        model = Net(a.features, a.hidden, a.classes, a.normalize and world == 1, a.narrow).to(dev)   # same seed on every rank
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4, capturable=a.hipgraph)
    if a.hipgraph and world > 1:
        raise SystemExit("--hipgraph: single GPU only")
    if world > 1:
        from isplib_amd.dist import DistGraph
        adj_t = DistGraph(rowptr, col, None, n, rank, world)
        r0, r1 = adj_t.row0, adj_t.row0 + adj_t.rows
        x, y, train_mask = x[r0:r1].contiguous(), y[r0:r1], train_mask[r0:r1]
        del rowptr, col

        def matmul(g, m, reduce):                    # every reduction has autograd under the partition (round 5)
            return g.matmul(m, reduce)
    else:
        adj_t = isplib_amd.SparseTensor.from_csr(rowptr, col, None, (n, n), validate=False)
        isplib_amd.iSpLibPlugin.patch_pyg()
        matmul = isplib_amd.matmul
    times, losses = [], []
    if a.hipgraph:
        # The boundary calls neither allocate nor synchronise, so a whole epoch is capturable once the per-graph
        # operands exist (built by the warm-up epochs).  Boolean-mask indexing would synchronise: use index lists.
        idx = train_mask.nonzero().squeeze(1)
        y_train = y[idx]

        def epoch_body():
            model.train()
            opt.zero_grad(set_to_none=False)
            out = model(x, adj_t, matmul)
            loss = F.nll_loss(out[idx], y_train, reduction="sum") / n_train
            loss.backward()
            opt.step()
            pred = model(x, adj_t, matmul).argmax(1)
            return loss.detach(), (pred[idx] == y_train).float().sum()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                epoch_body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            g_loss, g_hit = epoch_body()
        for epoch in range(a.epochs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            graph.replay()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            losses.append(float(g_loss))
        acc = float(g_hit) / n_train
    for epoch in range(0 if a.hipgraph else a.epochs + 1):  # epoch 0 builds the per-graph operands; not timed
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.train()
        opt.zero_grad()
        out = model(x, adj_t, matmul)
        loss = F.nll_loss(out[train_mask], y[train_mask], reduction="sum") / n_train
        loss.backward()
        if world > 1:                                       # replicated dense weights: sum the shard gradients
            for prm in model.parameters():
                dist.all_reduce(prm.grad)
            dist.all_reduce(loss.detach_())
        opt.step()
        pred = model(x, adj_t, matmul).argmax(1)            # second forward, as at :89
        hit = (pred[train_mask] == y[train_mask]).float().sum()
        if world > 1:
            dist.all_reduce(hit)
        acc = float(hit) / n_train
        torch.cuda.synchronize()
        if epoch:
            times.append(time.perf_counter() - t0)
            losses.append(float(loss.detach()))
    if world == 1:
        isplib_amd.iSpLibPlugin.unpatch_pyg()
    if world > 1:
        t = torch.tensor([statistics.mean(times)], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        times = [float(t)] * len(times)
    if rank == 0:
        name = {"gcn": "GCN", "sage": f"SAGE({a.aggr})", "gin": "GIN"}[a.model]
        print(json.dumps({"workload": f"2-layer {name} {a.features}->{a.hidden}->{a.classes}, {a.workload}-like N={n} nnz={nnz}"
                                      + (", whole epoch replayed from one hipGraph" if a.hipgraph else "")
                                      + (", aggregation on the narrower side of each linear layer" if a.narrow else ""),
                          "epochs": a.epochs, "epoch_ms_mean": statistics.mean(times) * 1e3,
                          "epoch_ms_std": statistics.pstdev(times) * 1e3, "first_loss": losses[0],
                          "last_loss": losses[-1], "train_acc": acc, "spmm_calls_per_epoch": 6 if a.model == "gcn" else 5, "n_gpus": world, "normalize": bool(a.normalize and world == 1),
                          "backend": backend if world > 1 else None}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
