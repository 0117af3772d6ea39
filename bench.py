#!/usr/bin/env python3
"""bench.py -- edges-aggregated/sec of SpMM-sum on a Reddit-shaped graph, K=128, fp32.

A "step" is one forward pass of the hot path (out = A @ X through the C ABI
``fusedMM_csr_hip``) over the whole graph, operands resident in HBM.
N = 1: the whole graph on one MI355X.
N > 1: the adjacency is 1-D row-partitioned (balanced by nnz) over the ranks;
       a step is ONE RCCL all-gather of X plus the local SpMM, so total work is
       fixed ("strong" scaling) and value = nnz(whole graph) / max-over-ranks time.

Prints ONE JSON line (rank 0).  ``roofline.achieved`` = algorithmic bytes per
launch (reference dtypes) / average kernel time from HIP
events recorded on the stream the kernel runs on.  ``cpu_baseline`` = the
oracle (restated FusedMM-semantics CPU kernel, OpenMP) on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# this pool's driver only supports dmabuf IPC: without it RCCL's peer mappings fail (hipIpcGetMemHandle)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--workload", default="reddit", choices=["cora", "reddit", "products"])
    p.add_argument("--generator", default="chunglu", choices=["chunglu", "rmat", "uniform"],
                   help="graph model at the workload's N and nnz (rmat / uniform: locality best / worst case)")
    p.add_argument("--k", type=int, default=None, help="feature width (default: 128 reddit, 16 cora, 256 products)")
    p.add_argument("--reduce", default="sum", choices=["sum", "mean", "max", "min"])
    p.add_argument("--scale", type=float, default=1.0, help="shrink the graph (debug only; result is then not the metric)")
    p.add_argument("--weighted", action="store_true", help="U(0,1) edge weights instead of unit weights")
    p.add_argument("--slices", type=int, default=-1,
                   help="column slices (0 = plain row kernel; -1 = isplib_amd.plugin.suggest_slices)")
    p.add_argument("--schedule", default="auto", choices=["auto", "stream", "sliced", "tasks"],
                   help="stream: rows resident in LDS, the plan's own edge stream (sum / mean); tasks: explicit task list; "
                        "sliced: (row, slice) segment per wave; auto: stream where isplib_suggest_stream says so, else tasks")
    p.add_argument("--stream-geom", default="", help="debug: streams:slices:chunk for the stream schedule instead of the rule")
    p.add_argument("--chunk", type=int, default=1024, help="tasks: edges per task")
    p.add_argument("--short", type=int, default=128, help="tasks: rows shorter than this are not sliced")
    p.add_argument("--tune", default="", help="debug: comma list of key=value for isplib_hip_tune")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extra", action="store_true", help="skip the other BASELINE.json configs (the `extra` array)")
    p.add_argument("--no-backward", action="store_true")
    p.add_argument("--only", default="", help="profiling: run ONE configuration of the `extra` array, named by its key in "
                   "profiles/traffic.json (reddit-{mean,max,min}-k64-weighted, reddit-sum-k128-weighted, reddit-sddmm-k128, "
                   "products-{chunglu,sbm}-sum-k256-{plain,ordered}), and print it instead of the metric line")
    return p.parse_args()


def cpu_baseline(rowptr, col, x, nnz):
    """The oracle's SpMM-sum on the host cores (oracle/fusedmm_oracle.c: oracle_spmm_sum_timed): the same inner loop as
    the parity oracle, run the way a careful OpenMP host would -- one contiguous nnz-balanced row block per thread,
    index / value / output streams first-touched by the thread that streams them, the gathered operand spread over
    all memory controllers.  Whole workload, 5 passes (about 5-10 s of CPU work), median.  For context: the oracle's
    plain `schedule(dynamic,16)` entry on NumPy's single-node arrays, and torch.sparse.mm on a CSR tensor (MKL)."""
    import numpy as np
    import oracle
    oracle.build()
    rp, cl, xx = rowptr.cpu().numpy(), col.cpu().numpy(), x.cpu().numpy()
    val = np.ones(cl.size, np.float32)   # the reference materialises unit weights (isplib/__init__.py:51-57)
    secs, _ = oracle.spmm_sum_timed(rp, cl, val, xx, reps=5)
    t = float(np.median(secs))
    t0 = time.perf_counter()
    oracle.spmm_fw(rp, cl, val, xx, "sum")
    plain_ms = (time.perf_counter() - t0) * 1e3
    torch_ms = None
    try:
        csr = torch.sparse_csr_tensor(torch.from_numpy(rp), torch.from_numpy(cl), torch.from_numpy(val), size=(rp.size - 1, xx.shape[0]))
        xt = torch.from_numpy(xx)
        torch.sparse.mm(csr, xt)
        tt = []
        for _ in range(2):
            t0 = time.perf_counter()
            torch.sparse.mm(csr, xt)
            tt.append(time.perf_counter() - t0)
        torch_ms = min(tt) * 1e3
    except Exception:  # noqa: BLE001 - context figure only
        pass
    k = xx.shape[1]
    return {"value": nnz / t, "unit": "edges/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": f"whole workload, 5 passes, median {t * 1e3:.1f} ms/pass (min {secs.min() * 1e3:.1f}); oracle/fusedmm_oracle.c "
                      "oracle_spmm_sum_timed (restated FusedMM-semantics kernel, -O3 -march=native -fopenmp, static nnz-balanced "
                      "row blocks, NUMA first touch)",
            "ms_per_step": t * 1e3, "gathered_GBps": nnz * (12 + 4 * k) / t / 1e9, "host_cpus": os.cpu_count(),
            "oracle_dynamic16_single_node_ms": plain_ms, "torch_sparse_mm_ms": torch_ms}


def _time_launches(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def extra_configs(dev, rowptr, col, n, with_cpu_epoch=True):
    """The other configurations of BASELINE.json under the same clock as the headline (N = 1, outside its timed region):
    config 3 (Reddit-shaped mean / max / min, K=64, weighted), config 2 with weights, config 4's shape on one GPU
    (ogbn-products-shaped, K=256) and config 5 (2-layer GCN epoch through the plug-in) with the oracle-aggregated CPU
    epoch beside it.  Each entry: ms per launch (HIP events, 5 launches), edges/s, roofline fraction from its own
    algorithmic bytes, and the schedule that ran."""
    import numpy as np
    from isplib_amd import cabi, synth
    from isplib_amd.plan import build_stream_plan, build_task_plan
    from isplib_amd.plugin import skew_adjusted, suggest_slices
    out = []
    nnz = col.numel()
    col32 = cabi.pack_indices(col)
    w = synth.edge_weights(nnz, device=dev)

    try:
        measured = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:  # noqa: BLE001
        measured = {}

    def entry(name, ms, m, nn, e, k, with_arg, schedule, key=None, **more):
        """`key`: this configuration's entry of profiles/traffic.json (fabric-side bytes per launch from separate rocprofv3
        --pmc passes of `bench.py --only <key>`, and the kernel-trace file its launch time can be recomputed from)."""
        b_alg = synth.algorithmic_bytes(m, nn, e, k, with_arg)
        rec = measured.get(key) if key else None
        roof = {"bound": "hbm", "achieved": b_alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None if not rec else rec.get("fabric_bytes_per_launch"),
                "kernel_avg_ms": ms, "algorithmic_bytes_per_launch": b_alg}
        if rec:
            roof["traffic_source"] = f"profiles/traffic.json[{key}]: " + rec.get("source", "rocprofv3 --pmc, separate passes")
        out.append(dict({"config": name, "ms": ms, "edges_per_s": e / (ms * 1e-3), "schedule": schedule, "roofline": roof}, **more))

    def run(red, k, val, name):
        x = synth.features(n, k, device=dev, integer=red in ("max", "min"))
        z = torch.empty((n, k), dtype=torch.float32, device=dev)
        arg = torch.empty((n, k), dtype=torch.int64, device=dev) if red in ("max", "min") else None
        msg = cabi.MESSAGE[red]
        geom = cabi.suggest_stream(n, n, nnz, k) if red in ("sum", "mean") else None
        geom_mm = cabi.suggest_stream_minmax(n, n, nnz, k) if red in ("max", "min") else None
        mplan = None if geom_mm is None else build_stream_plan(rowptr, col, val, n, geom_mm[1], None, None, geom_mm[0], geom_mm[2], minmax=True)
        if mplan is not None:
            ws = mplan.workspace(minmax=True)
            ms = _time_launches(lambda: cabi.fusedMM_csr_stream_minmax_hip(msg, rowptr, nnz, mplan, x, z, arg, ws))
            sched = f"stream (max / min kernel), {mplan.slices} slices, {mplan.gens} generation(s)"
        elif geom is not None:
            plan = build_stream_plan(rowptr, col, val, n, geom[1], None, None, geom[0], geom[2])
            ws = plan.workspace()
            ms = _time_launches(lambda: cabi.fusedMM_csr_stream_hip(msg, rowptr, nnz, plan, x, z, ws))
            sched = f"stream, {plan.slices} slices, {plan.gens} generation(s)"
        else:
            sl = skew_adjusted(rowptr, suggest_slices(n, n, nnz, k, red in ("max", "min")))
            plan = build_task_plan(rowptr, col, n, sl, col32=col32)
            ws = plan.workspace(red, k)
            ms = _time_launches(lambda: cabi.fusedMM_csr_tasks_hip(msg, rowptr, col, val, plan, x, z, arg, ws))
            sched = f"task list, {sl} slices"
        entry(name, ms, n, n, nnz, k, arg is not None, sched, key=f"reddit-{red}-k{k}-weighted")

    only = os.environ.get("ISPLIB_BENCH_ONLY", "")          # profiling: one configuration (its traffic.json key) instead of all
    for red in ("mean", "max", "min"):
        if not only or only == f"reddit-{red}-k64-weighted":
            run(red, 64, w, f"config 3: reddit-like SpMM-{red} K=64, U(0,1) weights" + (" (+arg)" if red != "mean" else ""))
    if not only or only == "reddit-sum-k128-weighted":
        run("sum", 128, w, "config 2: reddit-like SpMM-sum K=128, U(0,1) weights")
    # the third leg of config 2's backward when the edge weights are trainable: dA[e] = <X[col[e]], dY[row(e)]>, the SDDMM the
    # reference leaves commented out (csrc/fusedmm.cpp:270), through the graph handle (task list sized for whole rows)
    if not only or only == "reddit-sddmm-k128":
        h = cabi.GraphHandle(rowptr, col, w, n)
        xs, gs = synth.features(n, 128, device=dev), synth.features(n, 128, seed=5, device=dev)
        ms = _time_launches(lambda: h.sddmm(xs, gs))
        entry("config 2 backward, trainable weights: dA = SDDMM(X, dY) K=128", ms, n, n, nnz, 128, False,
              "task list (16 slices of whole rows) through isplib_graph_sddmm", key="reddit-sddmm-k128")
        h.close()
        del xs, gs
    del w, col32
    if only and not only.startswith("products"):
        return out

    if not only:
        out.append(gcn_epoch_config(dev, rowptr, col, n, with_cpu_epoch))
    out.extend(products_configs(dev, only))
    return out


def gcn_epoch_config(dev, rowptr, col, n, with_cpu_epoch):
    """config 5: the GCN epoch of tests/cpu/gcn-sparse.py:55-129 through iSpLibPlugin.patch_pyg (scripts/gcn_epoch.py restates
    it), with the same epoch on the host cores (the oracle doing every aggregation) beside it."""
    import numpy as np
    from isplib_amd import synth
    import importlib.util
    spec = importlib.util.spec_from_file_location("gcn_epoch", os.path.join(ROOT, "scripts", "gcn_epoch.py"))
    ge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ge)
    import isplib_amd
    import torch.nn.functional as F
    torch.manual_seed(0)
    feats, hidden, classes = 602, 32, 41
    x = synth.features(n, feats, device=dev)
    y = torch.randint(0, classes, (n,), device=dev)
    mask = torch.rand(n, device=dev) < 0.66
    n_train = int(mask.sum())
    model = ge.Net(feats, hidden, classes).to(dev)
    init = {k_: v.detach().cpu().clone() for k_, v in model.state_dict().items()}
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
    adj = isplib_amd.SparseTensor.from_csr(rowptr, col, None, (n, n), validate=False)
    isplib_amd.iSpLibPlugin.patch_pyg()
    times = []
    try:
        for epoch in range(6):                      # epoch 0 builds the per-graph operands; not timed
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.train()
            opt.zero_grad()
            o = model(x, adj, isplib_amd.matmul)
            loss = F.nll_loss(o[mask], y[mask], reduction="sum") / n_train
            loss.backward()
            opt.step()
            model(x, adj, isplib_amd.matmul).argmax(1)        # the second forward of :89
            torch.cuda.synchronize()
            if epoch:
                times.append(time.perf_counter() - t0)
    finally:
        isplib_amd.iSpLibPlugin.unpatch_pyg()
    gpu_epoch_ms = statistics.mean(times) * 1e3
    rec = {"config": "config 5: 2-layer GCN 602-32-41 epoch on the reddit-like graph through iSpLibPlugin.patch_pyg (6 SpMM + dense + Adam)",
           "ms": gpu_epoch_ms, "epoch_ms_std": statistics.pstdev(times) * 1e3, "epochs_timed": len(times)}
    if with_cpu_epoch:
        # the same epoch on the host cores with the oracle doing every aggregation (the reference's CPU mode `isplib`,
        # tests/cpu/gcn-sparse.py:29-36,83-92): A is symmetric with unit weights here, so A^T dY is the same call
        import oracle
        oracle.build()
        rp, cl = rowptr.cpu().numpy(), col.cpu().numpy()
        ones = np.ones(cl.size, np.float32)

        class CpuAgg(torch.autograd.Function):
            @staticmethod
            def forward(ctx, mat):
                return torch.from_numpy(oracle.spmm_fw(rp, cl, ones, mat.detach().numpy(), "sum")[0])

            @staticmethod
            def backward(ctx, g):
                return torch.from_numpy(oracle.spmm_fw(rp, cl, ones, g.contiguous().numpy(), "sum")[0])

        cpu_model = ge.Net(feats, hidden, classes)
        cpu_model.load_state_dict(init)
        cpu_opt = torch.optim.Adam(cpu_model.parameters(), lr=0.01, weight_decay=5e-4)
        xc, yc, mc = x.cpu(), y.cpu(), mask.cpu()
        agg = lambda a_, m_, r_: CpuAgg.apply(m_)  # noqa: E731
        ctimes = []
        for epoch in range(3):
            t0 = time.perf_counter()
            cpu_model.train()
            cpu_opt.zero_grad()
            o = cpu_model(xc, None, agg)
            loss = F.nll_loss(o[mc], yc[mc], reduction="sum") / n_train
            loss.backward()
            cpu_opt.step()
            cpu_model(xc, None, agg).argmax(1)
            if epoch:
                ctimes.append(time.perf_counter() - t0)
        rec["cpu_epoch"] = {"ms": statistics.mean(ctimes) * 1e3, "epochs_timed": len(ctimes), "cores": oracle.num_threads(), "kind": "port",
                            "what": "the same model and epoch structure on the host cores, every aggregation by oracle/fusedmm_oracle.c "
                                    "(OpenMP), dense layers and Adam by torch CPU"}
    del x, y, mask, model, opt, adj
    return rec


def products_configs(dev, only=""):
    """config 4's shape on ONE GPU: the dense operand (2.5 GB at K=256) is ten times the Infinity Cache, no schedule of this
    library reuses a gathered row of it (isplib_suggest_stream / isplib_suggest_slices both say so) and the plain
    row-per-wave kernel runs at the rate HBM serves random 1-KiB rows.  The only reuse there is lies in the graph: with the
    rows taken in a community order (isplib_amd/reorder.py, fusedMM_csr_ordered_hip; bit-identical results) the rows of a
    community are worked on together behind one XCD's L2.  Reported on BOTH graphs of this shape: the Chung-Lu graph
    BASELINE's generator makes (no structure: the order search finds none, nothing changes) and a degree-corrected
    stochastic block model of the same N, nnz and degree law (2,449 blocks, 80 % of the edges inside)."""
    from isplib_amd import cabi, reorder, synth
    try:
        measured = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:  # noqa: BLE001
        measured = {}
    out = []
    k = 256
    for tag, make in (("chunglu", lambda: synth.dataset_like("products", device=dev)), ("sbm", lambda: synth.sbm_like("products", device=dev))):
        if only and not only.startswith(f"products-{tag}"):
            continue
        torch.cuda.empty_cache()
        p_rowptr, p_col, pn = make()
        e = p_col.numel()
        px = synth.features(pn, k, device=dev)
        pz = torch.empty((pn, k), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        order = reorder.useful_order(p_rowptr, p_col)
        torch.cuda.synchronize()
        search_ms = (time.perf_counter() - t0) * 1e3
        graph = ("Chung-Lu graph (BASELINE's generator: no community structure)" if tag == "chunglu" else
                 "degree-corrected SBM of the same N, nnz and degree law, 2,449 blocks, 80 % of the edges inside")
        runs = [("plain row-per-wave kernel, rows in index order", None, f"products-{tag}-sum-k256-plain")]
        if order is not None:
            runs.append(("plain row-per-wave kernel, rows in the community order (label propagation, found once in "
                         f"{search_ms:.0f} ms; bit-identical result)", order, f"products-{tag}-sum-k256-ordered"))
        for sched, o, key in runs:
            if only and only != key:
                continue
            ms = _time_launches(lambda: cabi.fusedMM_csr_ordered_hip(cabi.MSG_SPMM_SUM, p_rowptr, p_col, None, o, px, pz))
            b_alg = synth.algorithmic_bytes(pn, pn, e, k, False)
            rec = measured.get(key)
            roof = {"bound": "hbm", "achieved": b_alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None if not rec else rec.get("fabric_bytes_per_launch"),
                    "kernel_avg_ms": ms, "algorithmic_bytes_per_launch": b_alg,
                    "gather_model_GBps": synth.gather_bytes(pn, e, k) / (ms * 1e-3) / 1e9}
            if rec:
                roof["traffic_source"] = f"profiles/traffic.json[{key}]: " + rec.get("source", "rocprofv3 --pmc, separate passes")
            out.append({"config": f"config 4 (one GPU): products-like SpMM-sum K=256, unit weights, N={pn}, nnz={e}; {graph}",
                        "ms": ms, "edges_per_s": e / (ms * 1e-3), "schedule": sched, "roofline": roof,
                        "order_search": ("a community order was kept" if order is not None else
                                         f"looked for a community order ({search_ms:.0f} ms, once): none worth keeping, index order")})
        del p_rowptr, p_col, px, pz, order
    return out


def replicated_graph(make, dev, rank, world):
    """(rowptr, col, n) on every rank: rank 0 generates, the others receive its copy -- the partition (row cuts, shard
    pitch, collective sizes) can then never differ between ranks, and ranks that share a GPU (rehearsals) do not all
    run the generator's sorts on it at once."""
    if world == 1:
        return make()
    import torch.distributed as dist
    if rank == 0:
        rowptr, col, n = make()
        shape = torch.tensor([rowptr.numel(), col.numel(), n], dtype=torch.int64, device=dev)
    else:
        shape = torch.zeros(3, dtype=torch.int64, device=dev)
    dist.broadcast(shape, 0)
    if rank != 0:
        rowptr = torch.empty(int(shape[0]), dtype=torch.int64, device=dev)
        col = torch.empty(int(shape[1]), dtype=torch.int64, device=dev)
        n = int(shape[2])
    dist.broadcast(rowptr, 0)
    dist.broadcast(col, 0)
    return rowptr, col, n


def dist_extra_configs(dev, rank, world, rowptr, col, n):
    """The two configurations of BASELINE.json that exist only on several GPUs, run on the ranks of this job after the
    headline (outside its timed region): config 5 -- the 2-layer GCN epoch on the 1-D row-partitioned graph
    (isplib_amd.dist.DistGraph: one all-gather of the layer input per aggregation, forward and backward; replicated
    dense weights, gradients all-reduced) -- and config 4 -- the ogbn-products-shaped SpMM-sum at K=256, one
    all-gather(X) + the local SpMM per step.  Wall clock between device synchronisations, max over ranks."""
    import importlib.util
    import torch.distributed as dist
    import torch.nn.functional as F
    from isplib_amd import cabi, synth
    from isplib_amd.dist import DistGraph, RowPartition
    out = []

    def over_ranks(seconds):
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)

    # config 5
    spec = importlib.util.spec_from_file_location("gcn_epoch", os.path.join(ROOT, "scripts", "gcn_epoch.py"))
    ge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ge)
    torch.manual_seed(0)                                     # the same replicated weights and labels on every rank
    feats, hidden, classes = 602, 32, 41
    x = synth.features(n, feats, device=dev)
    y = torch.randint(0, classes, (n,), device=dev)
    mask = torch.rand(n, device=dev) < 0.66
    n_train = int(mask.sum())
    model = ge.Net(feats, hidden, classes).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
    graph = DistGraph(rowptr, col, None, n, rank, world)
    r0, r1 = graph.row0, graph.row0 + graph.rows
    x, y, mask = x[r0:r1].contiguous(), y[r0:r1].contiguous(), mask[r0:r1].contiguous()
    agg = lambda g_, m_, red_: g_.matmul(m_)  # noqa: E731
    times = []
    for epoch in range(6):                                   # epoch 0 builds the per-graph operands; not timed
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        model.train()
        opt.zero_grad()
        o = model(x, graph, agg)
        loss = F.nll_loss(o[mask], y[mask], reduction="sum") / n_train
        loss.backward()
        for prm in model.parameters():                       # replicated dense weights: sum the shard gradients
            dist.all_reduce(prm.grad)
        opt.step()
        model(x, graph, agg).argmax(1)                       # the second forward of tests/cpu/gcn-sparse.py:89
        torch.cuda.synchronize()
        if epoch:
            times.append(time.perf_counter() - t0)
    out.append({"config": f"config 5 on {world} GPUs: 2-layer GCN 602-32-41 epoch on the reddit-like graph, rows of A, A^T and X "
                          "partitioned by nnz (6 aggregations = 6 all-gathers + local SpMM, dense layers on the local rows, "
                          "weight gradients all-reduced, Adam)",
                "ms": over_ranks(statistics.mean(times)) * 1e3, "epochs_timed": len(times), "n_gpus": world})
    del x, y, mask, model, opt, graph
    torch.cuda.empty_cache()

    # config 4
    k = 256
    p_rowptr, p_col, pn = replicated_graph(lambda: synth.dataset_like("products", device=dev), dev, rank, world)
    e = p_col.numel()
    part = RowPartition(p_rowptr, p_col, None, pn, rank, world)
    del p_rowptr, p_col
    shard = part.shard(synth.features(pn, k, device=dev))
    torch.cuda.empty_cache()
    buf = part.gather_buffer(k)
    z = torch.empty((part.rows, k), dtype=torch.float32, device=dev)

    # One step = the exchange of X + the local SpMM.  This is the workload where the exchange dominates (2.19 GB inbound
    # per rank at 8 ranks against ~2.3 ms of local SpMM), so the overlapped forms of isplib_amd/dist.py are candidates
    # beside the blocking all-gather, each validated against it once (1e-5 of sum |a||x| per element: other summation
    # orders) and timed for a few steps; the fastest on THIS node is kept and named.
    def gather_plain():
        part.all_gather(shard, buf)
        cabi.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, part.rowptr, part.col_padded, None, buf, z)

    # K = 256 as 4 panels of 64 columns: every panel its own all-gather, issued up front; panel c is aggregated (plain
    # kernel on the panel: columns are independent) while panels c+1.. are still on the links
    panels = [(c0, min(k, c0 + 64)) for c0 in range(0, k, 64)]
    p_send = [torch.zeros((part.max_rows, c1 - c0), dtype=torch.float32, device=dev) for c0, c1 in panels]
    p_recv = [torch.empty((part.ncols_padded, c1 - c0), dtype=torch.float32, device=dev) for c0, c1 in panels]

    def pipelined_plain():
        handles = []
        for (c0, c1), sb, rb in zip(panels, p_send, p_recv):
            sb.copy_(shard[:, c0:c1])
            handles.append(dist.all_gather_into_tensor(rb, sb, async_op=True))
        for (c0, c1), rb, h in zip(panels, p_recv, handles):
            h.wait()
            cabi.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, part.rowptr, part.col_padded, None, rb, z[:, c0:c1])

    candidates = {"gather + plain kernel": gather_plain, "pipelined x4 (64-column panels, plain kernel)": pipelined_plain}

    def all_agree(good):
        flag = torch.tensor([1 if good else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    # the library's pipelined form (task list or stream schedule per panel) where its rules accept this rank's shard
    for stream_form in (True, False):
        state = None
        try:
            state = part.pipeline_state(k, 4, "sum", stream=stream_form)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] config 4: pipeline_state(stream={stream_form}) raised {type(e).__name__}: {e}", file=sys.stderr)
        if all_agree(state is not None):
            candidates["pipelined x4 (" + ("stream schedule" if stream_form else "task list") + " per panel)"] = \
                (lambda st: lambda: part.spmm_pipelined(shard, z, st, "sum"))(state)
    # direct per-peer exchange: one column slice per shard (the slice rule itself may decline a graph this sparse), the
    # shards aggregated as their group lands; over gloo a point-to-point shard takes seconds (rehearsals skip it)
    if os.environ.get("ISPLIB_BENCH_BACKEND", "nccl") == "nccl" or os.environ.get("ISPLIB_BENCH_DIRECT") == "1":
        dplan = None
        try:
            dplan = part.plan(k, "sum", slices=world)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] config 4: sliced plan raised {type(e).__name__}: {e}", file=sys.stderr)
        if all_agree(dplan is not None):
            for nb in sorted({1, 2, world - 1}):
                if 1 <= nb <= world - 1:
                    candidates[f"direct x{nb} (per-peer send / receive, {world} column slices)"] = \
                        (lambda b: lambda: part.spmm_direct(shard, buf, z, dplan, "sum", None, batches=b))(nb)

    gather_plain()
    want = z.clone()
    cabi.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, part.rowptr, part.col_padded, None, buf.abs(), z)
    bound = z.abs() * 1e-5 + 1e-30
    times = {}
    for name, fn in list(candidates.items()):
        good = True
        try:
            z.zero_()
            fn()
            torch.cuda.synchronize()
            good = bool(((z - want).abs() <= bound).all())
        except Exception as e:  # noqa: BLE001
            print(f"[bench] config 4 schedule '{name}' raised {type(e).__name__}: {e}", file=sys.stderr)
            good = False
        if not all_agree(good):
            if rank == 0:
                print(f"[bench] config 4 schedule '{name}' disabled (mismatch or error)", file=sys.stderr)
            del candidates[name]
            continue
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        first = over_ranks(time.perf_counter() - t0) * 1e3
        if first > 2000.0:                     # a schedule this slow (a rehearsal over gloo) is not worth more steps
            times[name] = first
            continue
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        times[name] = over_ranks(time.perf_counter() - t0) / 5 * 1e3
    del want, bound
    chosen = min(times, key=times.get)
    ms = times[chosen]
    if rank == 0:
        print(f"[bench] config 4, N={world}: " + ", ".join(f"{n_} {t_:.3f} ms/step" for n_, t_ in times.items()) + f" -> {chosen}", file=sys.stderr)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    part.all_gather(shard, buf)
    ev0.record()
    cabi.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, part.rowptr, part.col_padded, None, buf, z)
    ev1.record()
    torch.cuda.synchronize()
    b_alg = synth.algorithmic_bytes(pn, pn, e, k, False)
    out.append({"config": f"config 4 on {world} GPUs: products-like SpMM-sum K=256 (N={pn}, nnz={e}), rows by nnz, one exchange of X "
                          f"({pn * k * 4 / 1e9:.2f} GB gathered per rank) + the local SpMM per step",
                "schedule": chosen, "candidates_ms": {n_: round(t_, 3) for n_, t_ in times.items()},
                "ms": ms, "edges_per_s": e / (ms * 1e-3), "local_spmm_ms_rank0": ev0.elapsed_time(ev1), "n_gpus": world,
                "roofline": {"bound": "hbm", "achieved": b_alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
                             "frac": b_alg / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBPS * world), "algorithmic_bytes_per_launch": b_alg}})
    return out


def launcher_command(gpus: int, argv, port: int):
    """The command `python bench.py --gpus N ...` turns into when no launcher set WORLD_SIZE: one fresh process per
    GPU under torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def visible_gpu_count(sysfs: str = "/sys/class/kfd/kfd/topology/nodes", dri: str = "/dev/dri", env=None):
    """How many GPUs a HIP process started from here would see, WITHOUT loading HIP (or amdsmi) in this process: the
    launcher parent must stay GPU-free (a process that has initialised the GPU may neither fork ranks nor be replaced).
    KFD topology nodes with simd_count > 0 are GPUs (CPU nodes have 0); a node whose render device this user cannot
    open is not ours (container isolation); ROCR_ / HIP_ / CUDA_VISIBLE_DEVICES narrow the list the way the runtime
    reads them (each a comma list of indices into what is left, or GPU-<uuid> tokens; the list ends at the first index
    that does not exist).  0 without a KFD driver; None when the driver is there but its topology is not readable (the
    ranks then find out themselves)."""
    import glob
    env = os.environ if env is None else env
    nodes = []
    for nd in glob.glob(os.path.join(sysfs, "*")):
        try:
            idx = int(os.path.basename(nd))
            with open(os.path.join(nd, "properties")) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
        except (OSError, ValueError):
            continue
        if int(props.get("simd_count", "0")) <= 0:
            continue
        minor = int(props.get("drm_render_minor", "-1"))
        if minor > 0 and os.path.isdir(dri) and not os.access(os.path.join(dri, f"renderD{minor}"), os.R_OK | os.W_OK):
            continue
        nodes.append(idx)
    if not os.path.isdir(sysfs):
        # no topology to read: without /dev/kfd there is no ROCm device at all; with it, the count is unknown from here
        return None if os.path.exists("/dev/kfd") else 0
    count = len(nodes)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        spec = env.get(var)
        if spec is None:
            continue
        kept = 0
        for tok in (t.strip() for t in spec.split(",")):
            if tok.lstrip("-").isdigit():
                if not 0 <= int(tok) < count:
                    break
            elif not tok.upper().startswith("GPU-"):
                break
            kept += 1
        count = min(count, kept)
    return count


def self_launch(a, argv) -> int:
    """`--gpus N` with N > 1 and no WORLD_SIZE: this process becomes the launcher.  It must not touch the GPU (a
    process that has initialised HIP may neither fork ranks nor be replaced): it only counts devices, starts the
    ranks as children, relays rank 0's JSON line and returns non-zero if any child failed.  Devices are counted from
    sysfs (visible_gpu_count): torch.cuda.device_count() goes through amdsmi or hipGetDeviceCount, i.e. may load HIP."""
    import socket
    import subprocess
    backend = os.environ.get("ISPLIB_BENCH_BACKEND", "nccl")
    have = visible_gpu_count()                   # from sysfs: no HIP call, no amdsmi, nothing that could initialise a GPU
    if backend == "nccl" and have is not None and have < a.gpus:
        print(f"bench.py: --gpus {a.gpus} but only {have} GPU(s) visible: refusing to print a number for fewer ranks "
              "(ISPLIB_BENCH_BACKEND=gloo rehearses N ranks on one GPU, labelled as a rehearsal)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = launcher_command(a.gpus, argv, port)
    print("[bench] launching: " + " ".join(cmd), file=sys.stderr)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"bench.py: the {a.gpus}-rank run failed (exit {proc.returncode}, JSON line {'present' if line else 'missing'})", file=sys.stderr)
        return proc.returncode or 3
    if json.loads(line).get("n_gpus") != a.gpus:
        print(f"bench.py: the ranks reported n_gpus={json.loads(line).get('n_gpus')}, expected {a.gpus}", file=sys.stderr)
        return 4
    print(line, flush=True)
    return 0


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(self_launch(a, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        # a rank stuck in a collective says where: every thread's Python stack on stderr after ISPLIB_BENCH_WATCHDOG
        # seconds (default 240) without finishing, and again every so many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ.get("ISPLIB_BENCH_WATCHDOG", "240")), repeat=True, file=sys.stderr)

    def note(msg):
        if world > 1 and rank == 0:
            print(f"[bench] {time.strftime('%H:%M:%S')} {msg}", file=sys.stderr, flush=True)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or drop WORLD_SIZE and let bench.py start them)")
    # ISPLIB_BENCH_FORCE_DIST=1 under `torch.distributed.run --nproc-per-node 1` walks the N > 1 code (RCCL init,
    # partition, async all-gather, overlapped schedule) with a single rank: a rehearsal of the API calls on a
    # 1-GPU box, labelled as such in the output line.
    forced = world == 1 and os.environ.get("ISPLIB_BENCH_FORCE_DIST") == "1"
    multi = world > 1 or forced
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    # one process per GPU; backend "nccl" is RCCL over xGMI.  ISPLIB_BENCH_BACKEND=gloo lets several ranks
    # share one GPU to rehearse the N > 1 code path on a single-GPU box (never used for a reported number).
    backend = os.environ.get("ISPLIB_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if a.only and not multi:
        # one configuration of the `extra` array on its own (what scripts/prof_pmc.sh and rocprofv3 --kernel-trace are
        # pointed at to fill profiles/traffic.json and the per-configuration kernel statistics)
        from isplib_amd import synth as _synth
        os.environ["ISPLIB_BENCH_ONLY"] = a.only
        if a.only.startswith("products"):
            extra = products_configs(dev, a.only)
        else:
            g_rowptr, g_col, g_n = _synth.dataset_like("reddit", device=dev)
            extra = extra_configs(dev, g_rowptr, g_col, g_n, with_cpu_epoch=False)
        print(json.dumps({"only": a.only, "extra": extra}), flush=True)
        return
    import torch.distributed as dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        if rank == 0:
            print(f"[bench] process group up: backend={dist.get_backend()} world_size={dist.get_world_size()} "
                  f"(RCCL ranks: {dist.get_world_size() if backend == 'nccl' else 0})", file=sys.stderr)

    from isplib_amd import cabi, synth
    k = a.k or {"reddit": 128, "cora": 16, "products": 256}[a.workload]
    def make_graph():
        if a.generator == "chunglu":
            return synth.dataset_like(a.workload, device=dev, scale=a.scale)
        n_, target = synth.SHAPES[a.workload][0], synth.SHAPES[a.workload][1]
        n_, target = max(64, int(n_ * a.scale)), int(target * a.scale) // 2 * 2
        r_, c_ = (synth.rmat_csr if a.generator == "rmat" else synth.uniform_csr)(n_, target, device=dev)
        return r_, c_, n_

    rowptr, col, n = replicated_graph(make_graph, dev, rank, world)
    nnz = col.numel()
    note(f"graph ready on every rank (nnz={nnz})")
    x = synth.features(n, k, device=dev)
    val = synth.edge_weights(nnz, device=dev) if a.weighted else None
    msg = cabi.MESSAGE[a.reduce]
    for kv in filter(None, a.tune.split(",")):
        key, value = kv.split("=")
        cabi.lib().isplib_hip_tune(int(key), int(value))

    if not multi:
        l_rowptr, l_col, l_val, m_local, x_in = rowptr, col, val, n, x
        out = torch.empty((n, k), dtype=torch.float32, device=dev)
        arg = torch.empty((n, k), dtype=torch.int64, device=dev) if a.reduce in ("max", "min") else None
        gather = None
    else:
        from isplib_amd.dist import RowPartition
        part = RowPartition(rowptr, col, val, n, rank, world)
        l_rowptr, l_col, l_val, m_local = part.rowptr, part.col_padded, part.val, part.rows
        x_shard = part.shard(x)
        x_in = part.gather_buffer(k)
        out = torch.empty((m_local, k), dtype=torch.float32, device=dev)
        arg = torch.empty((m_local, k), dtype=torch.int64, device=dev) if a.reduce in ("max", "min") else None
        gather = lambda: part.all_gather(x_shard, x_in)  # noqa: E731

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

    # per-graph preparation, outside the timed region (the reference also builds its per-graph
    # operands once, isplib/__init__.py:76-106): slice table + workspace of the column-sliced path
    from isplib_amd.plugin import suggest_slices
    table = work = plan = None
    user_slices = a.slices >= 0
    if a.slices < 0 and a.schedule == "sliced":      # the one-pass sliced kernel has no column panels: whole-row rule
        a.slices = int(cabi.lib().isplib_suggest_slices_whole_rows(m_local, x_in.size(0), l_col.numel(), k))
    if a.slices < 0:
        from isplib_amd.plugin import skew_adjusted
        a.slices = skew_adjusted(l_rowptr, suggest_slices(m_local, x_in.size(0), l_col.numel(), k, a.reduce in ("max", "min")))
    sliced_slices = a.slices        # slice count of the one-pass column-sliced kernel (differs from the task plan's at N > 1)
    if multi and a.slices > 0:
        # the sliced kernel has no column panels: its own rule (whole rows), rounded to a multiple of world
        plan = part.plan(k, a.reduce, slices=a.slices if user_slices else None)
        if plan is not None:
            sliced_slices, table, work = plan
    elif a.slices > 0:
        table, ok = cabi.spmm_slices(l_rowptr, l_col, x_in.size(0), a.slices)
        if not ok:
            raise SystemExit("synthetic graph rows are not column-sorted?")
        work = cabi.sliced_workspace(a.reduce, m_local, k, a.slices, dev)

    # stream schedule (sum / mean): the default wherever the rule expects it to win
    splan = swork = None
    if a.schedule in ("auto", "stream") and a.reduce in ("sum", "mean"):
        from isplib_amd.plan import build_stream_plan
        geom = tuple(int(v) for v in a.stream_geom.split(":")) if a.stream_geom else \
            cabi.suggest_stream(m_local, x_in.size(0), l_col.numel(), k)
        if geom is not None and not a.stream_geom:          # no degree skew: slices closer to the L2 size (the plug-in's rule)
            from isplib_amd.plugin import skew_adjusted as _skew
            geom = (geom[0], _skew(l_rowptr, geom[1], cap=512), geom[2])
        if geom is not None:
            splan = build_stream_plan(l_rowptr, l_col, l_val, x_in.size(0), geom[1], None, None, geom[0], geom[2])
            swork = None if splan is None else splan.workspace()
        if splan is None and a.schedule == "stream":
            raise SystemExit("--schedule stream: the stream schedule does not apply to this shape (isplib_suggest_stream)")
    if a.schedule in ("auto", "stream"):
        a.schedule = "tasks"            # what everything the stream schedule does not serve falls back to
    use_stream = splan is not None and not multi     # N > 1: decided by a short measurement below

    tplan = twork = None
    if a.schedule == "tasks" and a.slices > 0:
        from isplib_amd.plan import build_task_plan
        tplan = build_task_plan(l_rowptr, l_col, x_in.size(0), a.slices, a.chunk, a.short)
        if not user_slices and tplan is not None and l_col.numel() / max(tplan.n_tasks, 1) >= 120.0:
            # hub-dominated graph: tasks are long even with the whole-row slice count -> that plan, run in one pass
            whole = int(cabi.lib().isplib_suggest_slices_whole_rows(m_local, x_in.size(0), l_col.numel(), k))
            if whole > a.slices:
                a.slices = whole
                tplan = build_task_plan(l_rowptr, l_col, x_in.size(0), a.slices, a.chunk, a.short)
        twork = tplan.workspace(a.reduce, k)
    note("plans built")
    if world > 1:
        # Which schedules exist is decided from this rank's shard (its row count, its degree skew); a schedule that one
        # rank has and another lacks would leave the ranks in different collectives below.  Keep what EVERY rank has.
        have = torch.tensor([plan is not None, splan is not None, tplan is not None], dtype=torch.int32, device=dev)
        dist.all_reduce(have, op=dist.ReduceOp.MIN)
        have = have.tolist()
        if not have[0]:
            plan = None
        if not have[1]:
            splan = swork = None
        if not have[2]:
            tplan = twork = None
    use_tasks = tplan is not None and not multi      # N > 1: decided by a short measurement below

    def spmm(rp, cl, vl, tb, xin, o, ar, tp=None, sp=None):
        sp = splan if (sp is None and tp is None and rp is l_rowptr and use_stream) else sp
        if sp is not None:
            cabi.fusedMM_csr_stream_hip(msg, rp, cl.numel(), sp, xin, o, swork if sp is splan else swork_t)
            return
        tp = tplan if (tp is None and rp is l_rowptr and use_tasks) else tp
        if tp is not None:
            cabi.fusedMM_csr_tasks_hip(msg, rp, cl, vl, tp, xin, o, ar, twork)
        elif tb is not None:
            cabi.fusedMM_csr_sliced_hip(msg, rp, cl, vl, tb, sliced_slices, xin, o, ar, work)
        else:
            cabi.fusedMM_csr_hip(msg, rp, cl, vl, xin, o, ar)

    # N > 1: one step = the exchange of X plus the local SpMM.  Candidate schedules (isplib_amd/dist.py):
    #   gather+spmm      : one all-gather, then the SpMM (task list when a plan exists)
    #   overlapped sliced: local column slices aggregated while the all-gather is in flight
    #   pipelined xC     : X travels in C column panels; panel c is aggregated while panels c+1.. travel
    #   direct xB        : P-1 per-peer send / receive pairs in B groups, a group's shards aggregated as it lands
    #   gather+stream    : one all-gather, then the stream schedule
    #   pipelined stream xC : the pipelined exchange with the stream schedule on every panel
    # Every candidate is first checked against gather+spmm of the same kernel family (bit for bit, except that
    # panelled sums are held to the parity tests' 1e-5 bound: their summation order differs), then all
    # are timed for a few steps (max over ranks) and the fastest is kept: which one wins depends on how
    # long the collective takes on this node.  ISPLIB_OVERLAP=0 keeps gather+spmm.
    chosen = "single GPU"
    step_fn = None
    if multi:
        def all_ranks_agree(good):
            flag = torch.tensor([1 if good else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())

        def checked(name, fn, want_fn, exact=True):
            note(f"checking schedule '{name}'")
            try:
                want_fn()
                want = out.clone()
                out.zero_()
                fn()
                torch.cuda.synchronize()
                if exact:
                    good = torch.equal(out, want)
                else:           # other summation order: 1e-5 of sum |a||x| per element, the parity tests' bound
                    good = bool(((out - want).abs() <= 1e-5 * magnitude() + 1e-30).all())
            except Exception as e:  # noqa: BLE001
                print(f"[bench] schedule '{name}' raised {type(e).__name__}: {e}", file=sys.stderr)
                good = False
            good = all_ranks_agree(good)
            if not good and rank == 0:
                print(f"[bench] schedule '{name}' disabled (mismatch or error)", file=sys.stderr)
            return good

        def gather_then_sliced():
            gather()
            spmm(l_rowptr, l_col, l_val, table, x_in, out, arg)

        def gather_then_tasks():
            gather()
            cabi.fusedMM_csr_tasks_hip(msg, l_rowptr, l_col, l_val, tplan, x_in, out, arg, twork)

        def magnitude():
            gather()
            keep = out.clone()
            cabi.fusedMM_csr_hip(msg, l_rowptr, l_col, None if l_val is None else l_val.abs(), x_in.abs(), out, arg)
            if a.reduce == "mean":
                out.mul_((l_rowptr[1:] - l_rowptr[:-1]).clamp(min=1).unsqueeze(1))
            mag = out.abs().clone()
            out.copy_(keep)
            return mag

        def gather_then_stream():
            gather()
            cabi.fusedMM_csr_stream_hip(msg, l_rowptr, l_col.numel(), splan, x_in, out, swork)

        base = gather_then_tasks if tplan is not None else gather_then_sliced
        candidates = {"gather+spmm": base}
        if splan is not None and checked("gather+stream", gather_then_stream, base, exact=False):
            candidates["gather+stream"] = gather_then_stream
        if os.environ.get("ISPLIB_OVERLAP", "1") != "0":
            if plan is not None:
                fn = lambda: part.spmm_overlapped(x_shard, x_in, out, plan, a.reduce, arg)  # noqa: E731
                if checked("overlapped sliced", fn, gather_then_sliced):
                    candidates["overlapped sliced"] = fn
            # (over gloo a point-to-point transfer of a shard takes seconds: the rehearsal leaves these to tests/test_gpu_dist.py)
            if plan is not None and world > 1 and (backend == "nccl" or os.environ.get("ISPLIB_BENCH_DIRECT") == "1"):
                for nb in sorted({1, 2, world - 1}):
                    if nb > world - 1:
                        continue
                    fn = (lambda b: lambda: part.spmm_direct(x_shard, x_in, out, plan, a.reduce, arg, batches=b))(nb)
                    if checked(f"direct x{nb}", fn, gather_then_sliced):
                        candidates[f"direct x{nb}"] = fn
            if tplan is not None:
                for panels in (2, 4):
                    if k // panels < 16:
                        continue
                    state = part.pipeline_state(k, panels, a.reduce)        # own plan: slice count for the panel width
                    if not all_ranks_agree(state is not None):              # (a collective: every rank asks, whatever it got)
                        continue
                    fn = (lambda st: lambda: part.spmm_pipelined(x_shard, out, st, a.reduce, arg))(state)
                    if checked(f"pipelined x{panels}", fn, gather_then_tasks, exact=a.reduce in ("max", "min")):
                        candidates[f"pipelined x{panels}"] = fn

            if splan is not None and "gather+stream" in candidates:
                for panels in (2, 4):
                    if k // panels < 32:
                        continue
                    state = part.pipeline_state(k, panels, a.reduce, stream=True)
                    if not all_ranks_agree(state is not None):
                        continue
                    fn = (lambda st: lambda: part.spmm_pipelined(x_shard, out, st, a.reduce, arg))(state)
                    if checked(f"pipelined stream x{panels}", fn, gather_then_stream, exact=False):
                        candidates[f"pipelined stream x{panels}"] = fn

        def timed(fn, reps=4):
            def clocked(count):
                torch.cuda.synchronize()
                dist.barrier()
                t_ = time.perf_counter()
                for _ in range(count):
                    fn()
                torch.cuda.synchronize()
                tt = torch.tensor([time.perf_counter() - t_], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                return float(tt) / count * 1e3
            first = clocked(1)                # (also the warm-up; the same value on every rank, so is the branch below)
            return first if first > 250.0 else clocked(reps)      # a schedule this slow is not worth four more steps

        times = {}
        for name, fn in candidates.items():
            note(f"timing schedule '{name}'")
            times[name] = timed(fn)
        chosen = min(times, key=times.get)
        use_tasks = tplan is not None and chosen != "overlapped sliced"
        use_stream = chosen == "gather+stream"
        if chosen.startswith("pipelined stream"):
            use_tasks = False
        if chosen.startswith("direct"):
            use_tasks = False
        if chosen not in ("gather+spmm", "gather+stream"):
            step_fn = candidates[chosen]
        if rank == 0:
            print(f"[bench] N={world}: " + ", ".join(f"{n_} {t_:.3f} ms/step" for n_, t_ in times.items()) + f" -> {chosen}",
                  file=sys.stderr)

    def step(i=None):
        if step_fn is not None:              # exchange and compute are interleaved: the events bracket both
            if i is not None:
                ev[i][0].record()
            step_fn()
            if i is not None:
                ev[i][1].record()
            return
        if gather is not None:
            gather()
        if i is not None:
            ev[i][0].record()
        spmm(l_rowptr, l_col, l_val, table, x_in, out, arg)
        if i is not None:
            ev[i][1].record()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = [s.elapsed_time(e) for s, e in ev]
    kern_avg_ms = sum(kern_ms) / len(kern_ms)

    # SURVEY.md 8(d) extras, outside the timed region, N=1 only: (i) the same launch with L2 + Infinity Cache
    # flushed first (a 1 GiB fill evicts the 256 MiB MALL), (ii) this box's device-to-device copy rate, the
    # "measured peak" the roofline fraction is also quoted against.
    cold_ms = copy_gbps = None
    if not multi:
        scratch = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        colds = []
        for _ in range(3):
            scratch.fill_(1.0)
            s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            spmm(l_rowptr, l_col, l_val, table, x_in, out, arg)
            e0.record()
            torch.cuda.synchronize()
            colds.append(s0.elapsed_time(e0))
        cold_ms = sorted(colds)[1]
        half = scratch.numel() // 2
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        scratch[half:].copy_(scratch[:half])
        s0.record()
        for _ in range(10):
            scratch[half:].copy_(scratch[:half])
        e0.record()
        torch.cuda.synchronize()
        copy_gbps = 10 * 2 * half * 4 / (s0.elapsed_time(e0) * 1e-3) / 1e9      # bytes read + written
        del scratch

    # backward of SpMM-sum = the same kernel on A^T (csrc/fusedmm.cpp:285); reported beside the metric
    bwd = None
    swork_t = None
    if not multi and not a.no_backward and a.reduce == "sum":
        colptr, _, row_t, val_t = cabi.csr2csc(rowptr, col, val, n, want_perm=False, want_val=val is not None)
        dy = synth.features(n, k, seed=5, device=dev)
        dx = torch.empty((n, k), dtype=torch.float32, device=dev)
        table_t = cabi.spmm_slices(colptr, row_t, n, a.slices)[0] if a.slices > 0 else None
        tplan_t = splan_t = None
        if use_stream:
            splan_t = build_stream_plan(colptr, row_t, val_t, n, splan.slices, None, None, splan.streams, splan.chunk)
            swork_t = splan_t.workspace()
        elif tplan is not None:
            tplan_t = build_task_plan(colptr, row_t, n, a.slices, a.chunk, a.short)
            twork = tplan_t.workspace(a.reduce, k) if tplan_t.n_tasks > tplan.n_tasks else twork
        for _ in range(2):
            spmm(colptr, row_t, val_t, table_t, dy, dx, None, tplan_t, splan_t)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            spmm(colptr, row_t, val_t, table_t, dy, dx, None, tplan_t, splan_t)
        e.record()
        torch.cuda.synchronize()
        bms = s.elapsed_time(e) / 5
        bwd = {"ms": bms, "edges_per_s": nnz / (bms * 1e-3)}
        del colptr, row_t, val_t, dy, dx

    res = None
    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        with_arg = a.reduce in ("max", "min")
        # dominant kernel = the SpMM launch of this rank (rank 0's slice when partitioned)
        b_alg = synth.algorithmic_bytes(m_local, n, l_col.numel(), k, with_arg)
        achieved = b_alg / (kern_avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        # measured offline (separate --pmc passes cannot run inside this process): profiles/traffic.json
        if os.path.exists(tpath) and not multi and a.scale == 1.0 and not a.weighted and a.generator == "chunglu" \
                and (a.chunk, a.short) == (1024, 128):
            try:
                rec = json.load(open(tpath)).get(f"{a.workload}-{a.reduce}-k{k}-" + (f"stream{splan.slices}" if use_stream else
                                                 f"s{a.slices}" + ("-tasks" if tplan is not None else "")))
                traffic = rec.get("fabric_bytes_per_launch", rec.get("hbm_bytes_per_launch")) if rec else None
            except Exception:
                traffic = None
        if chosen.startswith("pipelined stream"):
            kernel_label = f"spmm_stream_kernel + sweep_hub_fold_kernel per column panel ({chosen}), exchange included in the events"
        elif use_stream:
            pw = 256 // splan.streams
            kernel_label = (f"spmm_stream_kernel x {splan.gens} generation(s) + sweep_hub_fold_kernel, "
                            f"{-(-k // pw)} pass(es) of {pw} columns per launch")
        elif use_tasks:
            kernel_label = "spmm_task_kernel + combine_tasks_kernel"
            if x_in.size(0) * k * 4 / max(a.slices, 1) > 9216 * 1024:          # else a whole-row plan: one pass
                if k >= 96 and k % 32 == 0:
                    kernel_label += f", {-(-k // 64)} passes of 64 columns per launch"
                elif k >= 192:
                    kernel_label += f", {-(-k // 128)} passes of 128 columns per launch"
        else:
            kernel_label = "spmm_csr_kernel" + (f"<sliced x{sliced_slices}> + combine_slices_kernel" if a.slices > 0 else "")
        res = {
            "metric": "edges_aggregated_per_sec", "value": nnz / (elapsed / a.steps), "unit": "edges/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" + (" (REHEARSAL of the N>1 path on one rank, not a result)" if forced else "" if backend == "nccl" else f" (REHEARSAL over {backend}, not a result)"),
            "config": {
                "workload": f"{a.workload}-like graph ({a.generator}, N={n}, nnz={nnz}), SpMM-{a.reduce} forward, K={k}, fp32"
                            + (", U(0,1) weights" if a.weighted else ", unit weights")
                            + ("" if a.scale == 1.0 else f", SCALED x{a.scale} (debug)"),
                "schedule": (f"stream: {splan.streams} streams x {splan.rows_per_wave // splan.streams} rows per wave, {splan.slices} column slices, "
                             f"{splan.gens} generation(s) of {splan.waves_per_gen} waves, rows > {splan.chunk} edges dealt to {splan.n_parts} virtual rows"
                             if use_stream else
                             f"stream schedule per column panel ({chosen})" if chosen.startswith("pipelined stream") else
                             f"task list: {a.slices} column slices, {tplan.n_tasks} tasks of <= {a.chunk} edges, rows < {a.short} unsliced"
                             if use_tasks else
                             f"{sliced_slices} column slices, XCD-affine" if a.slices > 0 else "row-per-wave, unsliced"),
                "partition": "none" if not multi else f"1-D rows by nnz, {world} ranks, one all-gather(X) per step"
                             + f", schedule: {chosen}",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "traffic_source": None if traffic is None else "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled (gfx950), per launch; these are the bytes leaving the XCD L2s, Infinity-Cache hits included",
                "kernel": kernel_label,
                "kernel_avg_ms": kern_avg_ms, "kernel_median_ms": sorted(kern_ms)[len(kern_ms) // 2], "kernel_min_ms": min(kern_ms),
                "kernel_cold_cache_ms": cold_ms, "peak_measured_copy": copy_gbps,
                "frac_of_measured_copy": None if not copy_gbps else achieved / copy_gbps,
                "algorithmic_bytes_per_launch": b_alg,
                "gather_model_GBps": synth.gather_bytes(m_local, l_col.numel(), k) / (kern_avg_ms * 1e-3) / 1e9,
            },
        }
        if bwd:
            res["backward"] = bwd
        if not multi and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(rowptr, col, x, nnz)
        if not multi and not a.no_extra and a.workload == "reddit" and a.scale == 1.0 and a.generator == "chunglu":
            del x, out
            torch.cuda.empty_cache()
            res["extra"] = extra_configs(dev, rowptr, col, n, with_cpu_epoch=not a.no_cpu_baseline)
    if world > 1 and not a.no_extra and a.workload == "reddit" and a.scale == 1.0 and a.generator == "chunglu":
        # every rank: the multi-GPU-only configurations, after the headline and outside its timed region
        step_fn = candidates = splan = tplan = plan = swork = twork = table = work = None
        del x_in, out, x_shard, part, x
        torch.cuda.empty_cache()
        dist_extra = dist_extra_configs(dev, rank, world, rowptr, col, n)
        if rank == 0:
            res["extra"] = dist_extra
    if rank == 0:
        print(json.dumps(res), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if world > 1:
        import faulthandler
        faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
