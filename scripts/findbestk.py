#!/usr/bin/env python3
"""Embedding-width sweep on one graph: the heir of the reference's tuner (autotuner/findbestk.py:34-38 runs the
FusedMM timer for K in 16..1024 on `<dataset>.mtx` and prints a table of speedups over the generic kernel,
README.md:120-142).  Here, per K: the plain row-per-wave kernel (no per-graph preparation) against the
best of the task-list and stream schedules over the geometries `iSpLibPlugin.autotune` sweeps (each really forced),
through the plug-in surface.

    python scripts/findbestk.py graph.mtx                 # a MatrixMarket adjacency (README.md:147-168)
    python scripts/findbestk.py --workload reddit         # the synthetic Reddit-shaped graph of bench.py
    python scripts/findbestk.py graph.mtx --reduce max --save tuning.json    # ISPLIB_TUNE_FILE=tuning.json later
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("mtx", nargs="?", help="MatrixMarket coordinate file")
    p.add_argument("--workload", choices=("reddit", "cora", "products"), help="synthetic graph instead of a file")
    p.add_argument("--scale", type=float, default=1.0)
    p.add_argument("--reduce", default="sum", choices=("sum", "mean", "max", "min"))
    p.add_argument("--ks", default="16,32,64,128,256,512,1024")      # findbestk.py:34
    p.add_argument("--candidates", default="0,2,4,6,8,12,16,24")
    p.add_argument("--save", help="merge the winning (schedule, geometry) choices into this JSON tuning file")
    a = p.parse_args()
    if (a.mtx is None) == (a.workload is None):
        p.error("give a .mtx file or --workload")
    import isplib_amd
    from isplib_amd import synth
    dev = torch.device("cuda", 0)
    if a.mtx:
        adj = isplib_amd.SparseTensor.from_mtx(a.mtx, device=dev)
        name = os.path.basename(a.mtx)
    else:
        rowptr, col, n = synth.dataset_like(a.workload, device=dev, scale=a.scale)
        adj = isplib_amd.SparseTensor.from_csr(rowptr, col, None, (n, n), validate=False)
        name = f"{a.workload}-like"
    m, n = adj.sparse_sizes()
    nnz = adj.nnz()
    print(f"==For dataset: {name}===  M={m} N={n} nnz={nnz} reduce={a.reduce}")
    print(f"{'K':>6} {'plain ms':>10} {'tasks ms':>10} {'stream ms':>10} {'tuned ms':>10} {'tuned schedule':>26} {'speedup':>8} {'G edges/s':>10}")
    cands = tuple(int(c) for c in a.candidates.split(","))
    best = None
    for k in (int(t) for t in a.ks.split(",")):
        try:
            times = isplib_amd.iSpLibPlugin.autotune(adj, k, a.reduce, candidates=cands if 0 in cands else (0,) + cands)
        except torch.OutOfMemoryError:
            print(f"{k:>6} out of memory")
            break
        plain = times.get(("plain",))               # the plain row-per-wave kernel, really forced (not whatever the rule picks)
        if plain is None:
            continue
        s_best = min(times, key=times.get)
        col_of = lambda kind: min((t for c, t in times.items() if c[0] == kind), default=float("nan"))  # noqa: E731
        rate = nnz / (times[s_best] * 1e-3) / 1e9
        label = s_best[0] + ("" if len(s_best) == 1 else " " + ":".join(str(v) for v in s_best[1:]))
        print(f"{k:>6} {plain:>10.3f} {col_of('tasks'):>10.3f} {col_of('stream'):>10.3f} {times[s_best]:>10.3f} {label:>26} "
              f"{plain / times[s_best]:>8.2f} {rate:>10.2f}")
        # the reference's advice is "choose the K with the highest speedup" (README.md:142); report the best rate too
        if best is None or plain / times[s_best] > best[1]:
            best = (k, plain / times[s_best])
    if best:
        print(f"highest speedup over the plain kernel: K={best[0]} ({best[1]:.2f}x)")
    if a.save:
        isplib_amd.iSpLibPlugin.save_tuning(a.save)
        print(f"tuning table merged into {a.save}")


if __name__ == "__main__":
    main()
