#!/usr/bin/env python3
"""Times every kernel on the path at Reddit scale (one MI355X): forward of the four reductions,
backward through autograd, SDDMM dA, the fused max/min backward, and the per-graph preparation."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import isplib_amd  # noqa: E402
from isplib_amd import cabi, synth  # noqa: E402

dev = torch.device("cuda:0")
rowptr, col, n = synth.dataset_like("reddit", device=dev)
nnz = col.numel()
w = synth.edge_weights(nnz, device=dev)


def timeit(fn, it=5, warm=2):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


res = {}
res["prep/csr2csc_ms"] = timeit(lambda: cabi.csr2csc(rowptr, col, w, n), 2, 1)
res["prep/slices16_ms"] = timeit(lambda: cabi.spmm_slices(rowptr, col, n, 16), 2, 1)
res["prep/row_ids_ms"] = timeit(lambda: cabi.csr_row_ids(rowptr, nnz), 2, 1)

for k in (64, 128):
    x = synth.features(n, k, device=dev)
    g = synth.features(n, k, seed=5, device=dev)
    for weighted in (False, True):
        adj = isplib_amd.SparseTensor.from_csr(rowptr, col, w if weighted else None, (n, n), validate=False)
        tag = f"k{k}/{'w' if weighted else 'unit'}"
        for red in ("sum", "mean", "max", "min"):
            res[f"{tag}/{red}/fwd_ms"] = timeit(lambda: isplib_amd.matmul(adj, x, red))

            def fb():
                xs = x.detach().requires_grad_(True)
                isplib_amd.matmul(adj, xs, red).backward(g)
            res[f"{tag}/{red}/fwd+bwd_ms"] = timeit(fb)
        del adj
    res[f"k{k}/sddmm_ms"] = timeit(lambda: cabi.sddmm(rowptr, col, x, g))
    from isplib_amd.plan import build_task_plan
    # the dot product needs whole rows: slices by the whole-row rule, not the SpMM's panel rule
    tp = build_task_plan(rowptr, col, n, max(1, cabi.lib().isplib_suggest_slices_whole_rows(n, n, nnz, k)))
    res[f"k{k}/sddmm_tasks_ms"] = timeit(lambda: cabi.sddmm_tasks(rowptr, col, tp, x, g))
    del tp
    out, arg = cabi.spmm(rowptr, col, w, x, "max")
    res[f"k{k}/minmax_bw_ms"] = timeit(lambda: cabi.spmm_minmax_bw(col, w, x, arg, g))
    del out, arg
for key, v in res.items():
    print(f"{key:34s} {v:9.3f} ms   {nnz / v / 1e6:8.2f} Gedges/s")
print(json.dumps(res))
