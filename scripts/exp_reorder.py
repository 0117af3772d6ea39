"""Experiment: the ogbn-products shape (N=2,449,029, nnz=123,718,280, K=256: X = 2.5 GB) on one GPU -- the plain kernel
against the same kernel with the rows in a community order (label propagation, isplib_amd/reorder.py), on a graph WITH block
structure (degree-corrected SBM, 2,449 blocks, 80 % of the edges inside) and on the structure-free Chung-Lu graph.
usage: exp_reorder.py [k] [p_in]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, reorder, synth

dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p_in = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8


def clock(fn, reps=5):
    for _ in range(2):
        fn()
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(reps):
        fn()
    e_.record()
    torch.cuda.synchronize()
    return s_.elapsed_time(e_) / reps


for name, make in (("SBM (2,449 blocks, p_in %.2f)" % p_in, lambda: synth.sbm_like("products", device=dev, p_in=p_in)),
                   ("Chung-Lu", lambda: synth.dataset_like("products", device=dev))):
    rowptr, col, n = make()
    nnz = col.numel()
    x = synth.features(n, k, device=dev)
    out = torch.empty((n, k), device=dev)
    t_plain = clock(lambda: cabi.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, x, out))
    want = out.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    order = reorder.community_order(rowptr, col)
    torch.cuda.synchronize()
    t_order = (time.perf_counter() - t0) * 1e3
    ident = torch.arange(n, dtype=torch.int32, device=dev)
    loc0, loc1 = reorder.ordered_gather_locality(rowptr, col, ident), reorder.ordered_gather_locality(rowptr, col, order)
    t_ord = clock(lambda: cabi.fusedMM_csr_ordered_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, order, x, out))
    same = bool(torch.equal(out, want))
    print(f"products shape, {name}, K={k}: plain {t_plain:.2f} ms, rows in community order {t_ord:.2f} ms ({1 - t_ord / t_plain:+.0%}), "
          f"bitwise equal: {same}; entries within 1024 positions of their row: {loc0:.1%} -> {loc1:.1%}; the order took {t_order:.0f} ms once",
          flush=True)
    del rowptr, col, x, out, want, order
    torch.cuda.empty_cache()
