#!/usr/bin/env python3
"""Experiment (round 5): config 4's shape WITH community structure (SBM twin), plain kernel in the
community order -- what the column-panel width of the row kernel does to the L2: K=256 in one pass (64 lanes per row: a
community's rows of y are 1 MB of an XCD's 4 MiB L2) against two 128-column passes and four 64-column passes
(isplib_hip_tune(0, lanes): the community's share of the L2 halves / quarters per pass, the index stream is read once per
pass).  Same bits in every setting (torch.equal)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from isplib_amd import cabi, reorder, synth  # noqa: E402


def clock(fn, reps=5):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    dev = torch.device("cuda:0")
    ks = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256").split(",")]
    for tag, make in (("sbm", synth.sbm_like), ("chunglu", synth.dataset_like)):
        rowptr, col, n = make("products", device=dev)
        order = reorder.useful_order(rowptr, col)
        print(f"== products-like {tag}: N={n} nnz={col.numel()} order kept: {order is not None}", flush=True)
        for k in ks:
            x = synth.features(n, k, device=dev)
            z = torch.empty((n, k), dtype=torch.float32, device=dev)
            want = None
            for o, oname in ((None, "index order"), (order, "community order")):
                if o is None and want is not None:
                    continue
                for lanes in (0, 32, 16):
                    if lanes * 4 >= k and lanes != 0:
                        continue
                    cabi.lib().isplib_hip_tune(0, lanes)
                    ms = clock(lambda: cabi.fusedMM_csr_ordered_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, o, x, z))
                    if want is None:
                        want = z.clone()
                    same = bool(torch.equal(z, want))
                    print(f"   K={k} {oname:16s} panels of {lanes * 4 if lanes else k:3d} columns: {ms:7.3f} ms  same bits: {same}", flush=True)
                if order is None:
                    break
            cabi.lib().isplib_hip_tune(0, 0)
            del x, z, want
        del rowptr, col, order
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
