#!/usr/bin/env python3
"""Experiment (round 5): where the plain kernel's time goes on config 4's shape.  Same row structure (degrees), three
column patterns at K=256: every gather an L2 hit (column ids folded into 1,024 rows = 1 MB of y), the SBM twin in its community
order (77 % of the entries within 1,024 positions), the Chung-Lu graph (every gather a miss of every cache)."""
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
    k = 256
    rowptr, col, n = synth.sbm_like("products", device=dev)
    order = reorder.useful_order(rowptr, col)
    x = synth.features(n, k, device=dev)
    z = torch.empty((n, k), dtype=torch.float32, device=dev)
    for lanes in (64, 32):
        cabi.lib().isplib_hip_tune(0, lanes)
        folded = (col % 1024).contiguous()
        t_hit = clock(lambda: cabi.fusedMM_csr_ordered_hip(cabi.MSG_SPMM_SUM, rowptr, folded, None, None, x, z))
        t_ord = clock(lambda: cabi.fusedMM_csr_ordered_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, order, x, z))
        t_idx = clock(lambda: cabi.fusedMM_csr_ordered_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, None, x, z))
        print(f"[ceiling] {lanes} lanes per row: all gathers L2 hits {t_hit:.3f} ms | SBM, community order {t_ord:.3f} ms | SBM, index order {t_idx:.3f} ms", flush=True)
        del folded
    cabi.lib().isplib_hip_tune(0, 0)


if __name__ == "__main__":
    main()
