"""Max/min backward at Reddit scale: the one-pass atomic scatter against the atomic-free (sort + run sums) form."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth

dev = torch.device("cuda:0")
rowptr, col, n = synth.dataset_like("reddit", device=dev)
for k in (64, 128):
    x = synth.features(n, k, device=dev)
    g = synth.features(n, k, seed=5, device=dev)
    out, arg = cabi.spmm(rowptr, col, None, x, "max")
    for need_val in (False, True):
        for det in (False, True):
            f = lambda: cabi.spmm_minmax_bw(col, None, x, arg, g, need_val=need_val, deterministic=det)  # noqa: E731
            f()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                f()
            e.record()
            torch.cuda.synchronize()
            print(f"K={k} {'dX + dval' if need_val else 'dX only'} {'atomic-free' if det else 'atomic':12s} {s.elapsed_time(e) / 5:.3f} ms", flush=True)
    a = cabi.spmm_minmax_bw(col, None, x, arg, g, deterministic=True)
    b = cabi.spmm_minmax_bw(col, None, x, arg, g, deterministic=True)
    c = cabi.spmm_minmax_bw(col, None, x, arg, g)
    print("atomic-free twice bitwise equal:", torch.equal(a[1], b[1]) and torch.equal(a[0], b[0]),
          "; max |atomic-free - atomic| dX:", (a[1] - c[1]).abs().max().item(), flush=True)
