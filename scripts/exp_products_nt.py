#!/usr/bin/env python3
"""Experiment (round 5): config 4's shape with community structure (SBM twin), plain kernel in the
community order: 77 % of the stored entries lie within 1,024 positions of their row, the L2 hits 68 %.  Label propagation
finds the blocks exactly (2,449 labels of ~1,000 rows + singletons) and the 20 % of entries that leave a block are uniform
over the graph: there is no second level of structure to find.  What is left between 68 % and the ~78 % the order allows is
what the L2 KEEPS: the remote gathers (each a miss that allocates eight lines) and the finished rows (written once) pass
through the same 4 MiB the community's own rows live in.  Two cache-policy experiments on the plain kernel (experiment
builds: scripts/exp_variant.sh "-DISPLIB_EXP_NT_STORE=1" / "-DISPLIB_EXP_NT_REMOTE=2"):
  * finished rows stored non-temporally;
  * a bit per stored entry -- set where the column lies more than `window` positions from the row in the community order --
    selects the non-temporal policy for that gather (one row per gather instruction at K=256: a scalar branch).
Same bits in every variant (checked)."""
import ctypes
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


def remote_mask(rowptr, col, order, window):
    """uint64 words, bit e = 1 when |position(col[e]) - position(row(e))| > window in `order`."""
    n = rowptr.numel() - 1
    pos = torch.empty(n, dtype=torch.int64, device=col.device)
    pos[order.to(torch.int64)] = torch.arange(n, dtype=torch.int64, device=col.device)
    row = cabi.csr_row_ids(rowptr, col.numel())
    remote = ((pos[row] - pos[col]).abs() > window)
    del row, pos
    share = float(remote.sum()) / col.numel()
    pad = (-remote.numel()) % 64
    bits = torch.cat([remote, torch.zeros(pad, dtype=torch.bool, device=col.device)]).view(-1, 8).to(torch.uint8)
    weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=col.device)
    packed = (bits * weights).sum(1, dtype=torch.uint8)           # little-endian bytes: byte b holds bits 8b .. 8b+7
    return packed.contiguous(), share


def main():
    dev = torch.device("cuda:0")
    rowptr, col, n = synth.sbm_like("products", device=dev)
    order = reorder.useful_order(rowptr, col)
    assert order is not None
    k = 256
    x = synth.features(n, k, device=dev)
    z = torch.empty((n, k), dtype=torch.float32, device=dev)
    L = cabi.lib()
    have_mask = hasattr(L, "isplib_debug_set_nt_mask")
    run = lambda: cabi.fusedMM_csr_ordered_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, order, x, z)  # noqa: E731
    base = clock(run)
    want = z.clone()
    print(f"[nt] community order, K=256: {base:.3f} ms (this build: stores {'non-temporal' if os.environ.get('NT_STORE') else 'as built'})", flush=True)
    if have_mask:
        L.isplib_debug_set_nt_mask.argtypes = [ctypes.c_void_p]
        L.isplib_debug_set_nt_mask.restype = None
        for window in (1024, 2048, 8192):
            mask, share = remote_mask(rowptr, col, order, window)
            L.isplib_debug_set_nt_mask(ctypes.c_void_p(mask.data_ptr()))
            ms = clock(run)
            same = bool(torch.equal(z, want))
            print(f"[nt] remote gathers (> {window} positions away: {share * 100:.1f} % of the entries) non-temporal: {ms:.3f} ms  same bits: {same}", flush=True)
            L.isplib_debug_set_nt_mask(None)
            del mask
    # the index-order run (two 128-column panels) for reference
    ms = clock(lambda: cabi.fusedMM_csr_ordered_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, None, x, z))
    print(f"[nt] index order (panels): {ms:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
