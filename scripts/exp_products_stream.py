"""Config 4 on one GPU, a graph WITH community structure: can the stream schedule serve it when the rows of a generation are a
contiguous run of the community order?  The stream rule refuses the products shape (19 generations would each sweep the whole
dense operand); but with the graph's rows AND column slices taken in the community order, a generation's edges fall mostly into
the few slices of its own diagonal block (~33 MB of a 64-column panel), which the L2s serve, and only the remote edges
(~23 % on the SBM twin) go to HBM -- with 32 gathers in flight per wave instead of the plain kernel's 4-8.
Here: the graph is RELABELLED in the community order (for the experiment; a plan could carry the order instead and leave the
operands where they are), cut into blocks of one generation's rows, and every block runs the existing stream entry on its own plan.
usage: exp_products_stream.py [sbm|chunglu] [k=256]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, reorder, synth
from isplib_amd.plan import build_stream_plan_native

dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "sbm"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rowptr, col, n = (synth.sbm_like if kind == "sbm" else synth.dataset_like)("products", device=dev)
nnz = col.numel()
order = reorder.useful_order(rowptr, col)
if order is None:
    print(f"[{kind}] no community order found: index order kept")
    order = torch.arange(n, device=dev, dtype=torch.int32)
order = order.to(torch.int64)
rank = torch.empty(n, dtype=torch.int64, device=dev)
rank[order] = torch.arange(n, device=dev)
deg = rowptr[1:] - rowptr[:-1]
row_of = torch.repeat_interleave(torch.arange(n, device=dev), deg)
key = rank[row_of] * n + rank[col]                       # (new row, new column), sorted: the relabelled CSR with sorted rows
key, _ = torch.sort(key)
col2 = (key % n).contiguous()
rowptr2 = torch.zeros(n + 1, dtype=torch.int64, device=dev)
rowptr2[1:] = torch.cumsum(deg[order], 0)
del key, row_of, rank
x = synth.features(n, k, device=dev)
z_ref = torch.empty((n, k), device=dev)
z = torch.zeros((n, k), device=dev)
msg = cabi.MSG_SPMM_SUM


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(reps):
        fn()
    e_.record()
    torch.cuda.synchronize()
    return s_.elapsed_time(e_) / reps


cabi.lib().isplib_hip_tune(0, 64)                        # one 256-column pass per row: the community-order form of the plain kernel
t_plain = timed(lambda: cabi.fusedMM_csr_hip(msg, rowptr2, col2, None, x, z_ref, None))
cabi.lib().isplib_hip_tune(0, 0)
print(f"[{kind}] K={k}: plain kernel on the relabelled graph (rows in community order, one pass): {t_plain:.3f} ms", flush=True)

for streams in (4, 2):
    rpw, resident = cabi.stream_geometry(streams)
    per_gen = rpw * resident
    panel = 1024 // streams
    for slice_mb in (1.9, 3.8, 7.6):
        slices = max(1, min(512, round(n * panel / (slice_mb * 1e6))))
        blocks, gens_max = [], 0
        rows_blk = int(per_gen * 0.85)                   # room for the virtual rows of cut hub rows: one generation per block
        for r0 in range(0, n, rows_blk):
            r1 = min(n, r0 + rows_blk)
            e0, e1 = int(rowptr2[r0]), int(rowptr2[r1])
            rp = (rowptr2[r0:r1 + 1] - e0).contiguous()
            cb = col2[e0:e1]
            chunk = max(256, int((e1 - e0) / (resident * streams) / 3.4))
            plan = build_stream_plan_native(rp, cb, n, slices, streams, chunk)
            assert plan is not None
            gens_max = max(gens_max, plan.gens)
            blocks.append((rp, e1 - e0, plan, plan.workspace(), z[r0:r1]))

        def run():
            for rp, ne, plan, ws, zb in blocks:
                cabi.fusedMM_csr_stream_hip(msg, rp, ne, plan, x, zb, ws)
        t = timed(run)
        err = float(((z - z_ref).abs().amax() / z_ref.abs().amax()).item())
        print(f"[{kind}] K={k}: stream schedule per generation of the community order: {streams} streams, {len(blocks)} blocks of {rows_blk} rows (<= {gens_max} generation(s) each), "
              f"{slices} slices of {slice_mb} MB: {t:.3f} ms (max rel diff to the plain kernel {err:.1e})", flush=True)
        del blocks
        torch.cuda.empty_cache()
