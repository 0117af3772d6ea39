"""Experiment: the hybrid form of the stream schedule (hot rows of y served from an LDS table) against the stream form,
Reddit-shaped graph.  usage: exp_hybrid.py [k] ; HYB="streams:slices[:min_refs],..." (default: a sweep)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_hybrid_plan, build_stream_plan

dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rowptr, col, n = synth.dataset_like(os.environ.get("WORKLOAD", "reddit"), device=dev)
nnz = col.numel()
x = synth.features(n, k, device=dev)
msg = cabi.MSG_SPMM_SUM


def clock(fn, reps=10):
    for _ in range(3):
        fn()
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(reps):
        fn()
    e_.record()
    torch.cuda.synchronize()
    return s_.elapsed_time(e_) / reps


streams, slices, chunk = cabi.suggest_stream(n, n, nnz, k)
plan = build_stream_plan(rowptr, col, None, n, slices, None, None, streams, chunk)
ws = plan.workspace()
want = torch.empty((n, k), device=dev)
cabi.fusedMM_csr_stream_hip(msg, rowptr, nnz, plan, x, want, ws)
torch.cuda.synchronize()
print(f"K={k} stream form ({streams} streams, {slices} slices, chunk {chunk}): {clock(lambda: cabi.fusedMM_csr_stream_hip(msg, rowptr, nnz, plan, x, want, ws)):.3f} ms", flush=True)
mag = torch.empty((n, k), device=dev)
cabi.fusedMM_csr_stream_hip(msg, rowptr, nnz, plan, x.abs(), mag, ws)
bound = mag * 1e-5 + 1e-30
del plan, ws
default = "4:31,4:62,4:93,4:124,4:186,8:31,8:62,8:124,8:186,8:248"
for spec in os.environ.get("HYB", default).split(","):
    f = [int(v) for v in spec.split(":")]
    st, sl, refs = f[0], f[1], (f[2] if len(f) > 2 else 2)
    rpw, resident, ht, cap = cabi.hybrid_geometry(st)
    if len(f) > 4:                       # st:slices:refs:table_rows:hot_cap -- another geometry than the built kernel's (the second
        ht, cap = f[3], f[4]             # version of commit 810ada5 took 64 rows x 2 buffers, 16 steps; today's entry refuses it)
    chunk_h = max(256, int(nnz / (-(-n // (rpw * resident)) * resident * st) / 3.4))
    try:
        hp = build_hybrid_plan(rowptr, col, n, sl, st, chunk_h, min_refs=refs, table_rows=ht, hot_cap=cap)
        torch.cuda.synchronize()
        hws = hp.workspace()
        out = torch.empty((n, k), device=dev)
        cabi.fusedMM_csr_hybrid_hip(msg, rowptr, nnz, hp, x, out, hws)
        again = torch.empty((n, k), device=dev)
        cabi.fusedMM_csr_hybrid_hip(msg, rowptr, nnz, hp, x, again, hws)
        torch.cuda.synchronize()
        bad = int(((out - want).abs() > bound).sum())
        same = bool(torch.equal(out, again))
        t = clock(lambda: cabi.fusedMM_csr_hybrid_hip(msg, rowptr, nnz, hp, x, out, hws))
        hot_steps_max = int((hp.hot_step_off[1:] - hp.hot_step_off[:-1]).max())
        print(f"K={k} hybrid {st} streams, {sl} slices, table {ht} rows, min refs {refs}: {hp.hot_edges / nnz:.1%} of the edges from LDS, "
              f"gens {hp.cold.gens}, cold steps {hp.cold.n_steps}, hot steps {hp.n_hot_steps} (longest chunk {hot_steps_max}/{cap}), "
              f"{t:.3f} ms, outside the bound: {bad}, bitwise repeatable: {same}", flush=True)
        del hp, hws, out, again
    except Exception as e:  # noqa: BLE001
        print(f"K={k} hybrid {spec}: {type(e).__name__}: {e}", flush=True)
    torch.cuda.empty_cache()
