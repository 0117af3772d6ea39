"""Experiment: ragged K (41: the GCN's class width) with the dense operand at its own pitch (164-byte rows straddle cache lines:
2.25 lines per row on average) against a padded pitch (192 / 256 bytes: 2 lines), copy included.  usage: exp_pitch.py [k]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_stream_plan

dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 41
rowptr, col, n = synth.dataset_like("reddit", device=dev)
nnz = col.numel()
x = synth.features(n, k, device=dev)
streams, slices, chunk = cabi.suggest_stream(n, n, nnz, k)
plan = build_stream_plan(rowptr, col, None, n, slices, None, None, streams, chunk)
ws = plan.workspace()
out = torch.empty((n, k), device=dev)


def clock(fn, reps=20):
    for _ in range(3):
        fn()
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(reps):
        fn()
    e_.record()
    torch.cuda.synchronize()
    return s_.elapsed_time(e_) / reps


print(f"K={k} ({streams} streams, {slices} slices) packed rows ({4 * k} B pitch): {clock(lambda: cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, x, out, ws)):.3f} ms", flush=True)
want = out.clone()
for pitch in (44, 48, 64):
    if pitch < k:
        continue
    buf = torch.zeros((n, pitch), device=dev)

    def run():
        buf[:, :k].copy_(x)
        cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, buf[:, :k], out, ws)
    t = clock(run)
    t0 = clock(lambda: cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, buf[:, :k], out, ws))
    print(f"K={k} rows at a {4 * pitch} B pitch: {t0:.3f} ms, with the copy into the padded buffer {t:.3f} ms; bitwise equal: {bool(torch.equal(out, want))}", flush=True)
