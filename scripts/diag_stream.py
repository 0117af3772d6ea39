"""Diagnostic: where does the stream form differ from the plain kernel on a scaled Reddit-shaped graph?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_stream_plan
dev = torch.device("cuda:0")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
k = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rowptr, col, n = synth.dataset_like("reddit", device=dev, scale=scale)
nnz = col.numel()
x = synth.features(n, k, device=dev)
ref, _ = cabi.spmm(rowptr, col, None, x, "sum")
for (S, wpg, rpw, streams, chunk) in ((16, None, 16, 4, 512), (16, 64, 16, 4, 512), (16, None, 16, 4, 1 << 20)):
    plan = build_stream_plan(rowptr, col, None, n, S, wpg, rpw, streams, chunk)
    out = cabi.spmm_stream(rowptr, nnz, plan, x, "sum")
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ws = plan.workspace()
    s.record()
    for _ in range(3):
        cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, x, out, ws)
    e.record(); torch.cuda.synchronize()
    err = (out - ref).abs().max(dim=1).values
    bad = torch.nonzero(err > 1e-2).flatten()
    deg = rowptr[1:] - rowptr[:-1]
    steps = plan.wave_step_off[1:] - plan.wave_step_off[:-1]
    print(f"scale {scale} n={n} nnz={nnz} S={S} wpg={plan.waves_per_gen} gens={plan.gens} chunk={chunk} parts={plan.n_parts} hubs={plan.n_hub} "
          f"steps={plan.n_steps} max/mean={steps.max().item() / steps.double().mean().item():.3f}: {s.elapsed_time(e) / 3:.3f} ms; "
          f"bad rows {bad.numel()} of {n}; hub rows among bad: {int((deg[bad] > chunk).sum())}; first bad {bad[:8].tolist()} deg {deg[bad[:8]].tolist()}", flush=True)
    if bad.numel():
        r = int(bad[0])
        print("   row", r, "out", out[r, :4].tolist(), "ref", ref[r, :4].tolist())
