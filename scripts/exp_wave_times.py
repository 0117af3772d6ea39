"""Experiment (library built with -DISPLIB_EXP_WAVE_TIMES, e.g. scripts/exp_build_variants.sh "python3 scripts/exp_wave_times.py" "-DISPLIB_EXP_WAVE_TIMES=1"): when do the waves of a stream dispatch start, enter their loop,
leave it and finish?  Prints, per dispatch of one K=128 launch, the spread of those four times over the 2,048 waves
(s_memtime ticks of 10 ns).  What it answers: how much of a dispatch is ramp / tail rather than steady gathering."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_stream_plan

dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rowptr, col, n = synth.dataset_like("reddit", device=dev)
nnz = col.numel()
x = synth.features(n, k, device=dev)
out = torch.empty((n, k), device=dev)
streams, slices, chunk = cabi.suggest_stream(n, n, nnz, k)
plan = build_stream_plan(rowptr, col, None, n, slices, None, None, streams, chunk)
ws = plan.workspace()
L = cabi.lib()
L.isplib_debug_wave_times.argtypes = [ctypes.c_void_p]
L.isplib_debug_wave_times.restype = None
for _ in range(3):
    cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, x, out, ws)
torch.cuda.synchronize()
nw = plan.waves_per_gen
buf = torch.zeros(4 * nw * 4, dtype=torch.int64, device=dev)
L.isplib_debug_wave_times(ctypes.c_void_p(buf.data_ptr()))
cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, x, out, ws)
torch.cuda.synchronize()
L.isplib_debug_wave_times(None)
t = buf.view(4, nw, 4).cpu().double()
steps = (plan.wave_step_off[1:] - plan.wave_step_off[:-1]).cpu().double().view(-1, nw)
for d in range(4):
    a = t[d]
    if a.max() == 0:
        continue
    t0 = a[:, 0].min()
    us = lambda v: v * 0.01     # 100 MHz
    start, loop, loop_end, end = (a[:, i] - t0 for i in range(4))
    q = lambda v: "min %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % tuple(us(torch.quantile(v, torch.tensor([0.0, 0.5, 0.9, 0.99, 1.0], dtype=torch.float64))).tolist())
    print(f"dispatch {d}: {us(end.max()):.1f} us from the first wave's start to the last wave's end")
    print("   wave start      :", q(start))
    print("   loop entered    :", q(loop))
    print("   loop left       :", q(loop_end))
    print("   wave end        :", q(end))
    print("   in the loop     :", q(loop_end - loop))
    dur = loop_end - loop
    wg = torch.arange(nw) // 4
    per_xcd = [dur[(wg % 8) == x].mean().item() for x in range(8)]
    print("   loop ticks by XCD (workgroup % 8): " + " ".join(f"{v / dur.mean().item():.3f}" for v in per_xcd) + "  (relative to the mean)")
    st = steps[d % steps.shape[0]]
    c = torch.corrcoef(torch.stack([dur, st]))[0, 1].item()
    print(f"   mean loop / max loop = {dur.mean().item() / dur.max().item():.3f}; correlation of loop ticks with the wave's step count {c:.2f}; steps max/mean = {st.max() / st.mean():.4f}")
    cu = wg // 8          # the 32 CUs of an XCD get workgroups round-robin (two each)
    per_wg = dur.view(-1, 4)
    print(f"   spread inside a workgroup (max - min of its 4 waves) / mean: {((per_wg.max(1).values - per_wg.min(1).values).mean() / dur.mean()).item():.3f}", flush=True)
