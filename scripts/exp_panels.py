"""Experiment: the two 64-column panels of a K=128 stream launch do not cost the same (0.63 against 0.71 ms per dispatch).
Is it the address (+256 bytes into every 512-byte row) or the order (second panel of the launch)?  Times each half of X
on its own (a K=64 call on a view with ldy = 128), in both orders, and the same halves of a 640-byte-pitch copy."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_stream_plan

dev = torch.device("cuda:0")
rowptr, col, n = synth.dataset_like("reddit", device=dev)
nnz = col.numel()
streams, slices, chunk = cabi.suggest_stream(n, n, nnz, 128)
plan = build_stream_plan(rowptr, col, None, n, slices, None, None, streams, chunk)
ws = plan.workspace()


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


for pitch in (128, 160, 192, 256):
    xfull = torch.empty((n, pitch), device=dev)
    xfull[:, :128] = synth.features(n, 128, device=dev)
    zfull = torch.empty((n, pitch), device=dev)
    res = []
    for c0 in (0, 64, 32):
        xv, zv = xfull[:, c0:c0 + 64], zfull[:, c0:c0 + 64]
        res.append((c0, timeit(lambda: cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, xv, zv, ws))))
    both = timeit(lambda: cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, xfull[:, :128], zfull[:, :128], ws))
    print(f"row pitch {pitch * 4} B: " + "  ".join(f"columns {c0}..{c0 + 63}: {t:.3f} ms" for c0, t in res) + f"  | K=128 call: {both:.3f} ms", flush=True)
x64 = synth.features(n, 64, device=dev)
z64 = torch.empty((n, 64), device=dev)
print(f"K=64, pitch 256 B: {timeit(lambda: cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, x64, z64, ws)):.3f} ms")
# single 128-byte lines of the 512-byte rows (32-column panels on 8-lane slots)
st8, sl8, ch8 = cabi.suggest_stream(n, n, nnz, 32)
plan8 = build_stream_plan(rowptr, col, None, n, sl8, None, None, st8, ch8)
ws8 = plan8.workspace()
for pitch in (128, 256):
    xfull = torch.empty((n, pitch), device=dev)
    xfull[:, :128] = synth.features(n, 128, device=dev)
    zfull = torch.empty((n, pitch), device=dev)
    res = []
    for c0 in (0, 32, 64, 96, 16, 48, 80):
        xv, zv = xfull[:, c0:c0 + 32], zfull[:, c0:c0 + 32]
        res.append((c0, timeit(lambda: cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan8, xv, zv, ws8))))
    print(f"row pitch {pitch * 4} B, 32-column panels: " + "  ".join(f"{c0}: {t:.3f}" for c0, t in res), flush=True)
# 64-column panels at every 16-column offset of a 512-byte row
xfull = torch.empty((n, 256), device=dev)
xfull[:, :128] = synth.features(n, 128, device=dev)
xfull[:, 128:] = xfull[:, :128]
zfull = torch.empty((n, 256), device=dev)
xv512 = torch.empty((n, 128), device=dev).copy_(xfull[:, :128])
zv512 = torch.empty((n, 128), device=dev)
res = []
for c0 in (0, 16, 32, 48, 64):
    xv, zv = xv512[:, c0:c0 + 64], zv512[:, c0:c0 + 64]
    res.append((c0, timeit(lambda: cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, rowptr, nnz, plan, xv, zv, ws))))
print("row pitch 512 B, 64-column panels at column: " + "  ".join(f"{c0}: {t:.3f}" for c0, t in res), flush=True)
