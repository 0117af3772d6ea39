"""Experiment (round 4): the stream plan's rows in snake order (ISPLIB_EXP_SNAKE=1, torch builder): forwards in even slices,
backwards in odd ones -- one change of row per slice and stream disappears.  Sum K=128 (16 rows per stream) and max K=64
(8 rows per stream), each against the plain order, bit-identical results checked.  The measurement behind the rule that max /
min plans are snake-ordered and sum / mean plans are not (profiles/r04_experiments.txt); since that rule is in the builders,
the "plain order" column of the max rows now shows the snake order too.  usage: exp_snake.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_stream_plan

dev = torch.device("cuda:0")
rowptr, col, n = synth.dataset_like("reddit", device=dev)
nnz = col.numel()
w = synth.edge_weights(nnz, device=dev)


def timeit(fn, it=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


def plans(minmax, val, geom):
    out = {}
    for snake in ("0", "1"):
        os.environ["ISPLIB_EXP_SNAKE"] = snake
        out[snake] = build_stream_plan(rowptr, col, val, n, geom[1], None, None, geom[0], geom[2], minmax=minmax)
    os.environ["ISPLIB_EXP_SNAKE"] = "0"
    return out


for k, red, val in ((128, "sum", None), (64, "sum", None), (64, "max", w), (64, "max", None), (32, "max", None)):
    x = synth.features(n, k, device=dev)
    z = torch.empty((n, k), device=dev)
    arg = torch.empty((n, k), dtype=torch.int64, device=dev)
    mm = red == "max"
    geom = cabi.suggest_stream_minmax(n, n, nnz, k) if mm else cabi.suggest_stream(n, n, nnz, k, val is not None)
    ps = plans(mm, val, geom)
    ws = {s_: p.workspace(minmax=True) if mm else p.workspace() for s_, p in ps.items()}
    msg = cabi.MESSAGE[red]

    def run(s_):
        if mm:
            cabi.fusedMM_csr_stream_minmax_hip(msg, rowptr, nnz, ps[s_], x, z, arg, ws[s_])
        else:
            cabi.fusedMM_csr_stream_hip(msg, rowptr, nnz, ps[s_], x, z, ws[s_])
    run("0")
    ref, ref_arg = z.clone(), arg.clone()
    run("1")
    same = torch.equal(z, ref) and (not mm or torch.equal(arg, ref_arg))
    for rnd in range(3):
        t0, t1 = timeit(lambda: run("0")), timeit(lambda: run("1"))
        print(f"K={k} {red} {'weighted' if val is not None else 'unit'} {geom}: plain order {t0:.3f} ms, snake order {t1:.3f} ms ({'bit-identical' if same else 'DIFFERENT'})", flush=True)
    del ps, ws, x, z, arg
