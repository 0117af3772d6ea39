"""Round-4 experiments on the Reddit-shaped graph (one MI355X), each checked before it is timed:
  mm     : max / min on the stream schedule (row pair carried in registers): slices x chunk sweep at K = 64 / 128 / 32,
           values and arg bit for bit against the task list first
  sddmm  : task-list SDDMM in whole rows against 64-column panels (isplib_hip_tune_experimental(12, cols), libisplib_hip_exp.so), several slice counts
  w128   : weighted SpMM-sum K=128: the rule's plan against 128-column slots (2 streams), alternating, same box
  k41    : K=41 at its packed 164-byte pitch against 192- / 256-byte pitches (a copy of Y), 16-lane slots
usage: exp_round4.py mm,sddmm,w128,k41"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_stream_plan, build_task_plan

dev = torch.device("cuda:0")
what = (sys.argv[1] if len(sys.argv) > 1 else "mm,sddmm,w128,k41").split(",")
rowptr, col, n = synth.dataset_like("reddit", device=dev)
nnz = col.numel()
w = synth.edge_weights(nnz, device=dev)


def timeit(fn, it=10, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


if "mm" in what:
    col32 = cabi.pack_indices(col)
    for k, weighted in [(int(v.split(":")[0]), v.endswith("w")) for v in os.environ.get("MM_CASES", "64:w,64:u,128:w,32:u,41:u").split(",")]:
        val = w if weighted else None
        x = synth.features(n, k, device=dev, integer=False)
        z = torch.empty((n, k), device=dev)
        arg = torch.empty((n, k), dtype=torch.int64, device=dev)
        msg = cabi.MESSAGE["max"]
        from isplib_amd.plugin import suggest_slices
        sl = suggest_slices(n, n, nnz, k, True)
        tp = build_task_plan(rowptr, col, n, sl, col32=col32)
        tws = tp.workspace("max", k)
        cabi.fusedMM_csr_tasks_hip(msg, rowptr, col, val, tp, x, z, arg, tws)
        want, want_arg = z.clone(), arg.clone()
        t_tasks = timeit(lambda: cabi.fusedMM_csr_tasks_hip(msg, rowptr, col, val, tp, x, z, arg, tws), 5, 2)
        del tp, tws
        streams, r_slices, r_chunk = cabi.suggest_stream_minmax(n, n, nnz, k)
        print(f"[mm] K={k} {'weighted' if weighted else 'unit'}: task list {t_tasks:.3f} ms; rule {streams}:{r_slices}:{r_chunk}, geometry {cabi.stream_minmax_geometry(streams)}", flush=True)
        grid = [int(v) for v in os.environ.get("SLICES", "8,12,16,24,31,40,48").split(",")]
        chunks = [int(v) for v in os.environ.get("CHUNKS", f"{r_chunk},{r_chunk // 2}").split(",")]
        for s_ in grid:
            for ch in chunks:
                plan = build_stream_plan(rowptr, col, val, n, s_, None, None, streams, ch, minmax=True)
                ws = plan.workspace(minmax=True)
                z.zero_(); arg.zero_()
                cabi.fusedMM_csr_stream_minmax_hip(msg, rowptr, nnz, plan, x, z, arg, ws)
                ok = torch.equal(z, want) and torch.equal(arg, want_arg)
                ms = timeit(lambda: cabi.fusedMM_csr_stream_minmax_hip(msg, rowptr, nnz, plan, x, z, arg, ws))
                print(f"[mm] K={k} {'w' if weighted else 'u'} {s_:3d} slices chunk {ch:5d} gens {plan.gens} parts {plan.n_parts:6d}: {ms:.3f} ms  {'bit-exact' if ok else 'MISMATCH'}", flush=True)
                del plan, ws
        del x, z, arg, want, want_arg
    del col32

if "mmv" in what:       # max / min with and without the winners' positions (z_arg = NULL: the values-only launch), the rule's plan
    for k, weighted in [(int(v.split(":")[0]), v.endswith("w")) for v in os.environ.get("MM_CASES", "64:w,64:u,128:w,32:u").split(",")]:
        val = w if weighted else None
        x = synth.features(n, k, device=dev, integer=False)
        z = torch.empty((n, k), device=dev)
        z2 = torch.empty((n, k), device=dev)
        arg = torch.empty((n, k), dtype=torch.int64, device=dev)
        streams, r_slices, r_chunk = cabi.suggest_stream_minmax(n, n, nnz, k)
        plan = build_stream_plan(rowptr, col, val, n, r_slices, None, None, streams, r_chunk, minmax=True)
        ws = plan.workspace(minmax=True)
        for red in ("max", "min"):
            msg = cabi.MESSAGE[red]
            cabi.fusedMM_csr_stream_minmax_hip(msg, rowptr, nnz, plan, x, z, arg, ws)
            cabi.fusedMM_csr_stream_minmax_hip(msg, rowptr, nnz, plan, x, z2, None, ws)
            same = torch.equal(z.view(torch.int32), z2.view(torch.int32))
            t_arg = timeit(lambda: cabi.fusedMM_csr_stream_minmax_hip(msg, rowptr, nnz, plan, x, z, arg, ws))
            t_val = timeit(lambda: cabi.fusedMM_csr_stream_minmax_hip(msg, rowptr, nnz, plan, x, z2, None, ws))
            print(f"[mmv] K={k} {'w' if weighted else 'u'} {red} rule {streams}:{r_slices}:{r_chunk}: with arg {t_arg:.3f} ms, values only {t_val:.3f} ms  {'same values' if same else 'VALUES DIFFER'}", flush=True)
        del plan, ws, x, z, z2, arg

if "sddmm" in what:
    L = cabi.lib()
    for k in (128, 64, 256):
        x = synth.features(n, k, device=dev)
        g = synth.features(n, k, seed=5, device=dev)
        base_sl = max(1, L.isplib_suggest_slices_whole_rows(n, n, nnz, k))
        ref = None
        for sl in sorted({base_sl, max(1, base_sl // 2), base_sl * 3 // 2, base_sl * 2}):
            tp = build_task_plan(rowptr, col, n, sl)
            for cols in (0, 64, 128):
                if cols and k < 2 * cols:
                    continue
                cabi.exp_lib().isplib_hip_tune_experimental(12, cols)
                out = cabi.sddmm_tasks(rowptr, col, tp, x, g)
                if ref is None:
                    ref = out.clone()
                    mag = cabi.sddmm_tasks(rowptr, col, tp, x.abs(), g.abs())
                err = float(((out - ref).abs() / (mag + 1e-30)).max())
                ms = timeit(lambda: cabi.sddmm_tasks(rowptr, col, tp, x, g), 5, 2)
                print(f"[sddmm] K={k} {sl:3d} slices ({tp.n_tasks} tasks) panels of {cols or k:3d}: {ms:.3f} ms  max err / sum|x||g| {err:.2e}", flush=True)
            del tp
        cabi.exp_lib().isplib_hip_tune_experimental(12, 0)
        del x, g, ref, mag

if "w128" in what:
    msg = cabi.MESSAGE["sum"]
    for k in (128, 256):
        x = synth.features(n, k, device=dev)
        z = torch.empty((n, k), device=dev)
        st, sl, ch = cabi.suggest_stream(n, n, nnz, k)
        plans = {f"rule {st}:{sl}:{ch}": build_stream_plan(rowptr, col, w, n, sl, None, None, st, ch)}
        for s2 in (63, 72):
            plans[f"2:{s2}:{ch}"] = build_stream_plan(rowptr, col, w, n, s2, None, None, 2, ch)
        wss = {name: p.workspace() for name, p in plans.items()}
        for rnd in range(3):
            line = []
            for name, p in plans.items():
                ms = timeit(lambda: cabi.fusedMM_csr_stream_hip(msg, rowptr, nnz, p, x, z, wss[name]), 20, 3)
                line.append(f"{name} {ms:.3f}")
            print(f"[w128] weighted sum K={k}, round {rnd}: " + " | ".join(line), flush=True)
        del plans, wss, x, z

if "k41" in what:
    k = 41
    msg = cabi.MESSAGE["sum"]
    x = synth.features(n, k, device=dev)
    z = torch.empty((n, k), device=dev)
    st, sl, ch = cabi.suggest_stream(n, n, nnz, k)
    plan = build_stream_plan(rowptr, col, None, n, sl, None, None, st, ch)
    ws = plan.workspace()
    for pitch in (41, 44, 48, 64):
        xp = torch.zeros((n, pitch), device=dev)
        xp[:, :k] = x
        view = xp[:, :k]
        ms = timeit(lambda: cabi.fusedMM_csr_stream_hip(msg, rowptr, nnz, plan, view, z, ws), 20, 3)
        cp = timeit(lambda: xp[:, :k].copy_(x), 20, 3)
        print(f"[k41] K=41 sum, {st}:{sl}:{ch}, row pitch {pitch * 4} B: {ms:.3f} ms (+ {cp:.3f} ms for the copy)", flush=True)
        del xp, view
