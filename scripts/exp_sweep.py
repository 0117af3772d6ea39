"""Experiment: the sweep schedule (rows resident in LDS, waves walk the slices together) against the task list on the
Reddit-shaped graph.  usage: exp_sweep.py [k] [reduce] ; prints ms per launch for a grid of plan geometries."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isplib_amd import cabi, synth
from isplib_amd.plan import build_stream_plan, build_sweep_plan, build_task_plan

dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
red = sys.argv[2] if len(sys.argv) > 2 else "sum"
workload = os.environ.get("WORKLOAD", "reddit")
gen = os.environ.get("GENERATOR", "chunglu")
if gen == "chunglu":
    rowptr, col, n = synth.dataset_like(workload, device=dev)
else:
    n, target = synth.SHAPES[workload][0], synth.SHAPES[workload][1]
    rowptr, col = (synth.rmat_csr if gen == "rmat" else synth.uniform_csr)(n, target, device=dev)
nnz = col.numel()
x = synth.features(n, k, device=dev)
col32 = cabi.pack_indices(col)


def timeit(fn, it=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


msg = cabi.MESSAGE[red]
out = torch.empty((n, k), device=dev)
arg = torch.empty((n, k), dtype=torch.int64, device=dev) if red in ("max", "min") else None
s_def = int(cabi.lib().isplib_suggest_slices(n, n, nnz, k, int(red in ("max", "min"))))
only = os.environ.get("ONLY_SWEEP") == "1"
ref = None
if only:
    pass
elif s_def > 0:
    tp = build_task_plan(rowptr, col, n, s_def, col32=col32)
    tw = tp.workspace(red, k)
    t = timeit(lambda: cabi.fusedMM_csr_tasks_hip(msg, rowptr, col, None, tp, x, out, arg, tw))
    print(f"task list S={s_def}: {t:.3f} ms  {nnz / t / 1e6:.2f} Gedges/s", flush=True)
    ref = out.clone()
    del tp, tw
else:
    cabi.fusedMM_csr_hip(msg, rowptr, col, None, x, out, arg)
    t = timeit(lambda: cabi.fusedMM_csr_hip(msg, rowptr, col, None, x, out, arg))
    print(f"plain: {t:.3f} ms  {nnz / t / 1e6:.2f} Gedges/s", flush=True)
    ref = out.clone()

# stream form: STREAMS=slices:streams:chunk,...   (rows per wave and waves per generation: the kernel's own geometry)
sgeoms = os.environ.get("STREAMS", "32:4:512,64:4:512,128:4:512" if red in ("sum", "mean") else "")
for sg in filter(None, sgeoms.split(",")):
    S, streams, chunk = (int(v) for v in sg.split(":"))
    rpw, wpg = cabi.stream_geometry(streams)
    wpg = int(os.environ.get("WPG", wpg))
    torch.cuda.synchronize()
    plan = build_stream_plan(rowptr, col, None, n, S, wpg, rpw, streams, chunk)
    ws = plan.workspace()
    steps = plan.wave_step_off[1:] - plan.wave_step_off[:-1]
    cabi.fusedMM_csr_stream_hip(msg, rowptr, nnz, plan, x, out, ws)
    torch.cuda.synchronize()
    err = (out - ref).abs().max().item() if ref is not None else float("nan")
    t = timeit(lambda: cabi.fusedMM_csr_stream_hip(msg, rowptr, nnz, plan, x, out, ws))
    print(f"stream S={S:3d} rows/wave={rpw:2d} streams={streams} chunk={chunk} gens={plan.gens} waves/gen={wpg} steps={plan.n_steps} "
          f"(padding {plan.n_steps * streams / nnz - 1:.3%}) hub parts={plan.n_parts} steps max/mean={steps.max().item() / steps.double().mean().item():.3f}: "
          f"{t:.3f} ms  {nnz / t / 1e6:.2f} Gedges/s  maxdiff vs tasks {err:.2e}", flush=True)
    del plan, ws

# max / min on the stream form: MMSTREAMS=slices:chunk,...
if red in ("max", "min"):
    ref_arg = arg.clone() if ref is not None else None
    for sg in filter(None, os.environ.get("MMSTREAMS", "16:0,32:0").split(",")):
        S, chunk = (int(v) for v in sg.split(":"))
        streams = 8 if k <= 32 else 4
        rpw, wpg = cabi.stream_minmax_geometry(streams)
        wpg = int(os.environ.get("WPG", wpg))
        if chunk <= 0:                                     # the rule of isplib_suggest_stream
            gens_est = -(-n // (rpw * wpg))
            chunk = max(256, int(nnz / (gens_est * wpg * streams) / 3.4))
        torch.cuda.synchronize()
        mm_val = synth.edge_weights(nnz, device=dev) if os.environ.get("WEIGHTED") == "1" else None   # (then not comparable with the task list above)
        plan = build_stream_plan(rowptr, col, mm_val, n, S, wpg, rpw, streams, chunk, minmax=True)
        ws = plan.workspace(minmax=True)
        cabi.fusedMM_csr_stream_minmax_hip(msg, rowptr, nnz, plan, x, out, arg, ws)
        torch.cuda.synchronize()
        same = bool(torch.equal(out, ref) and torch.equal(arg, ref_arg)) if ref is not None else None
        t = timeit(lambda: cabi.fusedMM_csr_stream_minmax_hip(msg, rowptr, nnz, plan, x, out, arg, ws))
        t0 = timeit(lambda: cabi.fusedMM_csr_stream_minmax_hip(msg, rowptr, nnz, plan, x, out, None, ws))
        print(f"stream {red} S={S:3d} rows/wave={rpw:2d} chunk={chunk} gens={plan.gens} waves/gen={wpg} steps={plan.n_steps} "
              f"(padding {plan.n_steps * streams / nnz - 1:.3%}) hub parts={plan.n_parts}: {t:.3f} ms ({t0:.3f} without arg)  "
              f"{nnz / t / 1e6:.2f} Gedges/s  identical to tasks: {same}", flush=True)
        del plan, ws


geoms = os.environ.get("GEOMS")
if geoms == "none":
    geoms = []
elif geoms:
    geoms = [tuple(int(v) for v in g.split(":")) for g in geoms.split(",")]
else:
    geoms = [(s, rpw, 2048, ms) for s in (8, 16, 32) for rpw in (16,) for ms in (16,)] + [(16, 32, 2048, 16), (16, 16, 2048, 8), (16, 16, 1024, 32), (24, 16, 2048, 16)]
for (S, rpw, chunk, min_seg) in geoms:
    if red in ("max", "min") and rpw > 16:
        continue
    wpg = cabi.sweep_resident_waves(red, k, rpw)
    wpg = int(os.environ.get("WPG", wpg))
    torch.cuda.synchronize()
    plan = build_sweep_plan(rowptr, col, n, S, wpg, rpw, chunk, min_seg, col32=col32)
    ws = plan.workspace(red, k)
    loads = torch.zeros(plan.gens * plan.waves_per_gen, dtype=torch.int64, device=dev)
    cnt = plan.wave_task_off[1:] - plan.wave_task_off[:-1]
    lens = (plan.task_meta & 0xFFFFFF).to(torch.int64)
    loads.index_add_(0, torch.repeat_interleave(torch.arange(loads.numel(), device=dev), cnt), lens)
    cabi.fusedMM_csr_sweep_hip(msg, rowptr, col, None, plan, x, out, arg, ws)
    torch.cuda.synchronize()
    err = (out - ref).abs().max().item() if ref is not None else float("nan")
    t = timeit(lambda: cabi.fusedMM_csr_sweep_hip(msg, rowptr, col, None, plan, x, out, arg, ws))
    print(f"sweep S={S:3d} rows/wave={rpw:2d} chunk={chunk} min_seg={min_seg:2d} gens={plan.gens} waves/gen={wpg} tasks={plan.n_tasks} "
          f"(avg {nnz / max(plan.n_tasks, 1):.1f} edges) hub parts={plan.n_parts} load max/mean={loads.max().item() / loads.double().mean().item():.3f}: "
          f"{t:.3f} ms  {nnz / t / 1e6:.2f} Gedges/s  maxdiff vs tasks {err:.2e}", flush=True)
    del plan, ws
