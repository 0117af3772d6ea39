"""GPU tests of the row-partitioned (multi-rank) path.  The box has ONE GPU, so the two ranks share it and
talk over gloo (the collective itself is torch.distributed's; RCCL timing is the driver's job): what is
checked is everything around it -- partition, padded gather layout, overlapped phase schedule, autograd."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import oracle
from isplib_amd.dist import DistGraph
from tests import cases
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
n, k = 3000, 32
rowptr, col = cases.random_csr(n, n, 90.0, seed=5, empty_rows=(0, 1500), hub=(7, 2900))
val = cases.weights(col.size, 4)
x, g = cases.dense(n, k, 3), cases.dense(n, k, 5)
t = lambda a: torch.from_numpy(a).to(dev)
# plain gather-then-SpMM, then the three schedules of a sliced graph (ISPLIB_DIST_SCHEDULE)
for slices, mode in (("0", "tasks"), ("8", "tasks"), ("8", "overlap"), ("6", "pipelined"), ("8", "direct")):
    os.environ["ISPLIB_DIST_SCHEDULE"] = mode
    os.environ["ISPLIB_SLICES"] = slices            # the one-pass sliced kernel of `overlap` has its own rule: force it
    import isplib_amd.plugin as plugin
    plugin.suggest_slices = (lambda *a, **kw: int(slices))
    graph = DistGraph(t(rowptr), t(col), t(val), n, rank, world)
    r0, r1 = graph.row0, graph.row0 + graph.rows
    xs = t(x[r0:r1].copy()).requires_grad_(True)
    out = graph.matmul(xs)
    out.backward(t(g[r0:r1].copy()))
    torch.cuda.synchronize()
    ref, _ = oracle.spmm_fw(rowptr, col, val, x, "sum")
    mag, _ = oracle.spmm_fw(rowptr, col, np.abs(val), np.abs(x), "sum")
    assert np.all(np.abs(out.detach().cpu().numpy() - ref[r0:r1]) <= 1e-5 * mag[r0:r1] + 1e-30), (slices, mode)
    dref = oracle.spmm_sum_bw(rowptr, col, val, n, g)
    dmag = oracle.spmm_sum_bw(rowptr, col, np.abs(val), n, np.abs(g))
    assert np.all(np.abs(xs.grad.cpu().numpy() - dref[r0:r1]) <= 1e-5 * dmag[r0:r1] + 1e-30), (slices, mode)
os.environ.pop("ISPLIB_SLICES", None)
# every reduction with autograd under the partition (DistGraph.matmul(x, reduce)): default exchange (one all-gather + the
# schedule the single-GPU rules pick for the shard; forced onto this small graph: stream plans, then the task list) and the
# overlapped / direct exchanges.  Forward rows bit for bit the single-device rows for max / min (values AND global
# positions), within the bound for sum / mean; gradients against the oracle's backward formulas.
os.environ["ISPLIB_DIST_SCHEDULE"] = "tasks"
import isplib_amd.cabi as cabi_mod
rule, rule_mm = cabi_mod.suggest_stream, cabi_mod.suggest_stream_minmax
for forced in ("stream", "tasks", "plain", "overlap", "direct"):
    cabi_mod.suggest_stream = (lambda m_, n_, e_, k_, w_=False: (4, 3, 200)) if forced == "stream" else (lambda *a_, **kw_: None)
    cabi_mod.suggest_stream_minmax = (lambda m_, n_, e_, k_: (4, 3, 200)) if forced == "stream" else (lambda *a_, **kw_: None)
    plugin.suggest_slices = (lambda *a_, **kw_: 0 if forced == "plain" else 6)
    os.environ["ISPLIB_DIST_SCHEDULE"] = forced if forced in ("overlap", "direct") else "tasks"
    for weighted in (True, False):
        vv = val if weighted else None
        ones = np.ones(col.size, np.float32)
        graph = DistGraph(t(rowptr), t(col), None if vv is None else t(vv), n, rank, world)
        r0, r1 = graph.row0, graph.row0 + graph.rows
        for red in ("sum", "mean", "max", "min"):
            kind, _, _ = graph.fwd.local_ops(k, red)
            if forced in ("stream", "tasks", "plain"):
                assert kind == forced, (forced, red, kind)
            xs = t(x[r0:r1].copy()).requires_grad_(True)
            out = graph.matmul(xs, red)
            out.backward(t(g[r0:r1].copy()))
            torch.cuda.synchronize()
            w_ = vv if weighted else ones
            ref, ref_arg = oracle.spmm_fw(rowptr, col, w_, x, red)
            mag, _ = oracle.spmm_fw(rowptr, col, np.abs(w_), np.abs(x), "sum")
            if red in ("max", "min"):
                assert np.array_equal(out.detach().cpu().numpy(), ref[r0:r1]), (forced, weighted, red)
                _, garg = graph.fwd.spmm_auto(t(x[r0:r1].copy()), red)
                assert np.array_equal(garg.cpu().numpy(), ref_arg[r0:r1]), (forced, weighted, red)
                dref = oracle.spmm_minmax_bw(col, w_, x, ref_arg, g)[1]
                dmag = oracle.spmm_minmax_bw(col, np.abs(w_), x, ref_arg, np.abs(g))[1]
            else:
                if red == "mean":
                    mag = mag / np.maximum(np.diff(rowptr), 1)[:, None]
                assert np.all(np.abs(out.detach().cpu().numpy() - ref[r0:r1]) <= 1e-5 * mag[r0:r1] + 1e-30), (forced, weighted, red)
                bw = oracle.spmm_sum_bw if red == "sum" else oracle.spmm_mean_bw
                dref, dmag = bw(rowptr, col, w_, n, g), bw(rowptr, col, np.abs(w_), n, np.abs(g))
            assert np.all(np.abs(xs.grad.cpu().numpy() - dref[r0:r1]) <= 1e-5 * dmag[r0:r1] + 1e-30), (forced, weighted, red)
cabi_mod.suggest_stream, cabi_mod.suggest_stream_minmax = rule, rule_mm
os.environ["ISPLIB_DIST_SCHEDULE"] = "tasks"
# pipelined K-panel schedule: bitwise the task-list SpMM run panel by panel after one gather (max/min: also
# bitwise the unpanelled call; sums differ from it in the last bits, the slots per wave depend on the width)
from isplib_amd import cabi
from isplib_amd.dist import RowPartition
part = RowPartition(t(rowptr), t(col), t(val), n, rank, world)
tplan = part.task_plan(6, chunk=128, short_row=32)
for kk in (32, 41):
    xk = cases.dense(n, kk, 9)
    shard = part.shard(t(xk))
    buf = part.gather_buffer(kk)
    for red in ("sum", "max"):
        part.all_gather(shard, buf)
        want, want_arg = cabi.spmm_tasks(part.rowptr, part.col_padded, part.val, tplan, buf, red)
        for panels in (2, 3):
            state = part.pipeline_state(kk, panels, red, tplan)
            out = torch.zeros((part.rows, kk), device=dev)
            arg = torch.zeros((part.rows, kk), dtype=torch.int64, device=dev) if red == "max" else None
            part.spmm_pipelined(shard, out, state, red, arg)
            part.spmm_pipelined(shard, out, state, red, arg)      # buffers are reused across steps
            torch.cuda.synchronize()
            by_panel = torch.cat([cabi.spmm_tasks(part.rowptr, part.col_padded, part.val, tplan, buf[:, c0:c1].contiguous(), red)[0]
                                  for c0, c1 in state[0]], 1)
            assert torch.equal(out, by_panel), (kk, red, panels)
            if red == "max":
                assert torch.equal(out, want) and torch.equal(arg, want_arg), (kk, red, panels)
            else:
                mag, _ = oracle.spmm_fw(rowptr, col, np.abs(val), np.abs(xk), "sum")
                lim = torch.from_numpy(mag[part.row0:part.row0 + part.rows]).to(dev) * 1e-5 + 1e-30
                assert bool(((out - want).abs() <= lim).all()), (kk, red, panels)
        ref, _ = oracle.spmm_fw(rowptr, col, val, xk, red)
        if red == "max":
            assert np.array_equal(want.cpu().numpy(), ref[part.row0:part.row0 + part.rows])
# the same pipelined exchange with the stream schedule on every panel (sum / mean; the rule would decline a graph this
# small, so the plan parameters are given): panels narrower and wider than a slot, ragged widths, weights in the plan
for kk, geom in ((64, (4, 3, 200)), (41, (8, 2, 64)), (130, (2, 4, 500))):
    xk = cases.dense(n, kk, 13)
    shard = part.shard(t(xk))
    for red in ("sum", "mean"):
        state = part.pipeline_state(kk, 2, red, stream=geom)
        assert state is not None and hasattr(state[3], "words")
        out = torch.zeros((part.rows, kk), device=dev)
        part.spmm_pipelined(shard, out, state, red)
        first = out.clone()
        part.spmm_pipelined(shard, out, state, red)               # buffers are reused across steps
        torch.cuda.synchronize()
        assert torch.equal(first, out), (kk, red)
        ref, _ = oracle.spmm_fw(rowptr, col, val, xk, red)
        mag, _ = oracle.spmm_fw(rowptr, col, np.abs(val), np.abs(xk), "sum")
        sl = slice(part.row0, part.row0 + part.rows)
        assert np.all(np.abs(out.cpu().numpy() - ref[sl]) <= 1e-5 * mag[sl] + 1e-30), (kk, red)
# direct per-peer exchange (P-1 send / receive pairs in 1, 2 or P-1 groups, shards aggregated as they land): bitwise
# the column-sliced SpMM over the all-gathered buffer, every reduction, arg included
plan = part.plan(32, "sum", slices=2 * world)
xk = cases.dense(n, 32, 11)
shard_src, buf = part.shard(t(xk)), part.gather_buffer(32)
shard = torch.zeros_like(shard_src)
for red in ("sum", "mean", "max", "min"):
    part.all_gather(shard_src, buf)
    want, want_arg = cabi.spmm_sliced(part.rowptr, part.col_padded, part.val, plan[1], plan[0], buf, red)
    for nb in sorted({{1, 2, max(world - 1, 1)}}):
        # Regression for the one red run of round 2 (world 4, ('sum', 1)): gloo's send / receive of a device tensor
        # carry no stream or event, so they used to read the shard and write the buffer while the kernels producing
        # both were still queued.  Here those kernels are held back behind a long sleep on the compute stream -- the
        # shard is still zero and the NaN fill has not run when the exchange is posted -- and NOTHING in the test
        # waits for them: RowPartition.post_direct has to (it does, for every backend that is not stream-ordered).
        shard.zero_()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda._sleep(60_000_000)               # ~25-30 ms of the compute stream
        shard.copy_(shard_src)                      # the producer of what is sent
        buf.fill_(float("nan"))                     # nothing may be read before it has landed
        out = torch.zeros((part.rows, 32), device=dev)
        arg = torch.zeros((part.rows, 32), dtype=torch.int64, device=dev) if red in ("max", "min") else None
        part.spmm_direct(shard, buf, out, (plan[0], plan[1], cabi.sliced_workspace(red, part.rows, 32, plan[0], dev)), red, arg, batches=nb)
        torch.cuda.synchronize()
        assert torch.equal(out, want), (red, nb, int(torch.isnan(out).sum()), int((out != want).sum()))
        if arg is not None:
            assert torch.equal(arg, want_arg), (red, nb)
        dist.barrier()
    ref, _ = oracle.spmm_fw(rowptr, col, val, xk, red)
    if red in ("max", "min"):
        assert np.array_equal(want.cpu().numpy(), ref[part.row0:part.row0 + part.rows])
# ... and the same delayed producers with the guard taken out: the exchange then ships the zeroed shard (and the late NaN
# fill lands on top of what was received), which is what the guard is for.  One configuration, once.
import isplib_amd.dist as idist
guard = idist._p2p_is_stream_ordered
idist._p2p_is_stream_ordered = lambda group, device: True
try:
    part.all_gather(shard_src, buf)
    want, _ = cabi.spmm_sliced(part.rowptr, part.col_padded, part.val, plan[1], plan[0], buf, "sum")
    shard.zero_()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda._sleep(60_000_000)
    shard.copy_(shard_src)
    buf.fill_(float("nan"))
    out = torch.zeros((part.rows, 32), device=dev)
    part.spmm_direct(shard, buf, out, (plan[0], plan[1], cabi.sliced_workspace("sum", part.rows, 32, plan[0], dev)), "sum", None, batches=1)
    torch.cuda.synchronize()
    bad = int((out != want).sum())
    print("rank", rank, "unguarded exchange:", int(torch.isnan(out).sum()), "NaN,", bad, "mismatches")
    assert bad > 0, "the unguarded gloo exchange was expected to race with the queued producers"
finally:
    idist._p2p_is_stream_ordered = guard
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world", (2, 4))
def test_dist_graph_forward_backward_ranks_share_one_gpu(gpu, tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29547 + world), WORLD_SIZE=str(world), OMP_NUM_THREADS="4")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        assert f"rank {r} ok" in o


def test_dist_graph_world1_matches_plugin_autograd(gpu, oracle_mod):
    import isplib_amd
    from isplib_amd.dist import DistGraph
    rowptr, col = cases.random_csr(500, 500, 20.0, seed=2)
    x, g = cases.dense(500, 24, 3), cases.dense(500, 24, 5)
    d = lambda a: torch.from_numpy(a).to(gpu)  # noqa: E731
    graph = DistGraph(d(rowptr), d(col), None, 500, 0, 1)
    xs = d(x).requires_grad_(True)
    graph.matmul(xs).backward(d(g))
    adj = isplib_amd.SparseTensor.from_csr(d(rowptr), d(col), None, (500, 500))
    xs2 = d(x).requires_grad_(True)
    out2 = isplib_amd.matmul(adj, xs2)
    out2.backward(d(g))
    assert torch.allclose(graph.matmul(d(x)), out2.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(xs.grad, xs2.grad, rtol=1e-5, atol=1e-5)
