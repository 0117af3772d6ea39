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
for slices in ("0", "8"):                       # plain gather-then-SpMM, then the overlapped sliced schedule
    os.environ["ISPLIB_SLICES"] = slices
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
    assert np.all(np.abs(out.detach().cpu().numpy() - ref[r0:r1]) <= 1e-5 * mag[r0:r1] + 1e-30), slices
    dref = oracle.spmm_sum_bw(rowptr, col, val, n, g)
    dmag = oracle.spmm_sum_bw(rowptr, col, np.abs(val), n, np.abs(g))
    assert np.all(np.abs(xs.grad.cpu().numpy() - dref[r0:r1]) <= 1e-5 * dmag[r0:r1] + 1e-30), slices
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_dist_graph_forward_backward_two_ranks_on_one_gpu(gpu, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2", OMP_NUM_THREADS="4")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
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
