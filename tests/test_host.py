"""CPU tests of the host side (no GPU, no compute calls into the HIP library):
the C ABI library loads and exports every symbol include/isplib_hip.h declares,
the torch operator surface has the reference's names, CPU tensors are refused
loudly (no silent fallback), the plug-in patches/unpatches LIFO, and the 1-D row
partition is equivalent to the whole graph on a 2-rank gloo job.
"""
import os
import re
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "isplib_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|void|size_t|char)\s*\*?\s*(\w+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_declares_expected_entry_points():
    names = _declared_functions()
    for must in ("fusedMM_csr_hip", "fusedMM_csr_sliced_hip", "performDummySpMM_hip", "isplib_spmm_minmax_bw_hip",
                 "isplib_sddmm_csr_hip", "isplib_csr2csc_hip", "isplib_spmm_slices_build_hip"):
        assert must in names


def test_cabi_library_exports_every_declared_symbol():
    from isplib_amd import _lib, cabi
    assert os.path.exists(_lib.CABI_PATH), "libisplib_hip.so not built (python -c 'import __graft_entry__ as g; g.build()')"
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.CABI_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    declared = _declared_functions()
    missing = [n for n in declared if n not in exported]
    assert not missing, f"declared in include/isplib_hip.h but not exported: {missing}"
    assert sorted(cabi.EXPORTS) == declared, "isplib_amd.cabi.EXPORTS out of sync with the header"
    L = cabi.lib()
    for n in declared:
        getattr(L, n)
    assert L.isplib_hip_abi_version() == 1
    assert cabi.last_error() == ""


def test_experimental_library_exports_its_own_header_and_nothing_of_it_is_in_the_default_one():
    """include/isplib_hip_experimental.h <-> libisplib_hip_exp.so (the measured losers: sweep schedule, LDS hot-row hybrid,
    stream-plan SDDMM, their knobs): every declared symbol is exported there, none of them by the default library, and the
    default header declares none of them -- a binding of the reference's path never sees them."""
    import re
    from isplib_amd import _lib, cabi
    text = open(os.path.join(ROOT, "include", "isplib_hip_experimental.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"^\s*(?:const\s+)?(?:int|void|size_t|char)\s*\*?\s*(\w+)\s*\(", text, flags=re.M)))
    assert sorted(cabi.EXP_EXPORTS) == declared
    exp_path = os.path.join(os.path.dirname(_lib.CABI_PATH), "libisplib_hip_exp.so")
    assert os.path.exists(exp_path), "libisplib_hip_exp.so not built"
    sym = lambda path: {ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--defined-only", path], text=True).splitlines() if " T " in ln}  # noqa: E731
    assert not [n for n in declared if n not in sym(exp_path)]
    core = sym(_lib.CABI_PATH)
    assert not [n for n in declared if n in core], "an experimental entry is still exported by the default library"
    assert not set(declared) & set(_declared_functions())
    L = cabi.exp_lib()
    for n in declared:
        getattr(L, n)
    assert L.isplib_hip_tune_experimental(9, 64) == 0 and L.isplib_hip_tune_experimental(10, 1) == cabi.FAIL
    assert cabi.lib().isplib_hip_tune(9, 64) == cabi.FAIL, "knobs 9-12 left the default library"


def test_cabi_argument_validation_without_gpu():
    """Status codes that are decided before any HIP call."""
    from isplib_amd import cabi
    L = cabi.lib()
    st = L.fusedMM_csr_hip(0x11108, 1, 1, 1, 1.0, 0, 1, 1, None, None, None, None, None, 1, None, 1, 0.0, None, 1, None, None)
    assert st == cabi.NO_OPT_IMPL and "VOP" in cabi.last_error()          # a flag value the header does not define
    st = L.fusedMM_csr_hip(0x11F02, 1, 1, 1, 1.0, 0, 1, 1, None, None, None, None, None, 1, None, 1, 0.0, None, 1, None, None)
    assert st == cabi.UNDEFINED_USER_FUNCTION                             # SOP_UDEF through the entry without a menu
    st = L.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, 1, 1, 1, 1.0, 0, 1, 1, None, None, None, None, None, 1, None, 1, 0.5, None, 1, None, None)
    assert st == cabi.FAIL and "beta" in cabi.last_error()
    st = L.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, 0, 0, 4, 1.0, 0, 0, 0, None, None, None, None, None, 4, None, 4, 0.0, None, 4, None, None)
    assert st == cabi.SUCCESS            # m == 0: nothing to do
    st = L.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, 2, 1 << 31, 4, 1.0, 0, 2, 2, None, None, None, None, None, 4, None, 4, 0.0, None, 4, None, None)
    assert st == cabi.FAIL and "2^31" in cabi.last_error()
    assert L.isplib_spmm_slices_bytes(10, 8) == 10 * 9 * 8
    assert L.isplib_spmm_sliced_workspace_bytes(cabi.MSG_SPMM_SUM, 10, 4, 8) >= 8 * 10 * 4 * 4
    assert L.isplib_spmm_sliced_workspace_bytes(cabi.MSG_SPMM_MAX, 10, 4, 8) >= 2 * 8 * 10 * 4 * 4


def test_torch_ops_have_reference_names_and_schemas():
    import isplib_amd  # noqa: F401
    ops = torch.ops.isplib
    s = str(ops.fusedmm_spmm.default._schema)
    assert "Tensor? row, Tensor rowptr, Tensor col, Tensor? value, Tensor? colptr, Tensor? csr2csc, Tensor mat" in s
    assert "value_index_select" in s and "row_index_select" in s
    s = str(ops.fusedmm_spmm_mean.default._schema)
    assert "Tensor? rowcount" in s and "new_row" in s and "new_rowcount" in s
    for name in ("fusedmm_spmm_max", "fusedmm_spmm_min"):
        s = str(getattr(ops, name).default._schema)
        assert "Tensor rowptr, Tensor col, Tensor? value, Tensor mat" in s and "-> (Tensor, Tensor)" in s
    assert "int flag" in str(ops.performDummySpMM.default._schema)
    for name in ("fusedmm_spmm_max_values", "fusedmm_spmm_min_values"):      # not the reference's: the plug-in's no-gradient path
        s = str(getattr(ops, name).default._schema)
        assert "Tensor rowptr, Tensor col, Tensor? value, Tensor mat, Tensor[] plan" in s and s.endswith("-> Tensor")


def test_cpu_tensors_are_refused_not_silently_served():
    import isplib_amd
    rowptr, col = torch.tensor([0, 1, 2]), torch.tensor([0, 1])
    x = torch.ones(2, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        torch.ops.isplib.fusedmm_spmm(None, rowptr, col, None, None, None, x, None, None)
    with pytest.raises(RuntimeError, match="no CPU path"):
        torch.ops.isplib.fusedmm_spmm_max(rowptr, col, None, x)
    with pytest.raises(RuntimeError, match="no CPU path"):
        torch.ops.isplib.fusedmm_spmm_max_values(rowptr, col, None, x, [])
    adj = isplib_amd.SparseTensor.from_csr(rowptr, col, None, (2, 2))
    with pytest.raises(RuntimeError, match="no CPU path"):
        isplib_amd.matmul(adj, x)
    with pytest.raises(ValueError, match="unknown reduce"):
        isplib_amd.matmul(adj, x, "prod")
    with pytest.raises(RuntimeError, match="no CPU path"):
        isplib_amd.cabi.spmm(rowptr, col, None, x)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "isplib_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "fusedmm_oracle" not in text, f


def test_sparse_tensor_coo_to_csr_keeps_torch_sparse_order():
    import isplib_amd
    # README.md:105-110: duplicates (0,0) keep their input order (3 then -2)
    adj = isplib_amd.SparseTensor(row=torch.tensor([2, 0, 1, 0, 0]), col=torch.tensor([1, 0, 0, 2, 0]),
                                  value=torch.tensor([3., 3., 4., 2., -2.]), sparse_sizes=(3, 3))
    rowptr, col, val = adj.csr()
    e_rowptr, e_col, e_val, *_ = cases.readme_case()
    assert rowptr.tolist() == e_rowptr.tolist() and col.tolist() == e_col.tolist() and val.tolist() == e_val.tolist()
    assert adj.storage.rowcount().tolist() == [3, 1, 1]
    with pytest.raises(ValueError, match="out of range"):
        isplib_amd.SparseTensor(row=torch.tensor([0]), col=torch.tensor([5]), sparse_sizes=(2, 2))
    with pytest.raises(ValueError, match="monotone"):
        isplib_amd.SparseTensor(rowptr=torch.tensor([0, 2, 1]), col=torch.tensor([0]), sparse_sizes=(2, 2))


def test_patch_unpatch_is_lifo_and_restores(monkeypatch):
    import isplib_amd
    from isplib_amd import plugin
    fake_ts = types.SimpleNamespace(matmul=lambda *a, **k: "torch_sparse.matmul")
    fake_typing = types.SimpleNamespace(WITH_PT2=True, WITH_PT20=True)
    monkeypatch.setattr(plugin, "_torch_sparse", fake_ts)
    monkeypatch.setattr(plugin, "_pyg_typing", fake_typing)
    orig_ts, orig_mm = fake_ts.matmul, torch.sparse.mm
    assert not isplib_amd.iSpLibPlugin.is_patched()
    isplib_amd.iSpLibPlugin.patch_pyg()
    assert fake_ts.matmul is plugin.spmm_autotuned and torch.sparse.mm is not orig_mm
    assert fake_typing.WITH_PT2 is False and fake_typing.WITH_PT20 is False       # isplib/__init__.py:168-169
    isplib_amd.iSpLibPlugin.patch_pyg()                                            # nested
    isplib_amd.iSpLibPlugin.unpatch_pyg()
    assert fake_ts.matmul is plugin.spmm_autotuned                                 # still patched (LIFO)
    isplib_amd.iSpLibPlugin.unpatch_pyg()
    assert fake_ts.matmul is orig_ts and torch.sparse.mm is orig_mm
    assert fake_typing.WITH_PT2 is True and fake_typing.WITH_PT20 is True
    isplib_amd.iSpLibPlugin.unpatch_pyg()                                          # extra unpatch is a no-op (:190)

    @isplib_amd.isplib_autotune
    def inside():
        return fake_ts.matmul is plugin.spmm_autotuned

    assert inside() and fake_ts.matmul is orig_ts


def test_patched_sparse_mm_still_serves_torch_sparse_tensors():
    import isplib_amd
    a = torch.eye(3).to_sparse()
    isplib_amd.iSpLibPlugin.patch_pyg()
    try:
        out = torch.sparse.mm(a, torch.ones(3, 2))
    finally:
        isplib_amd.iSpLibPlugin.unpatch_pyg()
    assert torch.equal(out, torch.ones(3, 2))


def test_synthetic_graphs_have_dataset_shapes():
    from isplib_amd import synth
    rowptr, col, n = synth.dataset_like("cora")
    assert n == 2708 and col.numel() == 10556 and rowptr.numel() == n + 1
    deg = rowptr[1:] - rowptr[:-1]
    row = torch.repeat_interleave(torch.arange(n), deg)
    key = row * n + col
    assert bool((key[1:] > key[:-1]).all()), "rows / in-row columns must be sorted, no duplicates"
    assert torch.equal(torch.sort(col * n + row).values, key), "must be symmetric"
    assert bool((row != col).all())
    # B_alg of config 2
    assert synth.algorithmic_bytes(232965, 232965, 114615892, 128) == 1615810592


# ---- 1-D row partition, world_size 2 over gloo -----------------------------------------------

_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import oracle
from isplib_amd.dist import RowPartition
from tests import cases
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
rowptr, col = cases.random_csr(97, 97, 11.0, seed=5, empty_rows=(0, 50), hub=(3, 400))
val = cases.weights(col.size, 4)
x = cases.dense(97, 24, 3, "integer")
t = torch.from_numpy
part = RowPartition(t(rowptr), t(col), t(val), 97, rank, world)
buf = part.gather_buffer(24)
part.all_gather(part.shard(t(x)), buf)                      # the ONE collective
assert torch.equal(part.unpad(buf), t(x))
for red in ("sum", "mean", "max", "min"):
    ref, ref_arg = oracle.spmm_fw(rowptr, col, val, x, red)
    # local SpMM of this rank's rows straight from the padded gather buffer (oracle stands in for the GPU kernel)
    out, arg = oracle.spmm_fw(part.rowptr.numpy(), part.col_padded.numpy(), part.val.numpy(), buf.numpy(), red)
    r0, r1 = part.row_cuts[rank], part.row_cuts[rank + 1]
    assert np.array_equal(out, ref[r0:r1]), red             # bit-identical to the single-device rows
    if arg is not None:
        garg = np.where(arg == part.nnz, part.total_nnz, arg + part.edge0)
        assert np.array_equal(garg, ref_arg[r0:r1]), red
# the direct per-peer exchange (P-1 send / receive pairs in 1, 2 or P-1 groups) fills the same buffer as the all-gather
want = buf.clone()
shard = part.shard(t(x))
for nb in sorted({{1, 2, max(world - 1, 1)}}):
    got = torch.full_like(buf, float("nan"))
    landed = set()
    for d0, d1, reqs in part.post_direct(shard, got, nb):
        for req in reqs:
            req.wait()
        landed.update((rank - d) % world for d in range(d0, d1))
    assert landed == set(range(world)) - {{rank}}
    got[rank * part.max_rows:(rank + 1) * part.max_rows] = shard
    assert torch.equal(got, want), nb
    dist.barrier()
sizes = [part.row_cuts[i + 1] - part.row_cuts[i] for i in range(world)]
assert sum(sizes) == 97
# the exchange of the max / min backward (RowPartition.minmax_backward: destinations + weighted gradients all-gathered in
# global row order, every rank keeps what lands in its own rows); the local kernel is replaced by a NumPy statement of it
def scatter(dest, gval, lo, n_):
    d, g_ = dest.numpy().astype(np.int64) - lo, gval.numpy()
    out = np.zeros((n_, g_.shape[1]), np.float32)
    rows, cols = np.nonzero((dest.numpy() >= 0) & (d >= 0) & (d < n_))
    np.add.at(out, (d[rows, cols], cols), g_[rows, cols])        # row-major = ascending global row: the kernel's order
    return t(out)
RowPartition.scatter_rows = staticmethod(scatter)
g = cases.dense(97, 24, 7)
r0, r1 = part.row_cuts[rank], part.row_cuts[rank + 1]
for red in ("max", "min"):
    ref, ref_arg = oracle.spmm_fw(rowptr, col, val, x, red)
    _, want = oracle.spmm_minmax_bw(col, val, x, ref_arg, g)
    got = part.minmax_backward(t(ref_arg[r0:r1].copy()), t(g[r0:r1].copy()))
    x0, x1 = part.x_cuts[rank], part.x_cuts[rank + 1]
    assert got.shape == (x1 - x0, 24)
    assert np.allclose(got.numpy(), want[x0:x1], rtol=1e-6, atol=1e-6), red
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world", (2, 3))
def test_row_partition_equivalence_gloo(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29541 + world), WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


_REPL_WORKER = r"""
import importlib.util, os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join({root!r}, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
calls = []
def make():
    calls.append(1)
    g = torch.Generator().manual_seed(1234)                    # only rank 0 may get here
    rowptr = torch.cumsum(torch.cat([torch.zeros(1, dtype=torch.int64), torch.randint(0, 9, (50,), generator=g)]), 0)
    return rowptr, torch.randint(0, 50, (int(rowptr[-1]),), generator=g), 50
rowptr, col, n = bench.replicated_graph(make, torch.device("cpu"), rank, world)
assert (len(calls) == 1) == (rank == 0), "rank 0 generates, the others receive"
g = torch.Generator().manual_seed(1234)
want_rowptr = torch.cumsum(torch.cat([torch.zeros(1, dtype=torch.int64), torch.randint(0, 9, (50,), generator=g)]), 0)
want_col = torch.randint(0, 50, (int(want_rowptr[-1]),), generator=g)
assert n == 50 and torch.equal(rowptr, want_rowptr) and torch.equal(col, want_col)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_bench_graph_is_generated_once_and_replicated_gloo(tmp_path):
    """bench.py at N > 1: rank 0 generates the graph, every other rank receives its copy (sizes first), so the partition
    can never differ between ranks."""
    world = 3
    script = tmp_path / "worker.py"
    script.write_text(_REPL_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


def test_nnz_balanced_cuts():
    from isplib_amd.dist import nnz_balanced_cuts
    rowptr, _ = cases.random_csr(1000, 1000, 20.0, seed=1, hub=(10, 5000))
    cuts = nnz_balanced_cuts(torch.from_numpy(rowptr), 4)
    assert cuts[0] == 0 and cuts[-1] == 1000 and cuts == sorted(cuts)
    per = [int(rowptr[cuts[i + 1]] - rowptr[cuts[i]]) for i in range(4)]
    assert max(per) - min(per) <= 5000 + 60


def test_task_entry_argument_validation_without_gpu():
    """fusedMM_csr_tasks_hip / plan builders: every rejection that is decided before a HIP call."""
    import ctypes
    from isplib_amd import cabi
    L = cabi.lib()
    lane = (ctypes.c_int64 * 9)(*([0] * 9))
    one = ctypes.c_void_p(8)      # never dereferenced: validation fails first

    def call(msg=cabi.MSG_SPMM_SUM, m=4, n=4, k=8, n_tasks=0, slices=8, ws=None, ws_bytes=0, lane_off=lane):
        return L.fusedMM_csr_tasks_hip(msg, m, n, k, 0, None, one, None, one, one, n_tasks, one, one, one, one, slices, lane_off,
                                       one, k, one, k, None, ws, ws_bytes, None)
    assert call(msg=0x11103) == cabi.NO_OPT_IMPL
    assert call(slices=0) == cabi.FAIL and "[1, 4096]" in cabi.last_error()
    assert call(k=3) == cabi.FAIL and "k >= 4" in cabi.last_error()
    assert call(n=1 << 30, k=8) == cabi.FAIL and "3.5 GiB" in cabi.last_error()
    assert call() == cabi.NOT_ENOUGH_MEM
    assert call(m=0) == cabi.SUCCESS
    bad = (ctypes.c_int64 * 9)(0, 0, 0, 0, 0, 0, 0, 0, 5)
    assert call(ws=ctypes.c_void_p(256), ws_bytes=1 << 20, lane_off=bad) == cabi.FAIL and "lane_off" in cabi.last_error()
    info = cabi.TaskPlanInfo()
    assert L.isplib_spmm_tasks_count_hip(4, one, one, one, 0, 1024, 128, one, one, 1 << 20, ctypes.byref(info), None) == cabi.FAIL
    assert L.isplib_spmm_tasks_count_hip(4, one, one, one, 8, 8, 128, one, one, 1 << 20, ctypes.byref(info), None) == cabi.FAIL
    assert L.isplib_spmm_tasks_workspace_bytes(cabi.MSG_SPMM_MAX, 10, 16) >= 2 * 10 * 16 * 4
    assert L.isplib_spmm_slices_build_hip(4, 4, 0, one, one, None, 4097, one, None, None) == cabi.FAIL
    assert L.fusedMM_csr_sliced_phase_hip(cabi.MSG_SPMM_SUM, 4, 4, 8, 0, None, one, one, one, one, 8, 8, 4, 1, one, 8, one, 8,
                                          None, ctypes.c_void_p(256), 1 << 20, None) == cabi.FAIL
    assert "slice range" in cabi.last_error()


# ---- MatrixMarket I/O (the reference tuner's graph format, README.md:147-168) -------------------

def test_mtx_round_trip_and_readme_case(tmp_path):
    import scipy.io
    import scipy.sparse as sp
    from isplib_amd import mtx
    from tests import cases
    rowptr, col, val, *_ = cases.readme_case()
    path = tmp_path / "readme.mtx"
    mtx.write_mtx(path, torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val), (3, 3), "README.md:105-116")
    rp, cl, vl, sizes = mtx.read_mtx(path)
    assert sizes == (3, 3) and rp.tolist() == rowptr.tolist() and cl.tolist() == col.tolist()      # duplicate (0,0) kept
    assert vl.tolist() == val.tolist()
    # pattern files: unit weights
    mtx.write_mtx(path, torch.from_numpy(rowptr), torch.from_numpy(col), None, (3, 3))
    rp, cl, vl, _ = mtx.read_mtx(path)
    assert vl is None and cl.tolist() == col.tolist()
    # a file written by scipy (symmetric storage, comments) expands to both triangles, in (row, col) order
    a = sp.random(40, 40, 0.1, random_state=3, format="coo")
    a = (a + a.T).tocoo()
    scipy.io.mmwrite(str(tmp_path / "sym.mtx"), a, comment="written by scipy", symmetry="symmetric")
    rp, cl, vl, sizes = mtx.read_mtx(tmp_path / "sym.mtx")
    want = a.tocsr()
    want.sort_indices()
    assert sizes == (40, 40) and rp.tolist() == want.indptr.tolist() and cl.tolist() == want.indices.tolist()
    assert np.allclose(vl.numpy(), want.data, rtol=1e-6)


def test_mtx_rejects_what_it_cannot_represent(tmp_path):
    from isplib_amd import mtx
    bad = tmp_path / "bad.mtx"
    bad.write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    with pytest.raises(ValueError, match="coordinate"):
        mtx.read_mtx(bad)
    bad.write_text("%%MatrixMarket matrix coordinate complex general\n2 2 1\n1 1 1 0\n")
    with pytest.raises(ValueError, match="field"):
        mtx.read_mtx(bad)
    bad.write_text("%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1.0\n")
    with pytest.raises(ValueError, match="promises"):
        mtx.read_mtx(bad)
    bad.write_text("%%MatrixMarket matrix coordinate real general\n2 2 1\n3 1 1.0\n")
    with pytest.raises(ValueError, match="outside"):
        mtx.read_mtx(bad)
    empty = tmp_path / "empty.mtx"
    empty.write_text("%%MatrixMarket matrix coordinate pattern general\n% nothing\n3 4 0\n")
    rp, cl, vl, sizes = mtx.read_mtx(empty)
    assert sizes == (3, 4) and rp.tolist() == [0, 0, 0, 0] and cl.numel() == 0 and vl is None


def test_tuning_table_persists_by_graph_signature(tmp_path):
    from isplib_amd import plugin
    from isplib_amd.sparse import SparseStorage
    rowptr = torch.tensor([0, 2, 2, 5], dtype=torch.int64)
    col = torch.tensor([0, 1, 0, 1, 2], dtype=torch.int64)
    st = SparseStorage(rowptr, col, None, (3, 3))
    sig = plugin.graph_signature(st)
    assert sig == "3x3:5:1,2"                  # degrees 2, 0, 3 -> floor(log2(max(d,1))) = 1, 0, 1
    plugin._tuning_db.clear()
    plugin._tuning_db[sig] = {"3:128:0": 16}
    path = tmp_path / "tune.json"
    plugin.iSpLibPlugin.save_tuning(path)
    plugin._tuning_db.clear()
    assert plugin.choose_slices(st, 3, 128) == 0                      # the rule: tiny graph, plain kernel
    assert plugin.iSpLibPlugin.load_tuning(path) == 1
    assert plugin.choose_slices(st, 3, 128) == 16                     # the persisted measurement wins
    same_shape = SparseStorage(rowptr.clone(), col.clone(), None, (3, 3))
    assert plugin.choose_slices(same_shape, 3, 128) == 16             # keyed by content, not by object or pointer
    assert plugin.choose_slices(same_shape, 3, 64) == 0
    plugin._tuning_db.clear()


def test_slice_rule_lives_in_the_c_abi_and_handle_entries_validate():
    import ctypes
    from isplib_amd import cabi, plugin
    L = cabi.lib()
    n, nnz = 232965, 114615892                                     # the Reddit shape: K=32 -> 4, 64 -> 8, >=128 -> 16
    assert [L.isplib_suggest_slices(n, n, nnz, k, 0) for k in (32, 64, 128, 256, 608)] == [4, 8, 8, 8, 8]   # 64-column panels
    assert [L.isplib_suggest_slices(n, n, nnz, k, 1) for k in (32, 64, 128, 256)] == [4, 8, 8, 8]
    assert [L.isplib_suggest_slices(n, n, nnz, k, 0) for k in (41, 100, 602)] == [5, 13, 16]               # ragged rows: one pass / 128-column panels
    assert L.isplib_suggest_slices(2449029, 2449029, 123718280, 256, 0) == 0      # products: mean degree 50 -> plain
    assert L.isplib_suggest_slices(2708, 2708, 10556, 16, 0) == 0                  # Cora: launch-bound, no preparation
    assert plugin.suggest_slices(n, n, nnz, 128, True) == 8 and plugin.suggest_slices(n, n, nnz, 100) == 13                    # the Python name is the same rule
    assert L.isplib_graph_spmm(None, cabi.MSG_SPMM_SUM, 8, None, 8, None, 8, None, None) == cabi.FAIL
    assert "null handle" in cabi.last_error()
    assert L.isplib_graph_set_slices(None, 4) == cabi.FAIL
    out = ctypes.c_void_p()
    assert L.isplib_graph_create(4, 1 << 31, 0, ctypes.c_void_p(8), None, None, ctypes.byref(out)) == cabi.FAIL
    assert L.isplib_graph_create(-1, 4, 0, ctypes.c_void_p(8), None, None, ctypes.byref(out)) == cabi.FAIL
    L.isplib_graph_destroy(None)                                   # a no-op, like free(NULL)


def test_stream_rule_offers_only_what_the_stream_entries_accept():
    """The schedule rule and the entries it feeds share one domain: a shape the stream entry / plan builder would refuse
    (dense operand beyond one 3.5 GiB buffer descriptor, 32-bit edge positions) is not offered the stream schedule, so it
    reaches the task list / plain kernel as it did before the schedule existed instead of raising (round-2 advisor)."""
    import ctypes
    from isplib_amd import cabi
    L = cabi.lib()

    def offered(fn, m, n, nnz, k):
        st, sl, ch = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        return bool(fn(m, n, nnz, k, ctypes.byref(st), ctypes.byref(sl), ctypes.byref(ch)))
    for fn in (L.isplib_suggest_stream, L.isplib_suggest_stream_minmax):
        assert offered(fn, 232965, 232965, 114615892, 64)                         # the Reddit shape: yes
        assert not offered(fn, 1_000_000, 1_000_000, 500_000_000, 1024)           # 4.1 GB of y: one descriptor cannot hold it
        assert offered(fn, 1_000_000, 1_000_000, 500_000_000, 512)                # 2 GB: fine
        assert not offered(fn, 2_000_000, 2_000_000, (1 << 31) + 5, 64)           # edge positions beyond 32 bits
    assert cabi.suggest_stream(1_000_000, 1_000_000, 500_000_000, 1024) is None   # what plugin.choose_stream asks
    # the Reddit shape: 64-column slots, 31 slices; a WEIGHTED plan at whole multiples of 128 columns rides 128-column slots
    # (its weight stream is read once per panel: round 4, 2.90 -> 2.82 ms at K=128, 5.89 -> 5.74 at K=256)
    n, nnz = 232965, 114615892
    assert cabi.suggest_stream(n, n, nnz, 128) == (4, 31, 2057) == cabi.suggest_stream(n, n, nnz, 128, False)
    assert cabi.suggest_stream(n, n, nnz, 128, True) == (2, 63, 2057) and cabi.suggest_stream(n, n, nnz, 256, True)[0] == 2
    assert cabi.suggest_stream(n, n, nnz, 192, True)[0] == 4 and cabi.suggest_stream(n, n, nnz, 64, True) == cabi.suggest_stream(n, n, nnz, 64)
    # max / min (round 4: the row's pair rides in registers, a change of row is an LDS swap): slices of ~3.3 MB of the panel
    assert cabi.stream_minmax_geometry(4)[0] == 32 and cabi.stream_minmax_geometry(8)[0] == 64
    st, sl, ch = cabi.suggest_stream_minmax(n, n, nnz, 64)
    assert (st, sl) == (4, 18) and 2800 <= ch <= 3200
    assert cabi.suggest_stream_minmax(n, n, nnz, 32)[:2] == (8, 9)
    assert not offered(L.isplib_suggest_stream_minmax, 1_100_000, 1_100_000, 500_000_000, 512)   # 2.25 GB: the max / min entry stops at 2 GiB
    assert offered(L.isplib_suggest_stream, 1_100_000, 1_100_000, 500_000_000, 512)


def test_degree_skew_adjustment_of_the_slice_rule():
    from isplib_amd import plugin
    flat = torch.arange(0, 101 * 50, 50, dtype=torch.int64)                       # every row has 50 entries
    assert plugin.degree_cv2(flat) == 0.0 and plugin.skew_adjusted(flat, 8) == 12 and plugin.skew_adjusted(flat, 0) == 0
    deg = torch.tensor([1] * 90 + [500] * 10, dtype=torch.int64)                   # a few hubs: strongly skewed
    skewed = torch.cat([torch.zeros(1, dtype=torch.int64), deg.cumsum(0)])
    assert plugin.degree_cv2(skewed) > 2.0 and plugin.skew_adjusted(skewed, 8) == 8
    assert plugin.skew_adjusted(flat, 60) == 64                                     # capped like the rule itself


def test_bench_self_launches_one_rank_per_gpu_and_refuses_missing_gpus():
    """`python bench.py --gpus N` with no launcher (WORLD_SIZE unset) must start N ranks itself, before touching the
    GPU, and must fail loudly -- never print an n_gpus=1 line -- when fewer than N GPUs are visible."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "3"], 29512)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29512"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    src = open(os.path.join(ROOT, "bench.py")).read()
    launch_at = src.index("raise SystemExit(self_launch(")
    assert launch_at < src.index("torch.cuda.is_available()"), "the launcher must run before anything initialises the GPU"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.device_count() < 8:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode != 0 and "refusing" in r.stderr and '"metric"' not in r.stdout
    # the launcher parent counts devices from sysfs and never through torch.cuda / HIP / amdsmi
    import ast
    used = {}
    for fn in ast.parse(src).body:
        if isinstance(fn, ast.FunctionDef) and fn.name in ("self_launch", "visible_gpu_count", "launcher_command"):
            used[fn.name] = {n.id for n in ast.walk(fn) if isinstance(n, ast.Name)} | \
                            {a_.name.split(".")[0] for n in ast.walk(fn) if isinstance(n, (ast.Import, ast.ImportFrom)) for a_ in n.names}
    assert set(used) == {"self_launch", "visible_gpu_count", "launcher_command"}
    for name, ids in used.items():
        assert not ids & {"torch", "ctypes", "amdsmi", "cabi", "isplib_amd"}, f"{name} must not reach HIP: {ids}"
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        sysfs, dri = os.path.join(tmp, "nodes"), os.path.join(tmp, "dri")
        os.makedirs(dri)
        for i, (simd, minor) in enumerate(((0, -1), (0, -1), (1024, 128), (1024, 129), (1024, 130), (1024, 131))):   # 2 CPU + 4 GPU nodes
            os.makedirs(os.path.join(sysfs, str(i)))
            with open(os.path.join(sysfs, str(i), "properties"), "w") as f:
                f.write(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\ndrm_render_minor {minor}\ngfx_target_version 90500\n")
            if simd and minor != 131:                                   # the fourth GPU's render node is not ours to open
                open(os.path.join(dri, f"renderD{minor}"), "w").close()
        count = lambda **env: bench.visible_gpu_count(sysfs, dri, env)  # noqa: E731
        assert count() == 3
        assert count(HIP_VISIBLE_DEVICES="0,2") == 2 and count(ROCR_VISIBLE_DEVICES="1") == 1
        assert count(ROCR_VISIBLE_DEVICES="0,1", HIP_VISIBLE_DEVICES="0,1,2") == 2      # the second list indexes what the first left
        assert count(HIP_VISIBLE_DEVICES="0,7,1") == 1 and count(HIP_VISIBLE_DEVICES="") == 0 and count(CUDA_VISIBLE_DEVICES="-1") == 0
        assert count(ROCR_VISIBLE_DEVICES="GPU-0123456789abcdef") == 1
        assert bench.visible_gpu_count(os.path.join(tmp, "missing"), dri, {}) in (0, None)
    # a launcher that disagrees with --gpus is an error too, not a silent single-rank run
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, env=env2, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def _slice_table_np(rowptr, col, n, slices):
    w = -(-n // slices)
    m = rowptr.size - 1
    table = np.zeros((m, slices + 1), np.int64)
    for i in range(m):
        c = col[rowptr[i]:rowptr[i + 1]]
        table[i] = rowptr[i] + np.searchsorted(c, np.arange(slices + 1) * w, side="left")
        table[i, slices] = rowptr[i + 1]
    return table


@pytest.mark.parametrize("geom", ((8, 8, 8, 512, 4), (5, 3, 16, 64, 16), (16, 2, 32, 100000, 1), (1, 4, 8, 40, 1000)))
def test_sweep_plan_covers_every_edge_once_in_csr_order(geom):
    """Host logic of the sweep schedule's plan (isplib_amd/plan.py: sweep_plan_arrays), replayed on the CPU the way
    the kernel walks it: every stored entry belongs to exactly one task of the wave that owns its (virtual) row, a
    row's tasks are met in ascending CSR order, a wave's tasks are in (slice, slot) order, hub rows are cut into
    chunks whose partial rows are contiguous and in CSR order, and the replay reproduces A @ X."""
    from isplib_amd.plan import sweep_plan_arrays
    slices, wpg, rpw, chunk, min_seg = geom
    m, n = 300, 257
    rowptr, col = cases.random_csr(m, n, 30, 1, empty_rows=(0, 5, 299), hub=(7, 5000))
    table = _slice_table_np(rowptr, col, n, slices)
    p = sweep_plan_arrays(torch.from_numpy(rowptr), torch.from_numpy(table), slices, wpg, rpw, chunk, min_seg)
    nw = p["gens"] * p["waves_per_gen"]
    wr = p["wave_row"].numpy().reshape(nw, rpw)
    wp = p["wave_part"].numpy().reshape(nw, rpw)
    off, tb, tm = p["wave_task_off"].numpy(), p["task_b"].numpy(), p["task_meta"].numpy()
    x = np.random.default_rng(0).random((n, 3))
    out, parts = np.zeros((m, 3)), np.zeros((p["n_parts"], 3))
    cover = np.zeros(col.size, int)
    width = -(-n // slices)
    rows_seen = np.zeros(m, int)
    for w in range(nw):
        acc = np.zeros((rpw, 3))
        last_pos = np.full(rpw, -1, np.int64)
        last_key = -1
        slot_edges = np.zeros(rpw, np.int64)
        for t in range(off[w], off[w + 1]):
            slot_edges[int(tm[t]) >> 24] += int(tm[t]) & 0xFFFFFF
        for t in range(off[w], off[w + 1]):
            slot, ln, b = int(tm[t]) >> 24, int(tm[t]) & 0xFFFFFF, int(tb[t])
            r = wr[w, slot]
            assert r >= 0 and ln > 0 and rowptr[r] <= b and b + ln <= rowptr[r + 1]
            assert b > last_pos[slot], "a row's tasks must come in ascending CSR order"
            last_pos[slot] = b
            if min_seg == 1 and slot_edges[slot] >= slices:    # no slice groups: a task's phase is the slice of its first edge
                key = int(col[b]) // width * rpw + slot
                assert key > last_key, "a wave's tasks must be in (slice, slot) order"
                last_key = key
            cover[b:b + ln] += 1
            acc[slot] += x[col[b:b + ln]].sum(0)
        for j in range(rpw):
            if wr[w, j] < 0:
                continue
            if wp[w, j] >= 0:
                parts[wp[w, j]] = acc[j]
            else:
                out[wr[w, j]] = acc[j]
                rows_seen[wr[w, j]] += 1
    hr, ho = p["hub_row"].numpy(), p["hub_off"].numpy()
    for h in range(hr.size):
        out[hr[h]] = parts[ho[h]:ho[h + 1]].sum(0)
        rows_seen[hr[h]] += 1
    assert (cover == 1).all() and (rows_seen == 1).all()
    ref = np.stack([x[col[rowptr[i]:rowptr[i + 1]]].sum(0) for i in range(m)])
    assert np.allclose(out, ref, rtol=1e-12, atol=1e-12)
    assert int((p["task_meta"] & 0xFFFFFF).sum()) == col.size


@pytest.mark.parametrize("streams,rpw", ((4, 32), (8, 64)))
def test_stream_order_keeps_the_lowest_csr_position_on_ties(streams, rpw):
    """The argument behind spmm_stream_minmax_kernel, replayed on the CPU: for rows whose columns ascend, a stream lists a
    row's edges in CSR order (slice by slice, CSR order inside a slice), so "only a strictly better candidate replaces
    the one held" -- tracked by WORD INDEX, translated through perm at the end -- leaves the lowest CSR position among
    equal candidates; the pieces of a hub row are merged by explicit position compare.  Integer operands: ties everywhere.
    Against the oracle, values and arg.  And: the plan builder refuses a graph with an unsorted row."""
    import oracle
    from isplib_amd.plan import build_stream_plan, stream_plan_arrays
    m, n, slices, wpg, chunk = 120, 90, 5, 3, 40
    rowptr, col = cases.random_csr(m, n, 25, 3, empty_rows=(0, 50), hub=(7, 700), duplicates=True)
    val = cases.weights(col.size, 4, "signed_int")
    x = cases.dense(n, 5, 3, "integer")
    p = stream_plan_arrays(torch.from_numpy(rowptr), torch.from_numpy(col), n, slices, wpg, rpw, streams, chunk)
    nw = p["gens"] * wpg
    words = p["words"].numpy().astype(np.int64) & 0xFFFFFFFF
    perm, off = p["perm"].numpy(), p["wave_step_off"].numpy()
    wr, wp = p["wave_row"].numpy().reshape(nw, rpw), p["wave_part"].numpy().reshape(nw, rpw)
    k = x.shape[1]
    lowest = np.finfo(np.float32).min
    out, arg = np.zeros((m, k), np.float32), np.full((m, k), col.size, np.int64)
    part_v, part_i = np.full((p["n_parts"], k), lowest, np.float32), np.full((p["n_parts"], k), np.iinfo(np.int64).max, np.int64)
    for w in range(nw):
        best = np.full((rpw, k), lowest, np.float32)
        widx = np.full((rpw, k), -1, np.int64)
        for st in range(off[w], off[w + 1]):
            for g in range(streams):
                i = st * streams + g
                c, lr = words[i] & 0xFFFFFF, words[i] >> 24
                if c == n:
                    continue                                     # padding: the kernel steers it to a spare row
                t = np.float32(val[perm[i]]) * x[c]
                win = t > best[lr]                               # strict, in stream order
                best[lr][win], widx[lr][win] = t[win], i
        for j in range(rpw):
            if wr[w, j] < 0:
                continue
            pos = np.where(widx[j] >= 0, perm[np.maximum(widx[j], 0)], np.iinfo(np.int64).max)
            if wp[w, j] >= 0:
                part_v[wp[w, j]], part_i[wp[w, j]] = best[j], pos
            elif rowptr[wr[w, j] + 1] > rowptr[wr[w, j]]:
                out[wr[w, j]], arg[wr[w, j]] = best[j], np.where(widx[j] >= 0, pos, col.size)
    hr, ho = p["hub_row"].numpy(), p["hub_off"].numpy()
    for h in range(hr.size):
        v, a = np.full(k, lowest, np.float32), np.full(k, np.iinfo(np.int64).max, np.int64)
        for q in range(ho[h], ho[h + 1]):
            take = (part_v[q] > v) | ((part_v[q] == v) & (part_i[q] < a))
            v[take], a[take] = part_v[q][take], part_i[q][take]
        out[hr[h]], arg[hr[h]] = v, a
    ref, ref_arg = oracle.spmm_fw(rowptr, col, val, x, "max")
    assert np.array_equal(out, ref) and np.array_equal(arg, ref_arg)
    # unsorted rows: no plan (the stream order would no longer be the CSR order)
    r2, c2 = cases.random_csr(40, 30, 6, 9, sort_cols=False)
    assert build_stream_plan(torch.from_numpy(r2), torch.from_numpy(c2), None, 30, 2, 3, rpw, streams, 16, minmax=True) is None


@pytest.mark.parametrize("geom", ((8, 8, 16, 4, 512), (5, 3, 32, 8, 64), (3, 2, 16, 2, 100000)))
def test_stream_plan_lists_every_edge_once_slice_by_slice(geom):
    """Host logic of the stream plan (isplib_amd/plan.py: stream_plan_arrays), replayed on the CPU the way
    spmm_stream_kernel walks it: every stored entry appears in exactly one stream, in the stream of the slot that owns
    its (virtual) row; a stream visits the column slices in ascending order and inside a slice its rows one after the
    other, each in ascending CSR order; hub rows are dealt round robin to virtual rows, whose partial rows are
    contiguous; padding words name column n and a row of the stream's own slot; the replay reproduces A @ X."""
    from isplib_amd.plan import stream_plan_arrays
    slices, wpg, rpw, streams, chunk = geom
    m, n = 300, 257
    rowptr, col = cases.random_csr(m, n, 30, 1, empty_rows=(0, 5, 299), hub=(7, 5000))
    p = stream_plan_arrays(torch.from_numpy(rowptr), torch.from_numpy(col), n, slices, wpg, rpw, streams, chunk)
    nw = p["gens"] * wpg
    words = p["words"].numpy().astype(np.int64) & 0xFFFFFFFF
    perm = p["perm"].numpy()
    off = p["wave_step_off"].numpy()
    wr, wp = p["wave_row"].numpy().reshape(nw, rpw), p["wave_part"].numpy().reshape(nw, rpw)
    per, width = rpw // streams, -(-n // slices)
    x = np.random.default_rng(0).random((n, 3))
    out, parts, seen = np.zeros((m, 3)), np.zeros((p["n_parts"], 3)), np.zeros(col.size, int)
    for w in range(nw):
        acc = np.zeros((rpw, 3))
        last = [(-1, -1, -1)] * streams                         # (slice, local row, CSR position) of the stream's previous word
        for st in range(off[w], off[w + 1]):
            for g in range(streams):
                wd = words[st * streams + g]
                c, lr = wd & 0xFFFFFF, wd >> 24
                assert lr // per == g, "a word names a row of its own slot"
                if c == n:
                    assert perm[st * streams + g] == -1
                    continue
                e = perm[st * streams + g]
                r = wr[w, lr]
                assert col[e] == c and rowptr[r] <= e < rowptr[r + 1]
                key = (c // width, lr, e)
                assert key > last[g], "streams go slice by slice, row by row, in CSR order"
                last[g] = key
                seen[e] += 1
                acc[lr] += x[c]
        for j in range(rpw):
            if wr[w, j] < 0:
                continue
            if wp[w, j] >= 0:
                parts[wp[w, j]] = acc[j]
            else:
                out[wr[w, j]] = acc[j]
    hr, ho = p["hub_row"].numpy(), p["hub_off"].numpy()
    for h in range(hr.size):
        out[hr[h]] = parts[ho[h]:ho[h + 1]].sum(0)
    assert (seen == 1).all()
    ref = np.stack([x[col[rowptr[i]:rowptr[i + 1]]].sum(0) for i in range(m)])
    assert np.allclose(out, ref, rtol=1e-12, atol=1e-12)
    if chunk < 5000:                                            # the hub row's pieces all span the whole column range
        hub_slots = [(w, j) for w in range(nw) for j in range(rpw) if wr[w, j] == 7]
        assert len(hub_slots) == -(-5000 // chunk)


@pytest.mark.parametrize("geom", ((4, 8, 8, 5, 16, 3, 100), (8, 16, 8, 12, 8, 2, 64), (4, 64, 16, 3, 32, 64, 50), (8, 128, 8, 40, 256, 64, 1000)))
def test_hybrid_plan_serves_every_edge_once_from_the_table_or_the_gather_stream(geom):
    """Host logic of the hybrid schedule's plan (isplib_amd/plan.py: hybrid_plan_arrays), replayed on the CPU the way the
    kernel walks it: every stored entry is either a cold word of its wave's stream (column = the entry's column) or a hot
    word of its wave's chunk of the entry's slice (the table row holds the entry's column); local rows stay inside their
    slot; chunks respect the cap; padding points at the zero row / column n; and the replay reproduces A x."""
    from isplib_amd.plan import hybrid_plan_arrays
    streams, rpw, wpg, slices, table_rows, cap, chunk = geom
    n = 700
    rowptr, col = cases.random_csr(n, n, 40.0, seed=3, empty_rows=(0, 350), hub=(9, 650), duplicates=True)
    a = hybrid_plan_arrays(torch.from_numpy(rowptr), torch.from_numpy(col), n, slices, wpg, rpw, streams, chunk, table_rows, cap)
    c = a["cold"]
    rng = np.random.default_rng(0)
    x = rng.integers(-3, 4, (n, 3)).astype(np.float64)
    xz = np.vstack([x, np.zeros((1, 3))])
    nw, per, width = c["gens"] * c["waves_per_gen"], rpw // streams, -(-n // slices)
    acc = np.zeros((nw, rpw, 3))
    seen = np.zeros(col.size, np.int64)
    words, perm, wso = c["words"].numpy().astype(np.int64) & 0xFFFFFFFF, c["perm"].numpy(), c["wave_step_off"].numpy()
    hw, hp, hso = a["hot_words"].numpy().astype(np.int64) & 0xFFFFFFFF, a["hot_perm"].numpy(), a["hot_step_off"].numpy()
    table = a["hot_rows"].numpy().reshape(slices, table_rows)
    assert np.all(table[:, table_rows - 1] == n), "the last table row is the zero row in every slice"
    for w in range(nw):
        for s in range(wso[w], wso[w + 1]):
            for g in range(streams):
                wd = words[s * streams + g]
                lr, cc, e = wd >> 24, wd & 0xFFFFFF, perm[s * streams + g]
                assert g * per <= lr < (g + 1) * per
                assert (col[e] == cc) if e >= 0 else (cc == n)
                acc[w, lr] += xz[cc]
                if e >= 0:
                    seen[e] += 1
        for sl in range(slices):
            o0, o1 = hso[w * slices + sl], hso[w * slices + sl + 1]
            assert 0 <= o1 - o0 <= cap
            for s in range(o0, o1):
                for g in range(streams):
                    wd = hw[s * streams + g]
                    lr, ti, e = wd >> 24, wd & 0xFFFF, hp[s * streams + g]
                    assert g * per <= lr < (g + 1) * per and ti < table_rows
                    cc = table[sl, ti]
                    assert (col[e] == cc and cc // width == sl) if e >= 0 else (ti == table_rows - 1)
                    acc[w, lr] += xz[cc]
                    if e >= 0:
                        seen[e] += 1
    assert np.all(seen == 1) and a["hot_edges"] == int((hp >= 0).sum()) and a["hot_edges"] > 0
    out = np.zeros((n, 3))
    wr = c["wave_row"].numpy().reshape(nw, rpw)
    for w in range(nw):
        for l in range(rpw):
            if wr[w, l] >= 0:
                out[wr[w, l]] += acc[w, l]
    ref = np.zeros((n, 3))
    np.add.at(ref, np.repeat(np.arange(n), np.diff(rowptr)), x[col])
    assert np.array_equal(out, ref)


# ---- bench.py at N > 1: failure containment (class Guard, explore_candidates), on the CPU over gloo ------------------------
_GUARD_WORKER = r"""
import importlib.util, json, os, sys, time
from datetime import timedelta
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join({root!r}, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode = os.environ["GUARD_MODE"]
guard = bench.Guard(rank, world)
guard.arm("north_star form", float(os.environ.get("T_SAFE", "20")))
dist.init_process_group("gloo", timeout=timedelta(seconds=30))
t = torch.ones(4)
parked = []                                    # what isplib_amd.dist.RowPartition._kernel / _raise_parked do for a real schedule

agree = bench.StoreAgreement(dist.distributed_c10d._get_default_store(), rank, world, timeout_s=float(os.environ.get("T_CAND", "6")) + 4)

def exchange():                                # a schedule: its collectives always complete, a local failure is raised after them
    dist.all_reduce(t)
    dist.all_reduce(t)
    if parked:
        e = parked.pop()
        e.collectives_complete = True
        raise e

def clock(fn):
    fn()
    return 1.0

if mode == "hang-before-result" and rank == 1:
    time.sleep(1e6)
dist.all_reduce(t)                             # the north_star form, measured: from here on there is a result
guard.offer({{"metric": "edges_aggregated_per_sec", "value": 1.0, "n_gpus": world}} if rank == 0 else None)
if mode == "sigterm":
    guard.arm("holding the result", 120)
    open(os.path.join(os.environ["HOLD_DIR"], f"holding.{{rank}}"), "w").close()
    time.sleep(1e6)
cands = {{"A": exchange, "B": exchange, "C": exchange}}
broken = None
times = {{}}
try:
    times = bench.explore_candidates(cands, check=lambda name, fn: (fn(), True)[1], clock=clock, agree=agree, guard=guard, rank=rank,
                                     per_candidate_s=float(os.environ.get("T_CAND", "6")), until_s=120.0, on_fault=parked.append)
except Exception as e:
    broken = repr(e)
res = {{"metric": "edges_aggregated_per_sec", "value": 1.0, "n_gpus": world, "candidates_ms": times}}
if broken:
    res["abandoned"] = broken
guard.emit(res)
sys.stdout.flush()
os._exit(0)
"""


def _run_guard_workers(tmp_path, mode, port, inject="", world=2, timeout=90, extra_env=None):
    script = tmp_path / "guard_worker.py"
    script.write_text(_GUARD_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), OMP_NUM_THREADS="1",
               GUARD_MODE=mode, ISPLIB_BENCH_INJECT=inject, **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    return procs


def _json_lines(text):
    import json
    return [json.loads(ln) for ln in text.splitlines() if ln.startswith("{") and '"metric"' in ln]


def test_bench_candidate_that_fails_on_one_rank_is_dropped_on_every_rank(tmp_path):
    """A local kernel failure inside an optional exchange schedule on ONE rank (parked until the schedule's collectives are
    done, as isplib_amd.dist does): the schedule is dropped on every rank and the job goes on with the remaining ones."""
    import time as _t
    t0 = _t.time()
    procs = _run_guard_workers(tmp_path, "explore", 29571, inject="kernel:B:1")
    outs = [p.communicate(timeout=90) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    lines = _json_lines(outs[0][0])
    assert len(lines) == 1 and sorted(lines[0]["candidates_ms"]) == ["A", "C"] and "abandoned" not in lines[0]
    assert _json_lines(outs[1][0]) == [] and "dropped on every rank" in outs[0][1]
    assert _t.time() - t0 < 60


def test_bench_rank_that_never_arrives_ends_the_job_nonzero_within_the_deadline(tmp_path):
    """Before any result exists a rank that blocks for ever must not hang the job: every rank's guard ends it with a
    non-zero exit code and no JSON line inside the phase's deadline."""
    import time as _t
    t0 = _t.time()
    procs = _run_guard_workers(tmp_path, "hang-before-result", 29572, extra_env={"T_SAFE": "6"})
    outs = [p.communicate(timeout=90) for p in procs]
    # (exit code 5 from a rank's own guard; a rank whose peer's guard fired first may instead see its collective fail)
    assert all(p.returncode != 0 for p in procs) and 5 in [p.returncode for p in procs], [o[1][-2000:] for o in outs]
    assert _json_lines(outs[0][0]) == [] and _json_lines(outs[1][0]) == []
    both = outs[0][1] + outs[1][1]
    assert "deadline passed in phase 'north_star form'" in both and "no result yet" in both and _t.time() - t0 < 60


@pytest.mark.parametrize("kind", ("hang", "raise"))
def test_bench_optional_schedule_that_hangs_or_desynchronises_cannot_lose_the_measured_result(tmp_path, kind):
    """After the north_star form has been measured, an optional schedule in which one rank never comes back (hang) or
    raises before its first collective while its peers wait in it (raise) ends with exit code 0 and rank 0's ONE JSON
    line: the result already measured, marked with what was abandoned."""
    import time as _t
    t0 = _t.time()
    procs = _run_guard_workers(tmp_path, "explore", 29573 + (kind == "raise"), inject=f"{kind}:B:1", extra_env={"T_CAND": "5"})
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), [(p.returncode, o[1][-2000:]) for p, o in zip(procs, outs)]
    lines = _json_lines(outs[0][0])
    assert len(lines) == 1 and lines[0]["value"] == 1.0 and lines[0]["n_gpus"] == 2
    if kind == "hang":
        assert "optional schedule 'B'" in lines[0]["abandoned"]
    assert _json_lines(outs[1][0]) == [] and _t.time() - t0 < 90


def test_bench_sigterm_from_the_launcher_prints_the_measured_result(tmp_path):
    """torchrun ends the surviving ranks with SIGTERM when one rank dies: rank 0's guard takes the signal on its own
    thread (the main thread may sit in a collective) and prints the result it holds before leaving."""
    import signal as _sig
    import time as _t
    procs = _run_guard_workers(tmp_path, "sigterm", 29575, extra_env={"HOLD_DIR": str(tmp_path)})
    deadline = _t.time() + 90
    while not all((tmp_path / f"holding.{r}").exists() for r in range(2)) and _t.time() < deadline:      # both ranks hold the result
        _t.sleep(0.2)
    assert all((tmp_path / f"holding.{r}").exists() for r in range(2))
    for p in procs:
        p.send_signal(_sig.SIGTERM)
    outs = [p.communicate(timeout=30) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], [o[1][-1500:] for o in outs]
    lines = _json_lines(outs[0][0])
    assert len(lines) == 1 and "SIGTERM" in lines[0]["abandoned"] and _json_lines(outs[1][0]) == []


def test_bench_launcher_has_a_time_limit_and_ends_the_process_group(tmp_path, monkeypatch):
    """`python bench.py --gpus N` as its own launcher: ranks that never finish are ended (whole process group) when the
    limit passes, the launcher returns 124 and prints no line; the second attempt (north_star form only) is a fresh set
    of children, never a re-used process."""
    import importlib.util
    import time as _t
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    marker = tmp_path / "started"
    sleeper = tmp_path / "sleeper.py"
    sleeper.write_text("import os, sys, time\nopen(sys.argv[1], 'a').write(os.environ.get('ISPLIB_OVERLAP', '1') + '\\n')\ntime.sleep(1e6)\n")
    monkeypatch.setattr(bench, "launcher_command", lambda gpus, argv, port: [sys.executable, str(sleeper), str(marker)])
    monkeypatch.setattr(bench, "visible_gpu_count", lambda *a_, **k_: 8)
    monkeypatch.setenv("ISPLIB_BENCH_LAUNCH_TIMEOUT", "3")
    t0 = _t.time()
    rc = bench.self_launch(types.SimpleNamespace(gpus=2), ["--gpus", "2"])
    assert rc == 124 and _t.time() - t0 < 40
    assert marker.read_text().split() == ["1"]            # no time left for a second attempt: none was started
    monkeypatch.setenv("ISPLIB_BENCH_LAUNCH_TIMEOUT", "200")
    quick = tmp_path / "quick.py"
    quick.write_text("import os, sys\nopen(sys.argv[1], 'a').write(os.environ.get('ISPLIB_OVERLAP', '1') + '\\n')\nsys.exit(7)\n")
    marker.write_text("")
    monkeypatch.setattr(bench, "launcher_command", lambda gpus, argv, port: [sys.executable, str(quick), str(marker)])
    assert bench.self_launch(types.SimpleNamespace(gpus=2), ["--gpus", "2"]) == 7
    assert marker.read_text().split() == ["1", "0"]       # first the full run, then once more restricted to the north_star form


def test_bench_launcher_stamps_a_retry_into_the_line_it_relays(tmp_path, monkeypatch, capsys):
    """A first attempt that fails without a JSON line and a clean retry must not look like a clean run: the relayed line
    carries `launch` = attempts, the first attempt's exit code / time-limit flag, and that the retry was restricted to the
    north_star form.  A clean first attempt says attempts = 1 and has no `first_attempt`."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    child = tmp_path / "child.py"
    # first attempt (ISPLIB_OVERLAP unset): dies with exit 9 and no line; the retry (ISPLIB_OVERLAP=0) prints a result
    child.write_text("import json, os, sys\n"
                     "if os.environ.get('ISPLIB_OVERLAP', '1') != '0' and os.environ.get('CLEAN') != '1':\n"
                     "    print('rank 1 aborted'); sys.exit(9)\n"
                     "print(json.dumps({'metric': 'edges_aggregated_per_sec', 'value': 1.0, 'n_gpus': 2}))\n")
    monkeypatch.setattr(bench, "launcher_command", lambda gpus, argv, port: [sys.executable, str(child)])
    monkeypatch.setattr(bench, "visible_gpu_count", lambda *a_, **k_: 8)
    monkeypatch.setenv("ISPLIB_BENCH_LAUNCH_TIMEOUT", "200")
    assert bench.self_launch(types.SimpleNamespace(gpus=2), ["--gpus", "2"]) == 0
    lines = _json_lines(capsys.readouterr().out)
    assert len(lines) == 1
    launch = lines[0]["launch"]
    assert launch["attempts"] == 2 and launch["restricted_to_north_star_form"] is True
    assert launch["first_attempt"]["rc"] == 9 and launch["first_attempt"]["timed_out"] is False
    assert launch["this_attempt"] == {"rc": 0, "timed_out": False}
    monkeypatch.setenv("CLEAN", "1")
    assert bench.self_launch(types.SimpleNamespace(gpus=2), ["--gpus", "2"]) == 0
    lines = _json_lines(capsys.readouterr().out)
    assert lines[0]["launch"] == {"attempts": 1, "restricted_to_north_star_form": False, "this_attempt": {"rc": 0, "timed_out": False}}


def test_exchange_schedules_park_local_kernel_errors_until_their_collectives_are_done():
    """isplib_amd.dist.RowPartition._kernel / _raise_parked: the first local kernel failure of an exchange is parked (later
    kernels of the same exchange are skipped, its communication is not), raised once the exchange is complete and marked
    `collectives_complete` -- what lets bench.py drop a schedule on every rank instead of leaving peers in a collective."""
    from isplib_amd.dist import RowPartition
    rowptr = torch.tensor([0, 2, 3, 5], dtype=torch.int64)
    col = torch.tensor([0, 2, 1, 0, 1], dtype=torch.int64)
    part = RowPartition(rowptr, col, None, 3, 0, 1)
    ran = []
    part._kernel(ran.append, "a")
    part.fail_next_kernel = RuntimeError("boom")
    part._kernel(ran.append, "b")                    # fails (injected): parked, not raised
    part._kernel(ran.append, "c")                    # skipped: the exchange's result is void anyway
    assert ran == ["a"] and part.fail_next_kernel is None
    with pytest.raises(RuntimeError, match="boom") as info:
        part._raise_parked()
    assert info.value.collectives_complete is True
    part._raise_parked()                             # nothing parked any more
    part._kernel(ran.append, "d")
    assert ran == ["a", "d"]
    # round-4 advisor: an error parked by an exchange whose waits raised before _raise_parked must not be what the NEXT exchange
    # reports (and must not make it skip its kernels): every schedule starts with a clean slate
    part._parked = RuntimeError("stale error of an earlier exchange")
    x = torch.zeros((part.max_rows, 4))
    out = torch.zeros((part.rows, 4))
    with pytest.raises(Exception) as info:           # (no GPU here: the schedule's own first kernel call fails -- THAT is what surfaces)
        part.spmm_pipelined(x, out, ([(0, 4)], [torch.zeros((part.max_rows, 4))], [torch.zeros((part.ncols_padded, 4))], object(), None), "sum")
    assert "stale" not in str(info.value) and part._parked is None
