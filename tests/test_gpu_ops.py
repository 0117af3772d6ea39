"""GPU tests of everything around the forward kernel: the torch operator + autograd
layer against the golden vectors produced by the reference's own autograd code
(tests/golden/spmm_ref_layer.npz), the fused max/min backward, SDDMM dA, the
device-side CSR->CSC preparation, and the plug-in surface.
"""
import os

import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "spmm_ref_layer.npz")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _close(got, ref, rtol=1e-5, atol=1e-5):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    assert np.allclose(got, ref, rtol=rtol, atol=atol), f"max abs err {np.max(np.abs(got - ref))}"


@pytest.fixture(scope="module")
def golden():
    z = np.load(GOLDEN)
    return z, sorted({k.split("/")[0] for k in z.files})


def test_reference_schema_ops_match_reference_layer(gpu, golden, oracle_mod):
    """torch.ops.isplib.fusedmm_spmm{,_mean,_max,_min} called exactly as the reference wrapper calls
    them (isplib/__init__.py:140-151), forward AND backward, against the reference layer's outputs."""
    z, names = golden
    ops = torch.ops.isplib
    for n in names:
        rowptr, col, val, x, g = (z[f"{n}/{k}"] for k in ("rowptr", "col", "val", "x", "g"))
        ncols = int(z[f"{n}/ncols"])
        row, rowcount, colptr, csr2csc = oracle_mod.csr_transpose(rowptr, col, ncols)
        _, new_row, new_rowcount = oracle_mod.mean_bw_weights(rowptr, col, val, ncols)
        d = lambda a: _t(a, gpu)  # noqa: E731
        tol = cases.sum_tolerance(oracle_mod, rowptr, col, val, x)

        xs = d(x).requires_grad_(True)
        out = ops.fusedmm_spmm(d(row), d(rowptr), d(col), d(val), d(colptr), d(csr2csc), xs, d(val[csr2csc]), d(row[csr2csc]))
        out.backward(d(g))
        assert np.all(np.abs(out.detach().cpu().numpy() - z[f"{n}/sum/out"]) <= tol), n
        _close(xs.grad, z[f"{n}/sum/dx"])

        xs = d(x).requires_grad_(True)
        out = ops.fusedmm_spmm_mean(d(row), d(rowptr), d(col), d(val), d(rowcount), d(colptr), d(csr2csc), xs,
                                    d(new_row), d(new_rowcount))
        out.backward(d(g))
        assert np.all(np.abs(out.detach().cpu().numpy() - z[f"{n}/mean/out"]) <= tol), n
        _close(xs.grad, z[f"{n}/mean/dx"])

        for red, fn in (("max", ops.fusedmm_spmm_max), ("min", ops.fusedmm_spmm_min)):
            xs, vs = d(x).requires_grad_(True), d(val).requires_grad_(True)
            out, arg = fn(d(rowptr), d(col), vs, xs)
            out.backward(d(g))
            assert np.array_equal(out.detach().cpu().numpy().view(np.uint32), z[f"{n}/{red}/out"].view(np.uint32)), (n, red)
            assert np.array_equal(arg.cpu().numpy(), z[f"{n}/{red}/arg"]), (n, red)
            assert not arg.requires_grad
            _close(xs.grad, z[f"{n}/{red}/dx"])          # float atomics: order differs from ATen's CPU scatter
            _close(vs.grad, z[f"{n}/{red}/dval"])


def test_ops_build_missing_transpose_operands_on_device(gpu, golden):
    """Unlike the reference (csrc/fusedmm.cpp:246-247,333) the cached operands may be None."""
    z, names = golden
    ops = torch.ops.isplib
    for n in names:
        rowptr, col, val, x, g = (_t(z[f"{n}/{k}"], gpu) for k in ("rowptr", "col", "val", "x", "g"))
        xs = x.clone().requires_grad_(True)
        ops.fusedmm_spmm(None, rowptr, col, val, None, None, xs, None, None).backward(g)
        _close(xs.grad, z[f"{n}/sum/dx"])
        xs = x.clone().requires_grad_(True)
        ops.fusedmm_spmm_mean(None, rowptr, col, val, None, None, None, xs, None, None).backward(g)
        _close(xs.grad, z[f"{n}/mean/dx"])


def test_unit_weight_ops_and_value_gradients(gpu, oracle_mod):
    rowptr, col = cases.random_csr(70, 50, 6.0, seed=8, empty_rows=(2,))
    x, g = cases.dense(50, 24, 3), cases.dense(70, 24, 5)
    ones = np.ones(col.size, np.float32)
    d = lambda a: _t(a, gpu)  # noqa: E731
    ops = torch.ops.isplib
    # value=None: unit weights, no value gradient
    xs = d(x).requires_grad_(True)
    out = ops.fusedmm_spmm(None, d(rowptr), d(col), None, None, None, xs, None, None)
    out.backward(d(g))
    ref, _ = oracle_mod.spmm_fw(rowptr, col, ones, x, "sum")
    _close(out, ref)
    _close(xs.grad, oracle_mod.spmm_sum_bw(rowptr, col, ones, 50, g))
    # dA of sum / mean (SDDMM) -- the reference leaves it undefined (csrc/fusedmm.cpp:268-272)
    val = cases.weights(col.size, 4)
    for name, mean in (("fusedmm_spmm", False), ("fusedmm_spmm_mean", True)):
        vs = d(val).requires_grad_(True)
        xs = d(x).requires_grad_(True)
        if mean:
            out = ops.fusedmm_spmm_mean(None, d(rowptr), d(col), vs, None, None, None, xs, None, None)
        else:
            out = ops.fusedmm_spmm(None, d(rowptr), d(col), vs, None, None, xs, None, None)
        out.backward(d(g))
        _close(vs.grad, oracle_mod.sddmm(rowptr, col, x, g, mean=mean), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("k", (1, 16, 41, 128, 260))
def test_sddmm_widths_and_hub(gpu, oracle_mod, k):
    from isplib_amd import cabi
    rowptr, col = cases.random_csr(80, 700, 5.0, seed=k, empty_rows=(0,), hub=(9, 6000))
    x, g = cases.dense(700, k, 3), cases.dense(80, k, 5)
    for mean in (False, True):
        got = cabi.sddmm(_t(rowptr, gpu), _t(col, gpu), _t(x, gpu), _t(g, gpu), mean)
        ref = oracle_mod.sddmm(rowptr, col, x, g, mean=mean)
        scale = np.abs(ref).max() + 1e-6
        assert np.max(np.abs(got.cpu().numpy() - ref)) <= 2e-6 * scale * max(1, k / 16)


@pytest.mark.parametrize("k", (4, 41, 128, 300))
def test_sddmm_over_task_plan(gpu, oracle_mod, k):
    from isplib_amd import cabi
    from isplib_amd.plan import build_task_plan
    rowptr, col = cases.random_csr(200, 700, 25.0, seed=k, empty_rows=(0, 199), hub=(9, 6000))
    x, g = cases.dense(700, k, 3), cases.dense(200, k, 5)
    d = [_t(a, gpu) for a in (rowptr, col, x, g)]
    for slices, chunk, short in ((8, 1024, 128), (16, 64, 0)):
        plan = build_task_plan(d[0], d[1], 700, slices, chunk, short)
        for mean in (False, True):
            got = cabi.sddmm_tasks(d[0], d[1], plan, d[2], d[3], mean)
            ref = oracle_mod.sddmm(rowptr, col, x, g, mean=mean)
            scale = np.abs(ref).max() + 1e-6
            assert np.max(np.abs(got.cpu().numpy() - ref)) <= 2e-6 * scale * max(1, k / 16)


@pytest.mark.gpu
@pytest.mark.parametrize("k", (128, 136, 129, 256, 67))
def test_sddmm_over_task_plan_in_column_panels(gpu, oracle_mod, k):
    """isplib_hip_tune_experimental(12, 64) (libisplib_hip_exp.so): the task-list SDDMM in 64-column panels -- panel c adds its share of every dot product to
    what panels 0..c-1 stored (a tail under 4 columns joins the panel before it; rows that are not whole cache lines keep
    the whole-row form).  Against the oracle, on integer operands exactly, and bitwise reproducible."""
    from isplib_amd import cabi
    from isplib_amd.plan import build_task_plan
    rowptr, col = cases.random_csr(300, 900, 30.0, seed=k, empty_rows=(0, 299), hub=(9, 5000))
    x, g = cases.dense(900, k, 3), cases.dense(300, k, 5)
    xi, gi = np.round(x * 4).astype(np.float32), np.round(g * 4).astype(np.float32)
    d = [_t(a_, gpu) for a_ in (rowptr, col, x, g, xi, gi)]
    plan = build_task_plan(d[0], d[1], 900, 6, 256, 32)
    L = cabi.lib()
    try:
        for cols in (64, 128):
            assert cabi.exp_lib().isplib_hip_tune_experimental(12, cols) == 0
            for mean in (False, True):
                got = cabi.sddmm_tasks(d[0], d[1], plan, d[2], d[3], mean)
                ref = oracle_mod.sddmm(rowptr, col, x, g, mean=mean)
                scale = np.abs(ref).max() + 1e-6
                assert np.max(np.abs(got.cpu().numpy() - ref)) <= 2e-6 * scale * max(1, k / 16), (cols, mean)
                assert torch.equal(got, cabi.sddmm_tasks(d[0], d[1], plan, d[2], d[3], mean)), "bitwise reproducible"
            got = cabi.sddmm_tasks(d[0], d[1], plan, d[4], d[5], False)             # small integers: every order of summation is exact
            assert np.array_equal(got.cpu().numpy(), oracle_mod.sddmm(rowptr, col, xi, gi, mean=False)), cols
    finally:
        cabi.exp_lib().isplib_hip_tune_experimental(12, 0)


def test_value_gradient_through_planned_ops(gpu, oracle_mod, monkeypatch):
    import isplib_amd
    monkeypatch.setenv("ISPLIB_SLICES", "8")
    rowptr, col = cases.random_csr(150, 150, 80.0, seed=23)
    val, x, g = cases.weights(col.size, 4), cases.dense(150, 32, 3), cases.dense(150, 32, 5)
    for red, mean in (("sum", False), ("mean", True)):
        vs = _t(val, gpu).requires_grad_(True)
        adj = isplib_amd.SparseTensor.from_csr(_t(rowptr, gpu), _t(col, gpu), vs, (150, 150))
        xs = _t(x, gpu).requires_grad_(True)
        isplib_amd.matmul(adj, xs, red).backward(_t(g, gpu))
        _close(vs.grad, oracle_mod.sddmm(rowptr, col, x, g, mean=mean), rtol=1e-5, atol=1e-6)


def test_minmax_backward_kernel(gpu, oracle_mod):
    from isplib_amd import cabi
    rowptr, col = cases.random_csr(90, 60, 8.0, seed=12, empty_rows=(1, 89), duplicates=True)
    val = cases.weights(col.size, 4, "signed_int")
    x, g = cases.dense(60, 20, 3, "integer"), cases.dense(90, 20, 5, "integer")
    for red in ("max", "min"):
        _, arg = oracle_mod.spmm_fw(rowptr, col, val, x, red)
        r_dval, r_dx = oracle_mod.spmm_minmax_bw(col, val, x, arg, g)
        dval, dx = cabi.spmm_minmax_bw(_t(col, gpu), _t(val, gpu), _t(x, gpu), _t(arg, gpu), _t(g, gpu))
        assert np.array_equal(dx.cpu().numpy(), r_dx)        # integer data: sums exact in any order
        assert np.array_equal(dval.cpu().numpy(), r_dval)
        # unit weights (val == NULL) and single-output calls
        _, dx1 = cabi.spmm_minmax_bw(_t(col, gpu), None, _t(x, gpu), _t(arg, gpu), _t(g, gpu), need_val=False)
        _, r_dx1 = oracle_mod.spmm_minmax_bw(col, np.ones_like(val), x, arg, g)
        assert np.array_equal(dx1.cpu().numpy(), r_dx1)
        # the atomic-free form: the same exact answers on integer data
        dval, dx = cabi.spmm_minmax_bw(_t(col, gpu), _t(val, gpu), _t(x, gpu), _t(arg, gpu), _t(g, gpu), deterministic=True)
        assert np.array_equal(dx.cpu().numpy(), r_dx) and np.array_equal(dval.cpu().numpy(), r_dval)
        _, dx1 = cabi.spmm_minmax_bw(_t(col, gpu), None, _t(x, gpu), _t(arg, gpu), _t(g, gpu), need_val=False, deterministic=True)
        assert np.array_equal(dx1.cpu().numpy(), r_dx1)


def test_minmax_backward_without_atomics_is_bitwise_reproducible(gpu, oracle_mod):
    """Real-valued operands, many rows competing for few columns (long runs per destination), a hub column: the
    atomic-free backward gives the same bits on every launch, agrees with the oracle's CPU scatter (row-major order:
    the SAME order -- ascending row per destination -- so the match is exact, not just close), and the operator uses it."""
    from isplib_amd import cabi
    rng = np.random.default_rng(3)
    m, n, k = 4000, 37, 48
    deg = rng.integers(1, 9, m)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = np.concatenate([np.sort(rng.choice(n, d, replace=False)) for d in deg]).astype(np.int64)
    val = cases.weights(col.size, 4)
    x, g = cases.dense(n, k, 3), cases.dense(m, k, 5)
    for red in ("max", "min"):
        out, arg = oracle_mod.spmm_fw(rowptr, col, val, x, red)
        r_dval, r_dx = oracle_mod.spmm_minmax_bw(col, val, x, arg, g)
        args = (_t(col, gpu), _t(val, gpu), _t(x, gpu), _t(arg, gpu), _t(g, gpu))
        dval, dx = cabi.spmm_minmax_bw(*args, deterministic=True)
        for _ in range(3):
            dval2, dx2 = cabi.spmm_minmax_bw(*args, deterministic=True)
            assert torch.equal(dx.view(torch.int32), dx2.view(torch.int32)) and torch.equal(dval.view(torch.int32), dval2.view(torch.int32))
        assert np.allclose(dx.cpu().numpy(), r_dx, rtol=1e-5, atol=1e-5) and np.allclose(dval.cpu().numpy(), r_dval, rtol=1e-5, atol=1e-6)
        # through the operator (autograd): same bits as the direct call
        xs, vs = _t(x, gpu).requires_grad_(True), _t(val, gpu).requires_grad_(True)
        fn = torch.ops.isplib.fusedmm_spmm_max if red == "max" else torch.ops.isplib.fusedmm_spmm_min
        o, a = fn(_t(rowptr, gpu), _t(col, gpu), vs, xs)
        o.backward(_t(g, gpu))
        assert torch.equal(xs.grad, dx) and torch.equal(vs.grad, dval)


@pytest.mark.parametrize("shape", ((1, 1), (50, 50), (37, 200), (300, 19)))
def test_csr2csc_and_row_ids_on_device(gpu, oracle_mod, shape):
    from isplib_amd import cabi
    m, n = shape
    rowptr, col = cases.random_csr(m, n, 7.0, seed=m + n, empty_rows=(0,) if m > 1 else (), duplicates=True)
    val = cases.weights(col.size, 4)
    row, rowcount, colptr, csr2csc = oracle_mod.csr_transpose(rowptr, col, n)
    d_colptr, d_perm, d_row_t, d_val_t = cabi.csr2csc(_t(rowptr, gpu), _t(col, gpu), _t(val, gpu), n)
    assert np.array_equal(d_colptr.cpu().numpy(), colptr)
    assert np.array_equal(d_perm.cpu().numpy(), csr2csc)                # stable: torch_sparse order
    assert np.array_equal(d_row_t.cpu().numpy(), row[csr2csc])
    assert np.array_equal(d_val_t.cpu().numpy(), val[csr2csc])
    assert np.array_equal(cabi.csr_row_ids(_t(rowptr, gpu), col.size).cpu().numpy(), row)
    _, new_row, new_rowcount = oracle_mod.mean_bw_weights(rowptr, col, val, n)
    _, _, r2, v2 = cabi.csr2csc(_t(rowptr, gpu), _t(col, gpu), _t(val, gpu), n, mean_scale=True, want_perm=False)
    assert np.array_equal(r2.cpu().numpy(), new_row)
    assert np.array_equal(v2.cpu().numpy().view(np.uint32), new_rowcount.view(np.uint32))   # IEEE division both sides
    _, _, _, v3 = cabi.csr2csc(_t(rowptr, gpu), _t(col, gpu), None, n, mean_scale=True, want_perm=False, want_row=False)
    deg = np.maximum(rowcount, 1).astype(np.float32)[new_row]
    assert np.array_equal(v3.cpu().numpy(), (np.float32(1) / deg).astype(np.float32))


def test_plugin_matmul_forward_backward_all_reduces(gpu, oracle_mod, monkeypatch):
    import isplib_amd
    rowptr, col = cases.random_csr(120, 120, 70.0, seed=14, empty_rows=(5,))
    val = cases.weights(col.size, 4)
    x, g = cases.dense(120, 32, 3), cases.dense(120, 32, 5)
    tol = cases.sum_tolerance(oracle_mod, rowptr, col, val, x)
    for forced in ("0", "8"):                       # plain kernels, then the column-sliced ones
        monkeypatch.setenv("ISPLIB_SLICES", forced)
        adj = isplib_amd.SparseTensor.from_csr(_t(rowptr, gpu), _t(col, gpu), _t(val, gpu), (120, 120))
        for red in cases.REDUCES:
            xs = _t(x, gpu).requires_grad_(True)
            out = isplib_amd.matmul(adj, xs, red)
            assert isinstance(out, torch.Tensor)    # tensor, not the reference's (out, arg) tuple
            out.backward(_t(g, gpu))
            ref, ref_arg = oracle_mod.spmm_fw(rowptr, col, val, x, red)
            if red in ("sum", "mean"):
                assert np.all(np.abs(out.detach().cpu().numpy() - ref) <= tol)
                bw = oracle_mod.spmm_sum_bw if red == "sum" else oracle_mod.spmm_mean_bw
                _close(xs.grad, bw(rowptr, col, val, 120, g), rtol=1e-5, atol=2e-5)
            else:
                assert np.array_equal(out.detach().cpu().numpy(), ref)
                _, r_dx = oracle_mod.spmm_minmax_bw(col, val, x, ref_arg, g)
                _close(xs.grad, r_dx)
        # unweighted graph + 1-D feature vector
        adj1 = isplib_amd.SparseTensor.from_csr(_t(rowptr, gpu), _t(col, gpu), None, (120, 120))
        v = isplib_amd.matmul(adj1, _t(x[:, 0].copy(), gpu))
        ref, _ = oracle_mod.spmm_fw(rowptr, col, np.ones_like(val), x[:, :1].copy(), "sum")
        _close(v, ref[:, 0])


def test_plugin_serves_foreign_sparse_tensor_and_patches_torch_sparse_mm(gpu, oracle_mod):
    import isplib_amd

    class ForeignStorage:            # torch_sparse-like: no row_t()/val_t() helpers
        _row = _rowcount = _csr2csc = _colptr = None

    class Foreign:
        def __init__(self, rowptr, col, value, sizes):
            self._csr, self._sizes, self.storage = (rowptr, col, value), sizes, ForeignStorage()

        def csr(self):
            return self._csr

        def sparse_sizes(self):
            return self._sizes

    rowptr, col = cases.random_csr(40, 30, 5.0, seed=4)
    val = cases.weights(col.size, 4)
    x, g = cases.dense(30, 8, 3), cases.dense(40, 8, 5)
    src = Foreign(_t(rowptr, gpu), _t(col, gpu), _t(val, gpu), (40, 30))
    isplib_amd.iSpLibPlugin.patch_pyg()
    try:
        xs = _t(x, gpu).requires_grad_(True)
        out = torch.sparse.mm(src, xs)                   # reference patches torch.sparse.mm too (:178)
        out.backward(_t(g, gpu))
        out2 = torch.sparse.mm(src, xs.detach(), "max")
    finally:
        isplib_amd.iSpLibPlugin.unpatch_pyg()
    ref, _ = oracle_mod.spmm_fw(rowptr, col, val, x, "sum")
    _close(out, ref)
    _close(xs.grad, oracle_mod.spmm_sum_bw(rowptr, col, val, 30, g))
    assert np.array_equal(out2.cpu().numpy(), oracle_mod.spmm_fw(rowptr, col, val, x, "max")[0])


def test_non_contiguous_mat_is_accepted(gpu, oracle_mod):
    rowptr, col = cases.random_csr(30, 25, 4.0, seed=6)
    val = cases.weights(col.size, 4)
    x = cases.dense(25, 16, 3)
    xt = _t(np.ascontiguousarray(x.T), gpu).t()          # [25,16] view with stride (1,25)
    assert not xt.is_contiguous()
    out = torch.ops.isplib.fusedmm_spmm(None, _t(rowptr, gpu), _t(col, gpu), _t(val, gpu), None, None, xt, None, None)
    _close(out, oracle_mod.spmm_fw(rowptr, col, val, x, "sum")[0])


def test_two_epoch_gcn_loss_trajectory(gpu, oracle_mod):
    """Row H of SURVEY.md 8a: a 2-layer GCN (aggregate after the linear layer, K = hidden then classes)
    trained for 2 epochs on the HIP path; the trajectory is replayed on the CPU with the oracle doing
    every aggregation (forward and backward), same seed, and must agree within 1e-4 relative."""
    import isplib_amd
    rowptr, col = cases.random_csr(200, 200, 9.0, seed=33)
    n, f, h, c = 200, 24, 32, 7
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, f)).astype(np.float32)
    y = rng.integers(0, c, n)
    w1 = (rng.standard_normal((f, h)) * 0.2).astype(np.float32)
    w2 = (rng.standard_normal((h, c)) * 0.2).astype(np.float32)
    ones = np.ones(col.size, np.float32)

    class OracleAgg(torch.autograd.Function):
        @staticmethod
        def forward(ctx, m):
            return torch.from_numpy(oracle_mod.spmm_fw(rowptr, col, ones, m.detach().numpy(), "sum")[0])

        @staticmethod
        def backward(ctx, go):
            return torch.from_numpy(oracle_mod.spmm_sum_bw(rowptr, col, ones, n, go.numpy()))

    def run(dev, agg):
        a, b = torch.tensor(w1, device=dev, requires_grad=True), torch.tensor(w2, device=dev, requires_grad=True)
        opt = torch.optim.Adam([a, b], lr=0.01, weight_decay=5e-4)      # tests/cpu/gcn-sparse.py:79
        xs, ys, losses = torch.tensor(x, device=dev), torch.tensor(y, device=dev), []
        for _ in range(2):
            opt.zero_grad()
            hid = torch.relu(agg(xs @ a))
            loss = torch.nn.functional.nll_loss(torch.log_softmax(agg(hid @ b), 1), ys)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        return losses

    adj = isplib_amd.SparseTensor.from_csr(_t(rowptr, gpu), _t(col, gpu), None, (n, n))
    got = run(gpu, lambda m: isplib_amd.matmul(adj, m, "sum"))
    ref = run("cpu", OracleAgg.apply)
    assert np.allclose(got, ref, rtol=1e-4), (got, ref)


@pytest.mark.parametrize("model,aggr", (("sage", "mean"), ("sage", "max"), ("sage", "min"), ("sage", "sum"), ("gin", "sum")))
def test_two_epoch_sage_and_gin_loss_trajectories(gpu, oracle_mod, model, aggr):
    """The reference's other two callers (tests/cpu/graphSAGE-sparse.py:65-78, aggr sum | mean | max; tests/cpu/gin-sparse.py:59-78)
    as scripts/gcn_epoch.py restates them, trained for 2 epochs through the patched matmul on the HIP path, against the same
    two epochs on the CPU with the ORACLE doing every aggregation and the oracle's restatements of the reference's backward
    formulas (csrc/fusedmm.cpp:285, :375, :410-451) doing every gradient: losses within 1e-4 relative.  Weighted graph, so the
    mean backward's intended weight pairing and the max / min positions are all exercised inside a training loop."""
    import importlib.util
    import isplib_amd
    spec = importlib.util.spec_from_file_location("gcn_epoch", os.path.join(ROOT, "scripts", "gcn_epoch.py"))
    ge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ge)
    n, f, h, c = 220, 16, 24, 5
    rowptr, col = cases.random_csr(n, n, 7.0, seed=41, empty_rows=(3, 100), hub=(7, 150))
    val = (np.abs(cases.weights(col.size, 9)) + np.float32(0.1)).astype(np.float32)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((n, f)).astype(np.float32)
    y = torch.from_numpy(rng.integers(0, c, n))

    class OracleAgg(torch.autograd.Function):
        @staticmethod
        def forward(ctx, m, red):
            out, arg = oracle_mod.spmm_fw(rowptr, col, val, m.detach().numpy(), red)
            ctx.red, ctx.arg, ctx.m = red, arg, m.detach().numpy()
            return torch.from_numpy(out)

        @staticmethod
        def backward(ctx, go):
            g = np.ascontiguousarray(go.numpy())
            if ctx.red == "sum":
                return torch.from_numpy(oracle_mod.spmm_sum_bw(rowptr, col, val, n, g)), None
            if ctx.red == "mean":
                return torch.from_numpy(oracle_mod.spmm_mean_bw(rowptr, col, val, n, g)), None
            return torch.from_numpy(oracle_mod.spmm_minmax_bw(col, val, ctx.m, ctx.arg, g)[1]), None

    def make():
        torch.manual_seed(7)
        return (ge.SAGENet(f, h, c, aggr) if model == "sage" else ge.GINNet(f, h, c))

    def run(net, xs, ys, adj, agg):
        opt = torch.optim.Adam(net.parameters(), lr=0.01, weight_decay=5e-4)
        net.eval() if model == "sage" else net.train()      # (SAGE: dropout off, identical arithmetic; GIN: BatchNorm in training mode)
        losses = []
        for _ in range(2):
            opt.zero_grad()
            loss = torch.nn.functional.nll_loss(net(xs, adj, agg), ys)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        return losses

    ref = run(make(), torch.from_numpy(x), y, None, lambda a_, m_, r_: OracleAgg.apply(m_, r_))
    adj = isplib_amd.SparseTensor.from_csr(_t(rowptr, gpu), _t(col, gpu), _t(val, gpu), (n, n))
    isplib_amd.iSpLibPlugin.patch_pyg()
    try:
        got = run(make().to(gpu), _t(x, gpu), y.to(gpu), adj, lambda a_, m_, r_: torch.sparse.mm(a_, m_, r_))
    finally:
        isplib_amd.iSpLibPlugin.unpatch_pyg()
    assert np.allclose(got, ref, rtol=1e-4), (model, aggr, got, ref)


def test_boundary_calls_are_hipgraph_capturable(gpu, oracle_mod):
    """include/isplib_hip.h promises no allocation / synchronisation inside the entry points: capture the
    plain and the sliced SpMM (two kernels) into one graph on a side stream and replay it on new inputs."""
    from isplib_amd import cabi
    rowptr, col = cases.random_csr(300, 300, 70.0, seed=41)
    val = cases.weights(col.size, 4)
    d_rowptr, d_col, d_val = _t(rowptr, gpu), _t(col, gpu), _t(val, gpu)
    x = torch.zeros((300, 32), device=gpu)
    out_a = torch.empty((300, 32), device=gpu)
    out_b = torch.empty((300, 32), device=gpu)
    table, ok = cabi.spmm_slices(d_rowptr, d_col, 300, 8)
    work = cabi.sliced_workspace("sum", 300, 32, 8, gpu)
    assert ok
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        cabi.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, d_rowptr, d_col, d_val, x, out_a)      # warm-up outside capture
        with torch.cuda.graph(graph, stream=side):
            cabi.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, d_rowptr, d_col, d_val, x, out_a)
            cabi.fusedMM_csr_sliced_hip(cabi.MSG_SPMM_SUM, d_rowptr, d_col, d_val, table, 8, x, out_b, None, work)
    torch.cuda.current_stream().wait_stream(side)
    for seed in (3, 4):
        xs = cases.dense(300, 32, seed)
        x.copy_(_t(xs, gpu))
        graph.replay()
        torch.cuda.synchronize()
        ref, _ = oracle_mod.spmm_fw(rowptr, col, val, xs, "sum")
        tol = cases.sum_tolerance(oracle_mod, rowptr, col, val, xs)
        assert np.all(np.abs(out_a.cpu().numpy() - ref) <= tol)
        assert np.all(np.abs(out_b.cpu().numpy() - ref) <= tol)


def test_autotune_keeps_the_fastest_slice_count(gpu, oracle_mod):
    import isplib_amd
    rowptr, col = cases.random_csr(600, 600, 90.0, seed=17)
    val = cases.weights(col.size, 4)
    x = cases.dense(600, 64, 3)
    adj = isplib_amd.SparseTensor.from_csr(_t(rowptr, gpu), _t(col, gpu), _t(val, gpu), (600, 600))
    times = isplib_amd.iSpLibPlugin.autotune(adj, 64, "sum", candidates=(0, 8, 16))
    # (a graph this small is below the stream rule: the candidates are the plain kernel and the two task lists)
    assert set(times) == {("plain",), ("tasks", 8), ("tasks", 16)} and all(t > 0 for t in times.values())
    assert adj.storage._tuned[(600, 64, False)] == min(times, key=times.get)
    out = isplib_amd.matmul(adj, _t(x, gpu))
    assert adj.storage._last_schedule == min(times, key=times.get)
    ref, _ = oracle_mod.spmm_fw(rowptr, col, val, x, "sum")
    assert np.all(np.abs(out.cpu().numpy() - ref) <= cases.sum_tolerance(oracle_mod, rowptr, col, val, x))


@pytest.mark.parametrize("reduce", ("sum", "max"))
def test_autotune_sweeps_the_schedule_that_runs_and_replays_the_persisted_choice(gpu, oracle_mod, tmp_path, reduce):
    """on a graph the stream rule ACCEPTS (a tenth of the Reddit shape: 23 K rows, 11 M edges) the tuner
    must time distinct stream plans (slice counts around the rule's, other hub-row chunks), the task list and the plain
    kernel, each really forced -- `_last_schedule` is what ran --, persist (schedule, streams, slices, chunk), and a fresh
    graph object of the same content must replay that choice from the saved table.  Result within the oracle bound."""
    import isplib_amd
    from isplib_amd import plugin, synth
    rowptr, col, n = synth.dataset_like("reddit", device=gpu, scale=0.1)
    k = 64
    adj = isplib_amd.SparseTensor.from_csr(rowptr, col, None, (n, n), validate=False)
    rule = plugin.stream_minmax_rule(adj.storage, n, n, k) if reduce == "max" else plugin.stream_rule(adj.storage, n, n, k)
    assert rule is not None, "the test graph must be one the stream rule accepts"
    plugin._tuning_db.clear()
    times = isplib_amd.iSpLibPlugin.autotune(adj, k, reduce, candidates=(0, 4, 8))
    streams = [c for c in times if c[0] == "stream"]
    assert len({c[2] for c in streams}) >= 3 and len({c[3] for c in streams}) >= 2, times      # distinct slice counts AND chunks ran
    assert ("stream",) + tuple(rule) in times and ("plain",) in times and ("tasks", 4) in times and ("tasks", 8) in times
    assert len(set(round(t, 6) for t in times.values())) > 1
    best = min(times, key=times.get)
    assert adj.storage._tuned[(n, k, reduce == "max")] == best
    # only the winner's stream plan stays on the graph
    kept = [p for p in adj.storage._streams if adj.storage._streams[p] is not None]
    assert all(best[0] == "stream" and p[1:4] == tuple(best[1:]) for p in kept), (best, kept)
    x = synth.features(n, k, device=gpu, integer=reduce == "max")
    out = isplib_amd.matmul(adj, x, reduce)
    assert adj.storage._last_schedule == best
    isplib_amd.iSpLibPlugin.save_tuning(tmp_path / "tune.json")
    plugin._tuning_db.clear()
    fresh = isplib_amd.SparseTensor.from_csr(rowptr.clone(), col.clone(), None, (n, n), validate=False)
    assert isplib_amd.iSpLibPlugin.load_tuning(tmp_path / "tune.json") == 1
    out2 = isplib_amd.matmul(fresh, x, reduce)
    assert fresh.storage._last_schedule == best, "the persisted (schedule, geometry) must be what a fresh graph object runs"
    assert torch.equal(out, out2)
    rp, cl, xx = rowptr.cpu().numpy(), col.cpu().numpy(), x.cpu().numpy()
    ones = np.ones(cl.size, np.float32)
    ref, _ = oracle_mod.spmm_fw(rp, cl, ones, xx, reduce)
    if reduce == "max":
        assert np.array_equal(out.cpu().numpy(), ref)
    else:
        assert np.all(np.abs(out.cpu().numpy() - ref) <= cases.sum_tolerance(oracle_mod, rp, cl, ones, xx))
    plugin._tuning_db.clear()


def test_mtx_graph_tuning_file_and_generic_pipeline_through_the_package(gpu, oracle_mod, tmp_path):
    """The tuner's workflow end to end: .mtx adjacency -> SparseTensor -> autotune -> saved table -> a fresh graph
    object with the same content picks the persisted choice up; plus isplib_amd.fusedmm on the same graph."""
    import isplib_amd
    from isplib_amd import plugin
    rowptr, col = cases.random_csr(500, 500, 80.0, seed=23, empty_rows=(9,))
    val = cases.weights(col.size, 4)
    src = isplib_amd.SparseTensor.from_csr(_t(rowptr, gpu), _t(col, gpu), _t(val, gpu), (500, 500))
    src.to_mtx(tmp_path / "g.mtx", "tests")
    adj = isplib_amd.SparseTensor.from_mtx(tmp_path / "g.mtx", device=gpu)
    assert adj.sparse_sizes() == (500, 500) and torch.equal(adj.csr()[1], src.csr()[1])
    assert torch.allclose(adj.csr()[2], src.csr()[2], rtol=1e-6)           # %.9g round trip of fp32
    plugin._tuning_db.clear()
    times = isplib_amd.iSpLibPlugin.autotune(adj, 32, "sum", candidates=(0, 3, 8))
    best = min(times, key=times.get)
    isplib_amd.iSpLibPlugin.save_tuning(tmp_path / "tune.json")
    plugin._tuning_db.clear()
    fresh = isplib_amd.SparseTensor.from_mtx(tmp_path / "g.mtx", device=gpu)
    assert isplib_amd.iSpLibPlugin.load_tuning(tmp_path / "tune.json") == 1
    assert plugin.tuned_choice(fresh.storage, 500, 32) == best
    assert plugin.choose_slices(fresh.storage, 500, 32) == (best[1] if best[0] == "tasks" else 0)
    x = cases.dense(500, 32, 3)
    out = isplib_amd.matmul(fresh, _t(x, gpu))
    v = fresh.csr()[2].cpu().numpy()
    ref, _ = oracle_mod.spmm_fw(rowptr, col, v, x, "sum")
    assert np.all(np.abs(out.cpu().numpy() - ref) <= cases.sum_tolerance(oracle_mod, rowptr, col, v, x))
    plugin._tuning_db.clear()
    y = cases.dense(500, 32, 5) * np.float32(0.2)
    z = isplib_amd.fusedmm(fresh, _t(x, gpu) * 0.2, _t(y, gpu), "sigmoid_embedding")
    _, zref, _ = oracle_mod.fusedmm_general(0x11F12, rowptr, col, None, (x * np.float32(0.2)).astype(np.float32), y, 1)
    assert np.all(np.abs(z.cpu().numpy() - zref) <= 1e-4 * np.abs(zref).max())


def _gcn_reference(oracle_mod, rowptr, col, x, bias, relu):
    """relu(D^-1/2 (A + I) D^-1/2 x + b) composed from the oracle's unit-weight sum (fp32 steps like the kernel)."""
    deg = np.diff(rowptr).astype(np.float32) + 1
    dinv = (deg ** np.float32(-0.5)).astype(np.float32)
    y = (x * dinv[:, None]).astype(np.float32)
    agg, _ = oracle_mod.spmm_fw(rowptr, col, np.ones(col.size, np.float32), y, "sum")
    out = (agg + y) * dinv[:, None] + (0 if bias is None else bias)
    return (np.maximum(out, 0) if relu else out).astype(np.float32), dinv


@pytest.mark.parametrize("forced", ("0", "8"))
def test_fused_gcn_normalised_aggregation(gpu, oracle_mod, monkeypatch, forced):
    """D^-1/2 (A+I) D^-1/2 X (+ bias, ReLU) without materialised edge weights; the epilogue is
    applied by the fold kernel when a task plan is used (forced=8) and composed from ATen ops otherwise."""
    import isplib_amd
    monkeypatch.setenv("ISPLIB_SLICES", forced)
    n, k = 300, 40
    rowptr, col = cases.random_csr(n, n, 50.0, seed=19, empty_rows=(4,))
    x, g = cases.dense(n, k, 3), cases.dense(n, k, 5)
    bias = cases.dense(1, k, 7)[0]
    adj = isplib_amd.SparseTensor.from_csr(_t(rowptr, gpu), _t(col, gpu), None, (n, n))
    for use_bias, relu in ((False, False), (True, True)):
        xs = _t(x, gpu).requires_grad_(True)
        bs = _t(bias, gpu).requires_grad_(True) if use_bias else None
        out = isplib_amd.gcn_norm_matmul(adj, xs, bs, relu)
        out.backward(_t(g, gpu))
        ref, dinv = _gcn_reference(oracle_mod, rowptr, col, x, bias if use_bias else None, relu)
        _close(out, ref, rtol=1e-5, atol=1e-5)
        dz = g * (ref > 0) if relu else g
        gy = (dz * dinv[:, None]).astype(np.float32)
        dref = (oracle_mod.spmm_sum_bw(rowptr, col, np.ones(col.size, np.float32), n, gy) + gy) * dinv[:, None]
        _close(xs.grad, dref, rtol=1e-5, atol=1e-5)
        if use_bias:
            _close(bs.grad, dz.sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("geom", ("4:3:64", "8:2:300", "2:5:2048"))
def test_fused_gcn_normalised_aggregation_on_the_stream_schedule(gpu, oracle_mod, monkeypatch, geom):
    """The same fused layer where the stream rule accepts the graph (forced here through ISPLIB_STREAM_GEOM: the rule
    declines graphs this small): the stream kernel applies row scale, self term, bias and ReLU when it writes a finished
    row -- hub rows (one is cut into virtual rows here) in their fold -- forward on A, backward on A^T."""
    import isplib_amd
    monkeypatch.delenv("ISPLIB_SLICES", raising=False)
    monkeypatch.setenv("ISPLIB_STREAM_GEOM", geom)
    n, k = 300, 40
    rowptr, col = cases.random_csr(n, n, 50.0, seed=19, empty_rows=(4,), hub=(9, 280))
    x, g = cases.dense(n, k, 3), cases.dense(n, k, 5)
    bias = cases.dense(1, k, 7)[0]
    adj = isplib_amd.SparseTensor.from_csr(_t(rowptr, gpu), _t(col, gpu), None, (n, n))
    for use_bias, relu in ((False, False), (True, True)):
        xs = _t(x, gpu).requires_grad_(True)
        bs = _t(bias, gpu).requires_grad_(True) if use_bias else None
        out = isplib_amd.gcn_norm_matmul(adj, xs, bs, relu)
        out.backward(_t(g, gpu))
        ref, dinv = _gcn_reference(oracle_mod, rowptr, col, x, bias if use_bias else None, relu)
        _close(out, ref, rtol=1e-5, atol=1e-5)
        dz = g * (ref > 0) if relu else g
        gy = (dz * dinv[:, None]).astype(np.float32)
        dref = (oracle_mod.spmm_sum_bw(rowptr, col, np.ones(col.size, np.float32), n, gy) + gy) * dinv[:, None]
        _close(xs.grad, dref, rtol=1e-5, atol=1e-5)
        if use_bias:
            _close(bs.grad, dz.sum(0), rtol=1e-5, atol=1e-4)
    assert any(key[0] is False for key in adj.storage._streams) and any(key[0] is True for key in adj.storage._streams), \
        "both directions must have run on stream plans"


@pytest.mark.parametrize("n,k,pitch", ((1000, 41, 48), (513, 7, None), (700, 300, None), (3, 64, 64), (2049, 32, None)))
def test_gcn_dense_passes_match_torch(gpu, n, k, pitch):
    """isplib_row_scale_hip / isplib_masked_scale_colsum_hip (the passes either side of the fused GCN aggregation) against the
    ATen expressions they replace: ragged row counts (blocks of 512 rows), widths below, at and above the 256 column lanes,
    a wider output pitch (its padding must be written, as zeros), every optional operand absent in turn; the column sums
    twice for bitwise reproducibility."""
    from isplib_amd import cabi
    gen = torch.Generator(device=gpu)
    gen.manual_seed(n * 1000 + k)
    x = torch.randn((n, k), generator=gen, device=gpu)
    dz = torch.randn((n, k), generator=gen, device=gpu)
    out = torch.relu(torch.randn((n, k), generator=gen, device=gpu))
    scale = torch.rand(n, generator=gen, device=gpu) + 0.1
    y = cabi.row_scale(x, scale, pitch)
    assert torch.equal(y, x * scale[:, None])
    if pitch and pitch > k:
        assert y.stride(0) == pitch and bool((torch.as_strided(y, (n, pitch - k), (pitch, 1), k) == 0).all())
    wide = torch.randn((n, k + 5), generator=gen, device=gpu)[:, 2:2 + k]               # a row-strided input
    assert torch.equal(cabi.row_scale(wide, scale), wide * scale[:, None])
    for mask, sc in ((out, scale), (None, scale), (out, None), (None, None)):
        g = dz if mask is None else dz * (mask > 0)
        gy, gb = cabi.masked_scale_colsum(dz, mask, sc, pitch=pitch)
        assert torch.equal(gy, g if sc is None else g * sc[:, None])
        want = g.double().sum(0)
        bound = 1e-6 * g.double().abs().sum(0) + 1e-30
        assert bool(((gb.double() - want).abs() <= bound).all())
        gy2, gb2 = cabi.masked_scale_colsum(dz, mask, sc, pitch=pitch)
        assert torch.equal(gb.view(torch.int32), gb2.view(torch.int32)), "column sums must be bitwise reproducible"
        _, only_bias = cabi.masked_scale_colsum(dz, mask, sc, want_gy=False)
        assert torch.equal(only_bias, gb)
        only_gy, none = cabi.masked_scale_colsum(dz, mask, sc, want_bias=False)
        assert none is None and torch.equal(only_gy, gy)


def test_task_epilogue_entry_point(gpu, oracle_mod):
    from isplib_amd import cabi
    from isplib_amd.plan import build_task_plan
    rowptr, col = cases.random_csr(200, 260, 30.0, seed=29, empty_rows=(0,), hub=(5, 200))
    x = cases.dense(260, 300, 3)                      # 300 columns: three 128-column panels
    self_term, rs, bias = cases.dense(200, 300, 4), cases.dense(200, 1, 5)[:, 0].copy(), cases.dense(1, 300, 6)[0]
    plan = build_task_plan(_t(rowptr, gpu), _t(col, gpu), 260, 8, 64, 16)
    for one_pass_kib in (0, 9216):               # 0: forced through the column panels (a graph this small would not use them)
        cabi.lib().isplib_hip_tune(8, one_pass_kib)
        for red in ("sum", "mean"):
            got = cabi.spmm_tasks_epilogue(_t(rowptr, gpu), _t(col, gpu), None, plan, _t(x, gpu), red, _t(rs, gpu),
                                           _t(self_term, gpu), _t(bias, gpu), True)
            base, _ = oracle_mod.spmm_fw(rowptr, col, np.ones(col.size, np.float32), x, red)
            ref = np.maximum((base + self_term) * rs[:, None] + bias, 0)
            _close(got, ref, rtol=1e-5, atol=2e-5)
    with pytest.raises(RuntimeError, match="sum / mean only"):
        cabi.spmm_tasks_epilogue(_t(rowptr, gpu), _t(col, gpu), None, plan, _t(x, gpu), "max", relu=True)


def test_graph_handle_runs_the_fast_path_for_torch_free_hosts(gpu, oracle_mod):
    """isplib_graph (include/isplib_hip.h): the library-owned per-graph state -- plans, packed ids, CSC operands,
    workspace -- behind create / spmm / spmm_backward / destroy, against the oracle for every reduction and for the
    sum and mean backward; plain kernel, rule and forced slice counts give the same answers."""
    from isplib_amd import cabi
    rowptr, col = cases.random_csr(300, 260, 40.0, seed=12, empty_rows=(0, 299), hub=(17, 3000), duplicates=True)
    val = cases.weights(col.size, 4)
    x, g = cases.dense(260, 48, 3), cases.dense(300, 48, 5)
    for weights in (val, None):
        h = cabi.GraphHandle(_t(rowptr, gpu), _t(col, gpu), None if weights is None else _t(weights, gpu), 260)
        hv = weights if weights is not None else np.ones(col.size, np.float32)
        tol = cases.sum_tolerance(oracle_mod, rowptr, col, hv, x)
        for slices in (-1, 0, 1, 7):
            h.set_slices(slices)
            for red in cases.REDUCES:
                ref, ref_arg = oracle_mod.spmm_fw(rowptr, col, hv, x, red)
                out, arg = h.spmm(_t(x, gpu), red)
                if red in ("max", "min"):
                    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32)), (red, slices)
                    assert np.array_equal(arg.cpu().numpy(), ref_arg), (red, slices)
                else:
                    deg = np.maximum(np.diff(rowptr), 1)[:, None] if red == "mean" else 1
                    assert np.all(np.abs(out.cpu().numpy() - ref) <= tol / deg + 1e-12), (red, slices)
            dx = h.spmm_backward(_t(g, gpu)).cpu().numpy()
            dref = oracle_mod.spmm_sum_bw(rowptr, col, hv, 260, g)
            dmag = oracle_mod.spmm_sum_bw(rowptr, col, np.abs(hv), 260, np.abs(g))
            assert np.all(np.abs(dx - dref) <= 1e-5 * dmag + 1e-30), slices
            dxm = h.spmm_backward(_t(g, gpu), mean=True).cpu().numpy()
            mref = oracle_mod.spmm_mean_bw(rowptr, col, hv, 260, g)
            assert np.all(np.abs(dxm - mref) <= 1e-5 * dmag + 1e-30), slices
            for mean in (False, True):
                da = h.sddmm(_t(x, gpu), _t(g, gpu), mean=mean).cpu().numpy()
                assert np.allclose(da, oracle_mod.sddmm(rowptr, col, x, g, mean=mean), rtol=1e-5, atol=1e-5), (slices, mean)
        h.close()
        h.close()                                                   # idempotent
    with pytest.raises(RuntimeError):
        cabi.GraphHandle(_t(rowptr, gpu), _t(col, gpu), None, 2 ** 31 + 5)


@pytest.mark.parametrize("k", (32, 48))
def test_graph_handle_takes_new_weights_without_rebuilding(gpu, oracle_mod, k):
    """isplib_graph_set_values on a graph large enough for the stream schedule (sum / mean forward and both backwards go
    through stream plans that own a copy of the weights, and so does max -- on 32-column slots at k = 32, 64-column ones at 48): another array, the same array edited in
    place, no weights at all -- every call after it must see the new weights, forward and backward, with the plans kept."""
    from isplib_amd import cabi
    n = 20000
    rowptr, col = cases.random_csr(n, n, 230.0, seed=41, empty_rows=(7,), hub=(3, 15000))
    assert col.size >= (1 << 22) and cabi.suggest_stream(n, n, col.size, k) is not None
    assert cabi.suggest_stream_minmax(n, n, col.size, k)[0] == (8 if k <= 32 else 4)    # max rides a stream plan too
    x, g = cases.dense(n, k, 3), cases.dense(n, k, 5)
    w1, w2 = cases.weights(col.size, 4), cases.weights(col.size, 9)
    d_w = _t(w1, gpu)
    h = cabi.GraphHandle(_t(rowptr, gpu), _t(col, gpu), d_w, n)

    def check(w):
        hv = np.ones(col.size, np.float32) if w is None else w
        for red in ("sum", "mean", "max"):
            out, arg = h.spmm(_t(x, gpu), red)
            ref, ref_arg = oracle_mod.spmm_fw(rowptr, col, hv, x, red)
            if red == "max":
                assert np.array_equal(out.cpu().numpy(), ref) and np.array_equal(arg.cpu().numpy(), ref_arg)
            else:
                deg = np.maximum(np.diff(rowptr), 1)[:, None] if red == "mean" else 1
                tol = cases.sum_tolerance(oracle_mod, rowptr, col, hv, x)
                assert np.all(np.abs(out.cpu().numpy() - ref) <= tol / deg + 1e-12), red
        dmag = oracle_mod.spmm_sum_bw(rowptr, col, np.abs(hv), n, np.abs(g))
        dx = h.spmm_backward(_t(g, gpu)).cpu().numpy()
        assert np.all(np.abs(dx - oracle_mod.spmm_sum_bw(rowptr, col, hv, n, g)) <= 1e-5 * dmag + 1e-30)
        dxm = h.spmm_backward(_t(g, gpu), mean=True).cpu().numpy()
        assert np.all(np.abs(dxm - oracle_mod.spmm_mean_bw(rowptr, col, hv, n, g)) <= 1e-5 * dmag + 1e-30)

    check(w1)
    d_w2 = _t(w2, gpu)
    h.set_values(d_w2)                  # another array
    check(w2)
    d_w2.mul_(-0.5)                     # the same array, edited in place
    h.set_values(d_w2)
    check((w2 * np.float32(-0.5)).astype(np.float32))
    h.set_values(None)                  # unit weights
    check(None)
    h.set_values(d_w)                   # and back
    check(w1)
    h.close()


def test_reference_schema_ops_get_the_task_schedule_through_cached_handles(gpu, oracle_mod):
    """INTEGRATION.md option B: iSpLib's own Python calls torch.ops.isplib.fusedmm_spmm* with just the CSR arrays and
    its cached CSC operands.  On a graph large enough for the slice rule, the operator library serves it from a
    per-graph isplib_graph handle (forward on A, backward on the cached A^T operands), validated by tensor identity."""
    import isplib_amd  # noqa: F401
    ops = torch.ops.isplib
    ops.graph_cache_clear()
    n, k = 4000, 64
    rowptr, col = cases.random_csr(n, n, 300.0, seed=31, empty_rows=(5,))
    assert col.size >= (1 << 20)
    val = cases.weights(col.size, 4)
    x, g = cases.dense(n, k, 3), cases.dense(n, k, 5)
    d_rowptr, d_col, d_val = _t(rowptr, gpu), _t(col, gpu), _t(val, gpu)
    from isplib_amd import cabi
    colptr, perm, row_t, val_t = cabi.csr2csc(d_rowptr, d_col, d_val, n)      # what isplib/__init__.py:76-80 caches
    row = cabi.csr_row_ids(d_rowptr, col.size)
    tol = cases.sum_tolerance(oracle_mod, rowptr, col, val, x)
    ref, _ = oracle_mod.spmm_fw(rowptr, col, val, x, "sum")
    dref = oracle_mod.spmm_sum_bw(rowptr, col, val, n, g)
    dmag = oracle_mod.spmm_sum_bw(rowptr, col, np.abs(val), n, np.abs(g))
    for it in range(2):
        xs = _t(x, gpu).requires_grad_(True)
        out = ops.fusedmm_spmm(row, d_rowptr, d_col, d_val, colptr, perm, xs, val_t, row_t)
        out.backward(_t(g, gpu))
        assert np.all(np.abs(out.detach().cpu().numpy() - ref) <= tol)
        assert np.all(np.abs(xs.grad.cpu().numpy() - dref) <= 1e-5 * dmag + 1e-30)
        assert ops.graph_cache_size() == 2, "one handle for A, one for the cached A^T operands; reused on the second pass"
    # dA (the SDDMM the reference leaves commented out) runs off the same handle of A (its structure is the key; the
    # weights are a per-call argument), on a plan sized for whole rows
    d_val.requires_grad_(True)
    out = ops.fusedmm_spmm(row, d_rowptr, d_col, d_val, colptr, perm, _t(x, gpu), val_t, row_t)
    out.backward(_t(g, gpu))
    assert np.allclose(d_val.grad.cpu().numpy(), oracle_mod.sddmm(rowptr, col, x, g), rtol=1e-5, atol=1e-4)
    assert ops.graph_cache_size() == 2
    d_val = d_val.detach()        # same storage, new tensor object: the handle is told about it, not replaced
    mx, arg = ops.fusedmm_spmm_max(d_rowptr, d_col, d_val, _t(x, gpu))
    rmx, rarg = oracle_mod.spmm_fw(rowptr, col, val, x, "max")
    assert np.array_equal(mx.cpu().numpy(), rmx) and np.array_equal(arg.cpu().numpy(), rarg)
    assert ops.graph_cache_size() == 2
    # an in-place edit of the weights (an optimiser step: the version counter moves) keeps the handle and its plans;
    # the answer follows the edit
    d_val.mul_(2.0)
    out2 = ops.fusedmm_spmm_max(d_rowptr, d_col, d_val, _t(x, gpu))[0]
    rmx2, _ = oracle_mod.spmm_fw(rowptr, col, (val * 2).astype(np.float32), x, "max")
    assert np.array_equal(out2.cpu().numpy(), rmx2)
    out2 = ops.fusedmm_spmm(row, d_rowptr, d_col, d_val, colptr, perm, _t(x, gpu), val_t, row_t)
    assert np.all(np.abs(out2.cpu().numpy() - 2 * ref) <= 2 * tol)
    # another weights tensor altogether, then none at all (unit weights), on the same structure: still one handle of A
    w3 = cases.weights(col.size, 21)
    out3 = ops.fusedmm_spmm_min(d_rowptr, d_col, _t(w3, gpu), _t(x, gpu))[0]
    assert np.array_equal(out3.cpu().numpy(), oracle_mod.spmm_fw(rowptr, col, w3, x, "min")[0])
    assert ops.graph_cache_size() == 2
    # an in-place edit of the STRUCTURE retires the handle (version counter of col), and the answer follows the edit
    d_col.copy_((d_col + 1) % n)
    col_now = d_col.cpu().numpy()
    out4 = ops.fusedmm_spmm_max(d_rowptr, d_col, d_val, _t(x, gpu))[0]
    rmx4, _ = oracle_mod.spmm_fw(rowptr, col_now, (val * 2).astype(np.float32), x, "max")
    assert np.array_equal(out4.cpu().numpy(), rmx4)
    # graphs whose tensors are gone are dropped the next time a new graph arrives
    del d_rowptr, d_col, d_val, colptr, perm, row_t, val_t, row, out, xs
    r2, c2 = cases.random_csr(n, n, 300.0, seed=32)
    out3 = ops.fusedmm_spmm_max(_t(r2, gpu), _t(c2, gpu), None, _t(x, gpu))[0]
    rmx3, _ = oracle_mod.spmm_fw(r2, c2, np.ones(c2.size, np.float32), x, "max")
    assert np.array_equal(out3.cpu().numpy(), rmx3)
    assert ops.graph_cache_size() <= 1
    ops.graph_cache_clear()
    assert ops.graph_cache_size() == 0


def test_config1_cora_k16_through_patch_pyg(gpu, oracle_mod, monkeypatch):
    """BASELINE.json configs[0]: the Cora-shaped adjacency (2,708 nodes, 10,556 nnz), K=16, through the reference's
    drop-in surface -- iSpLibPlugin.patch_pyg() and then torch_sparse.matmul / torch.sparse.mm exactly as PyG calls
    them (isplib/__init__.py:140-151,177-178) -- forward and backward, all four reductions, against the oracle.
    torch_sparse is not part of this image: a stand-in module object takes its place for the patch, and the graph is
    a duck-typed object with the members the wrapper reads (csr(), storage, sparse_sizes())."""
    import types

    import isplib_amd
    from isplib_amd import plugin, synth

    class ForeignStorage:
        _row = _rowcount = _csr2csc = _colptr = None

    class Foreign:
        def __init__(self, rowptr, col, value, sizes):
            self._csr, self._sizes, self.storage = (rowptr, col, value), sizes, ForeignStorage()

        def csr(self):
            return self._csr

        def sparse_sizes(self):
            return self._sizes

    fake_ts = types.SimpleNamespace(matmul=lambda *a, **k: (_ for _ in ()).throw(AssertionError("unpatched torch_sparse.matmul")))
    monkeypatch.setattr(plugin, "_torch_sparse", fake_ts)
    rowptr_t, col_t, n = synth.dataset_like("cora", device="cpu")
    assert n == 2708 and col_t.numel() == 10556
    rowptr, col = rowptr_t.numpy(), col_t.numpy()
    k = 16
    x = synth.features(n, k, device="cpu").numpy()
    g = synth.features(n, k, seed=5, device="cpu").numpy()
    for weighted in (False, True):
        val = synth.edge_weights(col.size, device="cpu").numpy() if weighted else np.ones(col.size, np.float32)
        src = Foreign(_t(rowptr, gpu), _t(col, gpu), _t(val, gpu) if weighted else None, (n, n))
        tol = cases.sum_tolerance(oracle_mod, rowptr, col, val, x)
        gtol = cases.sum_tolerance(oracle_mod, *_transpose_np(oracle_mod, rowptr, col, val, n), g)
        isplib_amd.iSpLibPlugin.patch_pyg()
        try:
            assert fake_ts.matmul is plugin.spmm_autotuned
            for red in cases.REDUCES:
                for entry in (fake_ts.matmul, torch.sparse.mm):
                    xs = _t(x, gpu).requires_grad_(True)
                    out = entry(src, xs, red) if red != "sum" or entry is fake_ts.matmul else entry(src, xs)
                    out.backward(_t(g, gpu))
                    ref, ref_arg = oracle_mod.spmm_fw(rowptr, col, val, x, red)
                    got, dx = out.detach().cpu().numpy(), xs.grad.cpu().numpy()
                    if red in ("sum", "mean"):
                        assert np.all(np.abs(got - ref) <= tol), red
                        bw = oracle_mod.spmm_sum_bw if red == "sum" else oracle_mod.spmm_mean_bw
                        assert np.all(np.abs(dx - bw(rowptr, col, val, n, g)) <= gtol), red
                    else:
                        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), red
                        _, r_dx = oracle_mod.spmm_minmax_bw(col, val, x, ref_arg, g)
                        _close(dx, r_dx, rtol=1e-5, atol=1e-5)
        finally:
            isplib_amd.iSpLibPlugin.unpatch_pyg()
    assert not isplib_amd.iSpLibPlugin.is_patched()


def _transpose_np(oracle_mod, rowptr, col, val, ncols):
    row, _, colptr, csr2csc = oracle_mod.csr_transpose(rowptr, col, ncols)
    return colptr, row[csr2csc], np.asarray(val, np.float32)[csr2csc]


def test_backward_sees_edge_weights_updated_in_place(gpu, oracle_mod):
    """Trainable edge weights stepped in place between two iterations: the second backward must use the NEW
    weights in A^T (the cached value[csr2csc] / mean weights are rebuilt when `value` changes), for sum and mean."""
    import isplib_amd
    rowptr, col = cases.random_csr(90, 70, 8.0, seed=12, empty_rows=(3,))
    x, g = cases.dense(70, 24, 3), cases.dense(90, 24, 5)
    val0 = cases.weights(col.size, 4)
    val1 = cases.weights(col.size, 99)
    for red, bw in (("sum", oracle_mod.spmm_sum_bw), ("mean", oracle_mod.spmm_mean_bw)):
        value = _t(val0, gpu)
        adj = isplib_amd.SparseTensor.from_csr(_t(rowptr, gpu), _t(col, gpu), value, (90, 70))
        for cur in (val0, val1):
            xs = _t(x, gpu).requires_grad_(True)
            isplib_amd.matmul(adj, xs, red).backward(_t(g, gpu))
            _close(xs.grad, bw(rowptr, col, cur, 70, g), rtol=1e-5, atol=2e-5)
            value.copy_(_t(val1, gpu))              # what optimizer.step() does to a trainable weight vector


@pytest.mark.parametrize("geom", ("4:3:64", "4:1:2048", "8:7:300"))
def test_plugin_runs_max_and_min_on_the_stream_schedule(gpu, oracle_mod, monkeypatch, geom):
    """The plug-in's default for max / min on large graphs with column-sorted rows, forced onto a small one
    (ISPLIB_STREAM_MINMAX_GEOM): values and arg bit for bit (integer operands: ties everywhere, a hub row cut into virtual
    rows), the backward scatter through the returned arg, weights replaced in place picked up by the plan; an unsorted
    graph quietly stays on the task list."""
    import isplib_amd
    monkeypatch.setenv("ISPLIB_STREAM_MINMAX_GEOM", geom)
    rowptr, col = cases.random_csr(150, 110, 30.0, seed=43, empty_rows=(5,), hub=(9, 900), duplicates=True)
    x, g = cases.dense(110, 40, 3, "integer"), cases.dense(150, 40, 5)
    val0, val1 = cases.weights(col.size, 4, "signed_int"), cases.weights(col.size, 98, "signed_int")
    for weighted in (True, False):
        value = _t(val0, gpu) if weighted else None
        adj = isplib_amd.SparseTensor.from_csr(_t(rowptr, gpu), _t(col, gpu), value, (150, 110))
        for cur in ((val0, val1) if weighted else (np.ones(col.size, np.float32),)):
            for red in ("max", "min"):
                xs = _t(x, gpu).requires_grad_(True)
                out = isplib_amd.matmul(adj, xs, red)
                out.backward(_t(g, gpu))
                ref, ref_arg = oracle_mod.spmm_fw(rowptr, col, cur, x, red)
                assert np.array_equal(out.detach().cpu().numpy().view(np.uint32), ref.view(np.uint32)), (red, weighted)
                _, gm = oracle_mod.spmm_minmax_bw(col, cur, x, ref_arg, g)
                _close(xs.grad, gm, rtol=1e-5, atol=1e-5)
                # nothing to differentiate: the values-only launch (torch.ops.isplib.fusedmm_spmm_{max,min}_values), same bits
                plain = isplib_amd.matmul(adj, _t(x, gpu), red)
                assert not plain.requires_grad and torch.equal(plain.view(torch.int32), out.detach().view(torch.int32)), (red, weighted)
            if weighted:
                value.copy_(_t(val1, gpu))
        assert any(k[-1] == "minmax" and adj.storage._streams[k] is not None for k in adj.storage._streams), "the max / min stream plan was not used"
    # rows that are not column-sorted: the plan builder declines, the call runs on the task list / plain kernel
    rowptr, col = cases.random_csr(150, 110, 30.0, seed=44, sort_cols=False)
    adj = isplib_amd.SparseTensor.from_csr(_t(rowptr, gpu), _t(col, gpu), None, (150, 110), validate=False)
    out = isplib_amd.matmul(adj, _t(x, gpu), "max")
    ref, _ = oracle_mod.spmm_fw(rowptr, col, np.ones(col.size, np.float32), x, "max")
    assert np.array_equal(out.cpu().numpy(), ref)
    assert all(adj.storage._streams[k] is None for k in adj.storage._streams if k[-1] == "minmax")


@pytest.mark.parametrize("k", (41, 47, 33))
def test_operator_layer_copies_33_to_47_columns_to_a_192_byte_pitch(gpu, oracle_mod, monkeypatch, k):
    """Round 4: on a stream plan the operator layer gathers a 33..47-column operand from a copy at a 192-byte row pitch (whole
    cache lines: the GCN's K=41 aggregation, 1.365 -> 1.285 ms on the Reddit shape) -- graphs of at least 65,536 columns;
    the output and the gradients stay packed.  Forward and backward of sum / mean / max through the plug-in on such a graph
    (forced onto stream plans), against the oracle; the same call on a strided `other` must see through the stride."""
    import isplib_amd
    monkeypatch.setenv("ISPLIB_STREAM_GEOM", "4:6:300")
    monkeypatch.setenv("ISPLIB_STREAM_MINMAX_GEOM", "4:4:300")
    n = 66000
    rowptr, col = cases.random_csr(n, n, 9.0, seed=k, empty_rows=(0, n - 1), hub=(17, 4000))
    val = cases.weights(col.size, 4)
    x, g = cases.dense(n, k, 3), cases.dense(n, k, 5)
    adj = isplib_amd.SparseTensor.from_csr(_t(rowptr, gpu), _t(col, gpu), _t(val, gpu), (n, n))
    wide = torch.zeros((n, 64), device=gpu)
    wide[:, :k] = _t(x, gpu)
    for red in ("sum", "mean", "max"):
        ref, ref_arg = oracle_mod.spmm_fw(rowptr, col, val, x, red)
        for other in (_t(x, gpu), wide[:, :k]):
            xs = other.detach().requires_grad_(True)
            out = isplib_amd.matmul(adj, xs, red)
            assert out.is_contiguous() and out.shape == (n, k)
            out.backward(_t(g, gpu))
            got = out.detach().cpu().numpy()
            if red == "max":
                assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
                _, gm = oracle_mod.spmm_minmax_bw(col, val, x, ref_arg, g)
                _close(xs.grad, gm, rtol=1e-5, atol=1e-5)
            else:
                deg = np.maximum(np.diff(rowptr), 1)[:, None] if red == "mean" else 1
                assert np.all(np.abs(got - ref) <= cases.sum_tolerance(oracle_mod, rowptr, col, val, x) / deg + 1e-12), red
                bw = oracle_mod.spmm_mean_bw if red == "mean" else oracle_mod.spmm_sum_bw
                dmag = oracle_mod.spmm_sum_bw(rowptr, col, np.abs(val), n, np.abs(g))
                assert np.all(np.abs(xs.grad.cpu().numpy() - bw(rowptr, col, val, n, g)) <= 1e-5 * dmag + 1e-30), red
    assert any(p is not None for p in adj.storage._streams.values()), "the stream plans were not used"


@pytest.mark.parametrize("geom", ("4:3:64", "8:2:100000", "2:5:300"))
def test_plugin_runs_sum_and_mean_on_the_stream_schedule(gpu, oracle_mod, monkeypatch, geom):
    """The plug-in's default for sum / mean on large graphs, forced onto a small one (ISPLIB_STREAM_GEOM): forward through
    a stream plan of A, backward through a stream plan of A^T whose weights are value[csr2csc] (sum) or the mean
    backward's value[csr2csc] / max(deg, 1); trainable weights stepped in place are picked up by both plans."""
    import isplib_amd
    monkeypatch.setenv("ISPLIB_STREAM_GEOM", geom)
    rowptr, col = cases.random_csr(150, 110, 30.0, seed=41, empty_rows=(5,), hub=(9, 900))
    x, g = cases.dense(110, 64, 3), cases.dense(150, 64, 5)
    val0, val1 = cases.weights(col.size, 4), cases.weights(col.size, 99)
    tol = cases.sum_tolerance(oracle_mod, rowptr, col, np.maximum(val0, val1), x)
    for weighted in (True, False):
        for red, bw in (("sum", oracle_mod.spmm_sum_bw), ("mean", oracle_mod.spmm_mean_bw)):
            value = _t(val0, gpu) if weighted else None
            adj = isplib_amd.SparseTensor.from_csr(_t(rowptr, gpu), _t(col, gpu), value, (150, 110))
            for cur in ((val0, val1) if weighted else (np.ones(col.size, np.float32),)):
                xs = _t(x, gpu).requires_grad_(True)
                out = isplib_amd.matmul(adj, xs, red)
                out.backward(_t(g, gpu))
                ref, _ = oracle_mod.spmm_fw(rowptr, col, cur, x, red)
                assert np.all(np.abs(out.detach().cpu().numpy() - ref) <= tol), (red, weighted)
                _close(xs.grad, bw(rowptr, col, cur, 110, g), rtol=1e-5, atol=2e-5)
                if weighted:
                    value.copy_(_t(val1, gpu))
            assert any(k[0] is False for k in adj.storage._streams) and any(k[0] is True for k in adj.storage._streams), "stream plans were not used"
            if not weighted and red == "mean":      # unweighted mean: 1 / deg is applied to the rows of dY, no edge weights anywhere
                assert adj.storage._mean_val_t is None and not adj.storage._stream_vals
