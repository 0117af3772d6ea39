"""GPU parity of the sweep schedule (fusedMM_csr_sweep_hip: rows resident in LDS, no partial rows) against the CPU
oracle, through the C ABI.  Bar as everywhere: max/min values and arg indices bit-exact, sum/mean within
1e-5 * sum|val*x| per element.  Small seeded cases here; the full-size ones are in test_gpu_fullsize.py."""
import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _check(oracle, rowptr, col, val, x, red, out, arg, tol=None):
    ref, ref_arg = oracle.spmm_fw(rowptr, col, val, x, red)
    out = out.cpu().numpy()
    if red in ("sum", "mean"):
        tol = cases.sum_tolerance(oracle, rowptr, col, val, x) if tol is None else tol
        fin = np.isfinite(ref) & np.isfinite(tol)
        assert np.array_equal(np.isnan(out[~fin]), np.isnan(ref[~fin]))
        err = np.abs(out[fin].astype(np.float64) - ref[fin].astype(np.float64))
        assert np.all(err <= tol[fin]), f"{red}: max err/tol = {np.max(err / tol[fin])}"
    else:
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), f"{red}: values not bit-exact"
        assert np.array_equal(arg.cpu().numpy(), ref_arg), f"{red}: arg indices differ"


def _sweep_all(gpu, oracle, rowptr, col, val, x, unit=False, geoms=((8, 64, 16, 64, 4), (5, 24, 8, 2048, 16), (16, 7, 32, 100, 1))):
    """geoms: (slices, waves_per_gen, rows_per_wave, chunk, min_seg)"""
    from isplib_amd import cabi
    from isplib_amd.plan import build_sweep_plan
    d_rowptr, d_col, d_x = _t(rowptr, gpu), _t(col, gpu), _t(x, gpu)
    d_val = None if unit else _t(val, gpu)
    for (s, wpg, rpw, chunk, min_seg) in geoms:
        plan = build_sweep_plan(d_rowptr, d_col, x.shape[0], s, wpg, rpw, chunk, min_seg)
        assert plan is not None
        assert int((plan.task_meta & 0xFFFFFF).sum()) == col.size
        for red in cases.REDUCES:
            if red in ("max", "min") and rpw > 16:
                continue
            out, arg = cabi.spmm_sweep(d_rowptr, d_col, d_val, plan, d_x, red)
            again, _ = cabi.spmm_sweep(d_rowptr, d_col, d_val, plan, d_x, red)
            torch.cuda.synchronize()
            assert torch.equal(out.view(torch.int32), again.view(torch.int32)), "sweep schedule must be bitwise reproducible"
            _check(oracle, rowptr, col, val, x, red, out, arg)


@pytest.mark.parametrize("k", (4, 16, 32, 64, 100, 128, 256, 600))
def test_sweep_widths_weighted(gpu, oracle_mod, k):
    rowptr, col = cases.random_csr(300, 257, 9.0, seed=10 + k, empty_rows=(0, 150, 299))
    val = cases.weights(col.size, 4)
    x = cases.dense(257, k, 3)
    _sweep_all(gpu, oracle_mod, rowptr, col, val, x)


@pytest.mark.parametrize("k", (16, 64, 128))
def test_sweep_unit_weights(gpu, oracle_mod, k):
    rowptr, col = cases.random_csr(200, 200, 12.0, seed=77)
    val = cases.weights(col.size, 0, "unit")
    x = cases.dense(200, k, 3)
    _sweep_all(gpu, oracle_mod, rowptr, col, val, x, unit=True)


@pytest.mark.parametrize("kind", ("integer", "constant", "signed_zero", "nonfinite", "denormal"))
def test_sweep_ties_and_nonfinite(gpu, oracle_mod, kind):
    rowptr, col = cases.random_csr(128, 96, 20.0, seed=5, empty_rows=(3,), duplicates=True)
    val = cases.weights(col.size, 4, "signed_int" if kind != "constant" else "unit")
    x = cases.dense(96, 64, 3, kind)
    _sweep_all(gpu, oracle_mod, rowptr, col, val, x)


@pytest.mark.parametrize("k", (32, 128))
def test_sweep_hub_row_is_cut_into_virtual_rows(gpu, oracle_mod, k):
    """A row of 12,345 edges among short ones: with chunk = 64 / 100 it becomes >100 virtual rows on different waves,
    folded by sweep_hub_fold_kernel; integer X makes every max/min a tie."""
    rowptr, col = cases.random_csr(64, 400, 6.0, seed=9, empty_rows=(0, 63), hub=(17, 12345), duplicates=True)
    val = cases.weights(col.size, 4, "signed_int")
    x = cases.dense(400, k, 3, "integer")
    _sweep_all(gpu, oracle_mod, rowptr, col, val, x)


def test_sweep_rectangular_and_strided(gpu, oracle_mod):
    """M != N (the A^T call of the backward) and leading dimensions larger than k."""
    from isplib_amd import cabi
    from isplib_amd.plan import build_sweep_plan
    rowptr, col = cases.random_csr(90, 333, 25.0, seed=3)
    val = cases.weights(col.size, 4)
    k, ld = 48, 64
    xfull = cases.dense(333, ld, 3)
    x = np.ascontiguousarray(xfull[:, :k])
    d_rowptr, d_col, d_val = _t(rowptr, gpu), _t(col, gpu), _t(val, gpu)
    d_x = _t(xfull, gpu)[:, :k]
    plan = build_sweep_plan(d_rowptr, d_col, 333, 4, 16, 16, 256, 8)
    zfull = torch.full((90, ld), 7.0, device=gpu)
    z = zfull[:, :k]
    cabi.fusedMM_csr_sweep_hip(cabi.MSG_SPMM_SUM, d_rowptr, d_col, d_val, plan, d_x, z, None, plan.workspace("sum", k))
    torch.cuda.synchronize()
    _check(oracle_mod, rowptr, col, val, x, "sum", z.contiguous(), None)
    assert bool((zfull[:, k:] == 7.0).all()), "columns beyond k must not be touched"


def test_sweep_epilogue_and_mean(gpu, oracle_mod):
    from isplib_amd import cabi
    from isplib_amd.plan import build_sweep_plan
    rowptr, col = cases.random_csr(150, 150, 15.0, seed=21, empty_rows=(4,), hub=(9, 700))
    k = 64
    x = cases.dense(150, k, 3)
    ones = np.ones(col.size, np.float32)
    d_rowptr, d_col, d_x = _t(rowptr, gpu), _t(col, gpu), _t(x, gpu)
    plan = build_sweep_plan(d_rowptr, d_col, 150, 4, 8, 16, 128, 8)
    rs = cases.dense(150, 1, 8)[:, 0].copy()
    bias = cases.dense(1, k, 9)[0].copy()
    out, _ = cabi.spmm_sweep(d_rowptr, d_col, None, plan, d_x, "sum", row_scale=_t(rs, gpu), self_term=d_x, bias=_t(bias, gpu), relu=True)
    ref, _ = oracle_mod.spmm_fw(rowptr, col, ones, x, "sum")
    want = np.maximum(rs[:, None] * (ref + x) + bias[None, :], 0.0)
    tol = cases.sum_tolerance(oracle_mod, rowptr, col, ones, x) * np.abs(rs[:, None]) + 1e-6
    assert np.all(np.abs(out.cpu().numpy() - want) <= tol)


def test_sweep_status_codes(gpu):
    from isplib_amd import cabi
    from isplib_amd.plan import build_sweep_plan
    rowptr, col = cases.random_csr(40, 40, 5.0, seed=1)
    d_rowptr, d_col = _t(rowptr, gpu), _t(col, gpu)
    plan = build_sweep_plan(d_rowptr, d_col, 40, 2, 4, 16)
    x = torch.zeros((40, 6), device=gpu)          # k % 4 != 0
    z = torch.zeros((40, 6), device=gpu)
    assert cabi.fusedMM_csr_sweep_hip(cabi.MSG_SPMM_SUM, d_rowptr, d_col, None, plan, x, z, None, check=False) == 1
    x = torch.zeros((40, 8), device=gpu)
    z = torch.zeros((41, 8), device=gpu)
    bad = torch.cat([d_rowptr, d_rowptr[-1:]])    # another row count than the plan's
    assert cabi.fusedMM_csr_sweep_hip(cabi.MSG_SPMM_SUM, bad, d_col, None, plan, x, z, None, check=False) == 1
    assert cabi.fusedMM_csr_sweep_hip(0x11101, d_rowptr, d_col, None, plan, x, z[:40], None, check=False) == 128


# ---- stream form (fusedMM_csr_stream_hip): sum / mean, the plan owns the edges in walking order ---------------------

def _stream_all(gpu, oracle, rowptr, col, val, x, unit=False, geoms=((8, 16, 4, 64), (5, 6, 8, 2048), (16, 7, 2, 100), (3, 3, 4, 300))):
    """geoms: (slices, waves_per_gen, streams, chunk); rows per wave are the kernel's (isplib_spmm_stream_geometry)"""
    from isplib_amd import cabi
    from isplib_amd.plan import build_stream_plan
    d_rowptr, d_col, d_x = _t(rowptr, gpu), _t(col, gpu), _t(x, gpu)
    d_val = None if unit else _t(val, gpu)
    for (s, wpg, streams, chunk) in geoms:
        plan = build_stream_plan(d_rowptr, d_col, d_val, x.shape[0], s, wpg, None, streams, chunk)
        assert plan is not None and int((plan.perm >= 0).sum()) == col.size
        for red in ("sum", "mean"):
            out = cabi.spmm_stream(d_rowptr, col.size, plan, d_x, red)
            again = cabi.spmm_stream(d_rowptr, col.size, plan, d_x, red)
            torch.cuda.synchronize()
            assert torch.equal(out.view(torch.int32), again.view(torch.int32)), "stream schedule must be bitwise reproducible"
            _check(oracle, rowptr, col, val, x, red, out, None)


@pytest.mark.parametrize("k", (4, 5, 16, 32, 41, 64, 67, 100, 101, 128, 256, 602))
def test_stream_widths_weighted(gpu, oracle_mod, k):
    rowptr, col = cases.random_csr(300, 257, 9.0, seed=10 + k, empty_rows=(0, 150, 299))
    val = cases.weights(col.size, 4)
    x = cases.dense(257, k, 3)
    _stream_all(gpu, oracle_mod, rowptr, col, val, x)


@pytest.mark.parametrize("k", (16, 64, 128))
def test_stream_unit_weights(gpu, oracle_mod, k):
    rowptr, col = cases.random_csr(200, 200, 12.0, seed=77)
    val = cases.weights(col.size, 0, "unit")
    x = cases.dense(200, k, 3)
    _stream_all(gpu, oracle_mod, rowptr, col, val, x, unit=True)


@pytest.mark.parametrize("kind", ("integer", "signed_zero", "nonfinite", "denormal"))
def test_stream_special_values(gpu, oracle_mod, kind):
    """Integer operands make the sums exact (any order must give the oracle's bits); NaN / Inf propagate through the
    LDS adds; subnormal products and sums are not flushed."""
    from isplib_amd import cabi
    from isplib_amd.plan import build_stream_plan
    rowptr, col = cases.random_csr(128, 96, 20.0, seed=5, empty_rows=(3,), duplicates=True)
    val = cases.weights(col.size, 4, "signed_int")
    x = cases.dense(96, 64, 3, kind)
    _stream_all(gpu, oracle_mod, rowptr, col, val, x)
    if kind in ("integer", "signed_zero", "denormal"):
        plan = build_stream_plan(_t(rowptr, gpu), _t(col, gpu), _t(val, gpu), 96, 4, 16, None, 4, 64)
        out = cabi.spmm_stream(_t(rowptr, gpu), col.size, plan, _t(x, gpu), "sum").cpu().numpy()
        ref, _ = oracle_mod.spmm_fw(rowptr, col, val, x, "sum")
        if kind == "denormal":
            assert np.any((np.abs(ref) > 0) & (np.abs(ref) < np.finfo(np.float32).tiny)), "case must produce subnormal sums"
            assert np.all(np.abs(out.astype(np.float64) - ref.astype(np.float64)) <= 1e-5 * np.abs(ref) + 1e-44 * 64)
        else:
            assert np.array_equal(out, ref)


@pytest.mark.parametrize("k", (32, 41, 128, 130))
def test_stream_hub_row_is_cut_into_virtual_rows(gpu, oracle_mod, k):
    rowptr, col = cases.random_csr(64, 400, 6.0, seed=9, empty_rows=(0, 63), hub=(17, 12345), duplicates=True)
    val = cases.weights(col.size, 4, "signed_int")
    x = cases.dense(400, k, 3, "integer")
    _stream_all(gpu, oracle_mod, rowptr, col, val, x)


def test_stream_rectangular_strided_epilogue_and_new_weights(gpu, oracle_mod):
    from isplib_amd import cabi
    from isplib_amd.plan import build_stream_plan
    rowptr, col = cases.random_csr(90, 333, 25.0, seed=3, hub=(5, 900))
    val = cases.weights(col.size, 4)
    k, ld = 48, 64
    xfull = cases.dense(333, ld, 3)
    x = np.ascontiguousarray(xfull[:, :k])
    d_rowptr, d_col = _t(rowptr, gpu), _t(col, gpu)
    d_x = _t(xfull, gpu)[:, :k]
    plan = build_stream_plan(d_rowptr, d_col, _t(val, gpu), 333, 4, 16, None, 4, 256)
    zfull = torch.full((90, ld), 7.0, device=gpu)
    z = zfull[:, :k]
    cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, d_rowptr, col.size, plan, d_x, z, plan.workspace())
    torch.cuda.synchronize()
    _check(oracle_mod, rowptr, col, val, x, "sum", z.contiguous(), None)
    assert bool((zfull[:, k:] == 7.0).all()), "columns beyond k must not be touched"
    # weights replaced: one gather through the plan's permutation, same plan
    val2 = cases.weights(col.size, 99)
    plan.set_values(_t(val2, gpu))
    out = cabi.spmm_stream(d_rowptr, col.size, plan, _t(x, gpu), "mean")
    _check(oracle_mod, rowptr, col, val2, x, "mean", out, None)
    # epilogue on a square unit-weight graph
    rowptr, col = cases.random_csr(150, 150, 15.0, seed=21, empty_rows=(4,), hub=(9, 700))
    x = cases.dense(150, 64, 3)
    ones = np.ones(col.size, np.float32)
    d_rowptr, d_col, d_x = _t(rowptr, gpu), _t(col, gpu), _t(x, gpu)
    plan = build_stream_plan(d_rowptr, d_col, None, 150, 4, 8, None, 4, 128)
    rs = cases.dense(150, 1, 8)[:, 0].copy()
    bias = cases.dense(1, 64, 9)[0].copy()
    out = cabi.spmm_stream(d_rowptr, col.size, plan, d_x, "sum", row_scale=_t(rs, gpu), self_term=d_x, bias=_t(bias, gpu), relu=True)
    ref, _ = oracle_mod.spmm_fw(rowptr, col, ones, x, "sum")
    want = np.maximum(rs[:, None] * (ref + x) + bias[None, :], 0.0)
    tol = cases.sum_tolerance(oracle_mod, rowptr, col, ones, x) * np.abs(rs[:, None]) + 1e-6
    assert np.all(np.abs(out.cpu().numpy() - want) <= tol)


def test_stream_status_codes(gpu):
    from isplib_amd import cabi
    from isplib_amd.plan import build_stream_plan
    rowptr, col = cases.random_csr(40, 40, 5.0, seed=1)
    d_rowptr, d_col = _t(rowptr, gpu), _t(col, gpu)
    plan = build_stream_plan(d_rowptr, d_col, None, 40, 2, 4, None, 4)
    x, z = torch.zeros((40, 8), device=gpu), torch.zeros((40, 8), device=gpu)
    assert cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_MAX, d_rowptr, col.size, plan, x, z, check=False) == 128      # sum / mean only
    x3, z3 = torch.zeros((40, 3), device=gpu), torch.zeros((40, 3), device=gpu)
    assert cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, d_rowptr, col.size, plan, x3, z3, check=False) == 1       # k < 4
    x41 = torch.zeros((41, 8), device=gpu)
    assert cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, d_rowptr, col.size, plan, x41, z, check=False) == 1       # other n than the plan's
    assert cabi.fusedMM_csr_stream_hip(cabi.MSG_SPMM_SUM, d_rowptr, col.size, plan, x, z, check=False) == 0


@pytest.mark.parametrize("geom", ((4, 8, 64, 16), (8, 5, 2048, 7), (2, 16, 100, 9), (4, 3, 300, 3)))
def test_native_stream_plan_equals_the_torch_built_one(gpu, oracle_mod, geom):
    """isplib_stream_plan_build_hip (the torch-free host's builder: rocPRIM sorts + HIP kernels) against
    isplib_amd/plan.py's construction, which the CPU tests replay edge by edge: every array identical, and the native
    plan drives the kernel to the oracle's answer (weights re-gathered through its own permutation)."""
    from isplib_amd import cabi
    from isplib_amd.plan import build_stream_plan
    streams, slices, chunk, wpg = geom
    rowptr, col = cases.random_csr(700, 500, 40.0, seed=31, empty_rows=(0, 350, 699), hub=(11, 4000))
    val = cases.weights(col.size, 4)
    x = cases.dense(500, 64, 3)
    d_rowptr, d_col, d_val, d_x = _t(rowptr, gpu), _t(col, gpu), _t(val, gpu), _t(x, gpu)
    ref_plan = build_stream_plan(d_rowptr, d_col, d_val, 500, slices, wpg, None, streams, chunk)
    nat = cabi.NativeStreamPlan(d_rowptr, d_col, d_val, 500, streams, slices, chunk, wpg)
    try:
        for name in ("gens", "waves_per_gen", "rows_per_wave", "streams", "n_steps", "n_parts", "n_hub", "slices"):
            assert getattr(nat, name) == getattr(ref_plan, name), name
        for name in ("words", "perm", "vals", "wave_step_off", "wave_row", "wave_part", "hub_row", "hub_off"):
            want = getattr(ref_plan, name)
            got = nat.array(name)
            assert torch.equal(got.to(want.dtype), want), name
        for red in ("sum", "mean"):
            out = cabi.spmm_stream(d_rowptr, col.size, nat, d_x, red)
            _check(oracle_mod, rowptr, col, val, x, red, out, None)
        val2 = cases.weights(col.size, 77)
        nat.set_values(_t(val2, gpu))
        _check(oracle_mod, rowptr, col, val2, x, "sum", cabi.spmm_stream(d_rowptr, col.size, nat, d_x, "sum"), None)
        nat.set_values(None)
        _check(oracle_mod, rowptr, col, np.ones_like(val), x, "sum", cabi.spmm_stream(d_rowptr, col.size, nat, d_x, "sum"), None)
    finally:
        nat.close()


@pytest.mark.gpu
@pytest.mark.parametrize("streams", (4, 8))
def test_minmax_plans_pad_with_the_spare_row_in_both_builders(gpu, streams):
    """Plans of the max / min kernel: padding words carry the local row `rows_per_wave` -- the kernel's spare LDS row, so its
    loop needs no test for padding -- from the torch builder and the native one alike (identical arrays); sum / mean plans
    keep the first row of the word's own stream."""
    from isplib_amd import cabi
    from isplib_amd.plan import build_stream_plan
    rowptr, col = cases.random_csr(600, 500, 30.0, seed=5, empty_rows=(0, 300), hub=(7, 3000))
    d_rowptr, d_col = _t(rowptr, gpu), _t(col, gpu)
    ref = build_stream_plan(d_rowptr, d_col, None, 500, 5, 6, None, streams, 200, minmax=True)
    nat = cabi.NativeStreamPlan(d_rowptr, d_col, None, 500, streams, 5, 200, 6, minmax=True)
    try:
        rpw = cabi.stream_minmax_geometry(streams)[0]
        assert ref.rows_per_wave == rpw == nat.rows_per_wave
        for name in ("words", "perm", "wave_step_off", "wave_row", "wave_part", "hub_row", "hub_off"):
            assert torch.equal(nat.array(name).to(getattr(ref, name).dtype), getattr(ref, name)), name
        words, perm = ref.words.cpu().numpy().view(np.uint32), ref.perm.cpu().numpy()
        pad = perm < 0
        assert pad.any() and np.all(words[pad] == ((rpw << 24) | 500)) and np.all((words[~pad] >> 24) < rpw)
        plain = build_stream_plan(d_rowptr, d_col, None, 500, 5, 6, None, streams, 200)
        w2, p2 = plain.words.cpu().numpy().view(np.uint32), plain.perm.cpu().numpy()
        per = plain.rows_per_wave // streams
        assert np.all((w2[p2 < 0] >> 24) == (np.flatnonzero(p2 < 0) % streams) * per)
    finally:
        nat.close()


@pytest.mark.gpu
def test_native_plan_arrays_live_as_long_as_a_tensor_looks_at_them(gpu):
    """The plug-in's stream plans are zero-copy views of arrays the C library owns (plan.build_stream_plan_native), and
    autograd saves those tensors for backward: the owner must stay alive for as long as ANY such tensor does -- a graph
    object built inside forward() and dropped before backward() must not take the arrays with it."""
    import gc
    import weakref
    from isplib_amd.plan import build_stream_plan_native
    rowptr, col = cases.random_csr(300, 300, 20.0, seed=9)
    plan = build_stream_plan_native(_t(rowptr, gpu), _t(col, gpu), 300, 4, 4, 256)
    owner = weakref.ref(plan._native)
    words, copy = plan.words, plan.words.clone()
    saved = [words.view(-1)[: words.numel()]]           # what ctx.saved_data holds: another tensor on the same storage
    del plan, words
    gc.collect()
    assert owner() is not None, "the library's arrays were freed while a tensor still looks at them"
    assert torch.equal(saved[0], copy)
    del saved
    gc.collect()
    assert owner() is None, "nothing looks at the arrays any more: the owner goes, and frees them"


# ---- max / min on the stream schedule (fusedMM_csr_stream_minmax_hip): (value, CSR position) pairs --------------------

def _stream_minmax_all(gpu, oracle, rowptr, col, val, x, unit=False, geoms=((4, 8, 16, 64), (8, 5, 6, 2048), (4, 3, 3, 300), (8, 2, 4, 100)),
                       native=False):
    """geoms: (streams, slices, waves_per_gen, chunk); rows per wave are the kernel's (isplib_spmm_stream_minmax_geometry)"""
    from isplib_amd import cabi
    from isplib_amd.plan import build_stream_plan
    d_rowptr, d_col, d_x = _t(rowptr, gpu), _t(col, gpu), _t(x, gpu)
    d_val = None if unit else _t(val, gpu)
    for (streams, s, wpg, chunk) in geoms:
        if native:
            plan = cabi.NativeStreamPlan(d_rowptr, d_col, d_val, x.shape[0], streams, s, chunk, wpg, minmax=True)
        else:
            plan = build_stream_plan(d_rowptr, d_col, d_val, x.shape[0], s, wpg, None, streams, chunk, minmax=True)
        assert plan.streams == streams and plan.rows_per_wave == cabi.stream_minmax_geometry(streams)[0]
        for red in ("max", "min"):
            out, arg = cabi.spmm_stream_minmax(d_rowptr, col.size, plan, d_x, red)
            torch.cuda.synchronize()
            _check(oracle, rowptr, col, val, x, red, out, arg)
            out2, none = cabi.spmm_stream_minmax(d_rowptr, col.size, plan, d_x, red, want_arg=False)
            assert none is None and torch.equal(out.view(torch.int32), out2.view(torch.int32))
        if native:
            plan.close()


@pytest.mark.parametrize("k", (4, 5, 16, 41, 64, 67, 100, 128, 256, 602))
def test_stream_minmax_widths_weighted(gpu, oracle_mod, k):
    rowptr, col = cases.random_csr(300, 257, 9.0, seed=10 + k, empty_rows=(0, 150, 299))
    val = cases.weights(col.size, 4)
    x = cases.dense(257, k, 3)
    _stream_minmax_all(gpu, oracle_mod, rowptr, col, val, x)


@pytest.mark.parametrize("kind", ("integer", "constant", "signed_zero", "nonfinite", "denormal"))
def test_stream_minmax_ties_and_nonfinite(gpu, oracle_mod, kind):
    """Ties must go to the lowest CSR position: the stream meets a column-sorted row's edges slice by slice, which is
    its CSR order, and only a strictly better candidate replaces the one held (duplicates=True: equal neighbours inside
    rows; integer / constant operands: ties everywhere); NaN candidates never win, as in the oracle."""
    rowptr, col = cases.random_csr(128, 96, 20.0, seed=5, empty_rows=(3,), duplicates=True)
    val = cases.weights(col.size, 4, "signed_int" if kind != "constant" else "unit")
    x = cases.dense(96, 64, 3, kind)
    _stream_minmax_all(gpu, oracle_mod, rowptr, col, val, x, unit=(kind == "constant"))


@pytest.mark.parametrize("k", (32, 41, 128))
def test_stream_minmax_hub_row(gpu, oracle_mod, k):
    """A row of 12,345 edges cut into virtual rows on different waves: partial (value, position) pairs, folded by
    sweep_hub_fold_kernel; integer X makes every max / min a tie across chunks."""
    rowptr, col = cases.random_csr(64, 400, 6.0, seed=9, empty_rows=(0, 63), hub=(17, 12345), duplicates=True)
    val = cases.weights(col.size, 4, "signed_int")
    x = cases.dense(400, k, 3, "integer")
    _stream_minmax_all(gpu, oracle_mod, rowptr, col, val, x)
    _stream_minmax_all(gpu, oracle_mod, rowptr, col, val, x, native=True, geoms=((4, 8, 16, 64), (8, 3, 5, 200)))


def test_stream_minmax_values_only_hub_fold_ignores_the_unwritten_index_plane(gpu, oracle_mod):
    """Round-4 advisor: the values-only launch (z_arg = NULL) never writes the partial rows' index plane, yet the hub fold
    read it to break ties -- the sign of a +0 / -0 tie between chunks of a hub row then depended on stale workspace memory.
    The fold of a values-only launch now breaks ties by chunk order.  Whatever the index plane holds (zeros, ones, the
    positions of an earlier launch), the values must be the with-positions launch's, bit for bit."""
    from isplib_amd import cabi
    from isplib_amd.plan import build_stream_plan
    rowptr, col = cases.random_csr(64, 400, 6.0, seed=9, empty_rows=(0, 63), hub=(17, 12345), duplicates=True)
    x = cases.dense(400, 64, 3, "signed_zero")
    d_rowptr, d_col, d_x = _t(rowptr, gpu), _t(col, gpu), _t(x, gpu)
    for (streams, s, wpg, chunk) in ((4, 8, 16, 64), (8, 3, 5, 200)):
        plan = build_stream_plan(d_rowptr, d_col, None, 400, s, wpg, None, streams, chunk, minmax=True)
        assert plan.n_parts > 0 and plan.n_hub > 0
        for red in ("max", "min"):
            want, arg = cabi.spmm_stream_minmax(d_rowptr, col.size, plan, d_x, red)
            _check(oracle_mod, rowptr, col, np.ones(col.size, np.float32), x, red, want, arg)
            for fill in (0x00, 0xFF, 0x7F):
                ws = plan.workspace(minmax=True)
                ws.view(torch.uint8).fill_(fill)
                got, _ = cabi.spmm_stream_minmax(d_rowptr, col.size, plan, d_x, red, workspace=ws, want_arg=False)
                assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (streams, red, fill)


def test_stream_minmax_refuses_unsorted_rows(gpu):
    """The tie rule needs rows whose columns ascend: both plan builders decline anything else (-> task list)."""
    from isplib_amd import cabi
    from isplib_amd.plan import build_stream_plan
    rowptr, col = cases.random_csr(60, 50, 8.0, seed=3, sort_cols=False)
    d_rowptr, d_col = _t(rowptr, gpu), _t(col, gpu)
    assert build_stream_plan(d_rowptr, d_col, None, 50, 2, 4, None, 4, 64, minmax=True) is None
    assert build_stream_plan(d_rowptr, d_col, None, 50, 2, 4, None, 4, 64) is not None          # sum / mean do not care
    with pytest.raises(RuntimeError, match="column-sorted"):
        cabi.NativeStreamPlan(d_rowptr, d_col, None, 50, 4, 2, 64, 4, minmax=True)


def test_stream_minmax_status_codes(gpu):
    from isplib_amd import cabi
    from isplib_amd.plan import build_stream_plan
    rowptr, col = cases.random_csr(40, 40, 5.0, seed=1)
    d_rowptr, d_col = _t(rowptr, gpu), _t(col, gpu)
    sum_plan = build_stream_plan(d_rowptr, d_col, None, 40, 2, 4, None, 4)
    mm_plan = build_stream_plan(d_rowptr, d_col, None, 40, 2, 4, None, 4, minmax=True)
    x, z = torch.zeros((40, 8), device=gpu), torch.zeros((40, 8), device=gpu)
    assert cabi.fusedMM_csr_stream_minmax_hip(cabi.MSG_SPMM_SUM, d_rowptr, col.size, mm_plan, x, z, check=False) == 128   # max / min only
    if sum_plan.rows_per_wave != mm_plan.rows_per_wave:                                                                    # a sum plan is not a max plan
        assert cabi.fusedMM_csr_stream_minmax_hip(cabi.MSG_SPMM_MAX, d_rowptr, col.size, sum_plan, x, z, check=False) == 1
    assert cabi.fusedMM_csr_stream_minmax_hip(cabi.MSG_SPMM_MAX, d_rowptr, col.size, mm_plan, x, z, check=False) == 0


# ---- SDDMM over the stream plan (isplib_sddmm_stream_hip): the dA of sum / mean on the SpMM's front end ---------------

@pytest.mark.parametrize("k", (4, 5, 16, 32, 41, 64, 66, 67, 100, 128, 130, 256, 300))
def test_sddmm_over_stream_plan(gpu, oracle_mod, k):
    """dval[e] = <y[col[e]], g[row(e)]> (/ max(deg, 1) for mean; csrc/fusedmm.cpp:270,351) over the stream plan of the SpMM,
    from both plan builders: every slot width (8, 16, 32 lanes), one and several column panels, a sliver panel of 1-3
    columns (k = 66, 67, 130), ragged k, empty rows, a hub row cut into virtual rows, rectangular; twice for bitwise
    reproducibility; integer operands make the dot products exact."""
    from isplib_amd import cabi
    from isplib_amd.plan import build_stream_plan
    rowptr, col = cases.random_csr(200, 700, 25.0, seed=k, empty_rows=(0, 199), hub=(9, 6000))
    x, g = cases.dense(700, k, 3), cases.dense(200, k, 5)
    xi, gi = cases.dense(700, k, 3, "integer"), cases.dense(200, k, 5, "integer")
    d = [_t(a, gpu) for a in (rowptr, col, x, g, xi, gi)]
    for (s, wpg, streams, chunk) in ((8, 16, 4, 64), (5, 6, 8, 2048), (16, 7, 2, 100), (3, 3, 4, 300)):
        plan = build_stream_plan(d[0], d[1], None, 700, s, wpg, None, streams, chunk)
        nat = cabi.NativeStreamPlan(d[0], d[1], None, 700, streams, s, chunk, wpg)
        for mean in (False, True):
            got = cabi.sddmm_stream(d[0], col.size, plan, d[2], d[3], mean)
            again = cabi.sddmm_stream(d[0], col.size, nat, d[2], d[3], mean)
            assert torch.equal(got.view(torch.int32), again.view(torch.int32)), "bitwise reproducible, whichever builder made the plan"
            ref = oracle_mod.sddmm(rowptr, col, x, g, mean=mean)
            bound = 1e-5 * oracle_mod.sddmm(rowptr, col, np.abs(x), np.abs(g), mean=mean) + 1e-30
            assert np.all(np.abs(got.cpu().numpy() - ref) <= bound), (k, streams, mean)
        exact = cabi.sddmm_stream(d[0], col.size, plan, d[4], d[5], False).cpu().numpy()
        assert np.array_equal(exact, oracle_mod.sddmm(rowptr, col, xi, gi)), (k, streams)
        nat.close()


def test_sddmm_stream_status_codes(gpu):
    import ctypes
    from isplib_amd import cabi
    from isplib_amd.plan import build_stream_plan
    rowptr, col = cases.random_csr(40, 40, 5.0, seed=1)
    d_rowptr, d_col = _t(rowptr, gpu), _t(col, gpu)
    plan = build_stream_plan(d_rowptr, d_col, None, 40, 2, 4, None, 4, 64)
    L = cabi.exp_lib()          # include/isplib_hip_experimental.h
    y, g, dval = torch.zeros((40, 8), device=gpu), torch.zeros((40, 8), device=gpu), torch.zeros(col.size, device=gpu)
    rp = d_rowptr.data_ptr()

    def call(ps, k=8, ldy=8, nnz=col.size):
        return L.isplib_sddmm_stream_hip(40, 40, k, nnz, ctypes.c_void_p(rp), ctypes.c_void_p(rp + 8), ps, ctypes.c_void_p(y.data_ptr()),
                                         ldy, ctypes.c_void_p(g.data_ptr()), 8, 0, ctypes.c_void_p(dval.data_ptr()), None)
    ps = plan.struct()
    assert call(ctypes.byref(ps)) == 0
    assert call(None) == 1 and "plan is required" in cabi.last_error()
    assert call(ctypes.byref(ps), k=3) == 1 and call(ctypes.byref(ps), ldy=4) == 1
    no_perm = plan.struct()
    no_perm.perm = None
    assert call(ctypes.byref(no_perm)) == 1 and "perm" in cabi.last_error()
    other = plan.struct()
    other.rows_per_wave = 8
    assert call(ctypes.byref(other)) == 1


# ---- hybrid form (fusedMM_csr_hybrid_hip): hot rows of y from an LDS table, cold edges through the gather pipeline ------

def _hybrid_geoms(streams):
    from isplib_amd import cabi
    rpw, _, ht, cap = cabi.hybrid_geometry(streams)
    return rpw, ht, cap


@pytest.mark.parametrize("k,streams", ((4, 8), (16, 8), (32, 8), (41, 4), (64, 4), (67, 4), (128, 4), (130, 4), (31, 8), (100, 8)))
def test_hybrid_widths(gpu, oracle_mod, k, streams):
    """sum / mean, unit weights, through the hybrid kernel: slices with more and fewer hot rows than the table holds, chunks
    that hit the hot-step cap (the rest stays cold), a wave with fewer cold batches than slices (padding pairs), empty rows,
    a hub row in virtual rows, ragged k and sliver panels; twice for bitwise reproducibility; integer X: exact."""
    from isplib_amd import cabi
    from isplib_amd.plan import build_hybrid_plan
    rowptr, col = cases.random_csr(900, 1200, 30.0, seed=100 + k, empty_rows=(0, 450, 899), hub=(17, 9000), duplicates=True)
    val = cases.weights(col.size, 0, "unit")
    x = cases.dense(1200, k, 3)
    xi = cases.dense(1200, k, 3, "integer")
    d_rowptr, d_col = _t(rowptr, gpu), _t(col, gpu)
    for slices, wpg, chunk, refs in ((3, 8, 300, 2), (7, 16, 2048, 1), (40, 8, 64, 2)):
        plan = build_hybrid_plan(d_rowptr, d_col, 1200, slices, streams, chunk, waves_per_gen=wpg, min_refs=refs)
        assert plan is not None and plan.hot_edges + int((plan.cold.perm >= 0).sum()) == col.size and plan.hot_edges > 0
        for red in ("sum", "mean"):
            out = cabi.spmm_hybrid(d_rowptr, col.size, plan, _t(x, gpu), red)
            again = cabi.spmm_hybrid(d_rowptr, col.size, plan, _t(x, gpu), red)
            torch.cuda.synchronize()
            assert torch.equal(out.view(torch.int32), again.view(torch.int32)), "hybrid schedule must be bitwise reproducible"
            _check(oracle_mod, rowptr, col, val, x, red, out, None)
        out = cabi.spmm_hybrid(d_rowptr, col.size, plan, _t(xi, gpu), "sum").cpu().numpy()
        ref, _ = oracle_mod.spmm_fw(rowptr, col, val, xi, "sum")
        assert np.array_equal(out, ref), (k, streams, slices)


def test_hybrid_epilogue_and_status_codes(gpu, oracle_mod):
    import ctypes
    from isplib_amd import cabi
    from isplib_amd.plan import build_hybrid_plan
    n, k = 500, 64
    rowptr, col = cases.random_csr(n, n, 20.0, seed=3, empty_rows=(7,))
    d_rowptr, d_col = _t(rowptr, gpu), _t(col, gpu)
    plan = build_hybrid_plan(d_rowptr, d_col, n, 4, 4, 256, waves_per_gen=8)
    x, self_t = cases.dense(n, k, 3), cases.dense(n, k, 8)
    rs, bias = np.abs(cases.dense(n, 1, 4)).ravel() + 0.5, cases.dense(1, k, 6).ravel()
    got = cabi.spmm_hybrid(d_rowptr, col.size, plan, _t(x, gpu), "sum", row_scale=_t(rs, gpu), self_term=_t(self_t, gpu),
                           bias=_t(bias, gpu), relu=True).cpu().numpy()
    ref, _ = oracle_mod.spmm_fw(rowptr, col, np.ones(col.size, np.float32), x, "sum")
    want = np.maximum(rs[:, None] * (ref + self_t) + bias[None, :], 0.0)
    mag, _ = oracle_mod.spmm_fw(rowptr, col, np.ones(col.size, np.float32), np.abs(x), "sum")
    assert np.all(np.abs(got - want) <= 1e-5 * (rs[:, None] * (mag + np.abs(self_t)) + np.abs(bias)[None, :]) + 1e-30)
    L = cabi.exp_lib()          # include/isplib_hip_experimental.h
    y, z = _t(x, gpu), torch.zeros((n, k), device=gpu)
    ws = plan.workspace()
    rp = d_rowptr.data_ptr()

    def call(msg, ps, kk=k):
        return L.fusedMM_csr_hybrid_hip(msg, n, n, kk, col.size, ctypes.c_void_p(rp), ctypes.c_void_p(rp + 8), ps, ctypes.c_void_p(y.data_ptr()),
                                        k, ctypes.c_void_p(z.data_ptr()), k, ctypes.c_void_p(ws.data_ptr()), ws.numel(), None, None)
    ps = plan.struct()
    assert call(cabi.MSG_SPMM_SUM, ctypes.byref(ps)) == 0
    assert call(cabi.MSG_SPMM_MAX, ctypes.byref(ps)) == 128
    assert call(cabi.MSG_SPMM_SUM, None) == 1 and call(cabi.MSG_SPMM_SUM, ctypes.byref(ps), kk=3) == 1
    bad = plan.struct()
    bad.table_rows = 64
    assert call(cabi.MSG_SPMM_SUM, ctypes.byref(bad)) == 1 and "geometry" in cabi.last_error()
    odd = plan.struct()
    odd.cold.waves_per_gen = 12
    assert call(cabi.MSG_SPMM_SUM, ctypes.byref(odd)) == 1


def test_stream_entries_can_be_captured_in_a_hip_graph(gpu, oracle_mod):
    """The planned entries do nothing on the hot path but launch kernels on the caller's stream (no allocation, no
    synchronisation, no host read-back: the workspace and the plan are the caller's), so a training loop can capture them in a
    HIP graph (torch.cuda.CUDAGraph is one on ROCm) and replay it: sum on the stream schedule (two generations, a hub fold),
    max with positions, and the values-only max, replayed on NEW operand values written into the captured buffers, against
    the eager calls bit for bit and against the oracle."""
    from isplib_amd import cabi
    from isplib_amd.plan import build_stream_plan
    rowptr, col = cases.random_csr(700, 600, 25.0, seed=21, empty_rows=(0, 350), hub=(11, 5000))
    val = cases.weights(col.size, 4)
    d_rowptr, d_col, d_val = _t(rowptr, gpu), _t(col, gpu), _t(val, gpu)
    k = 72
    xs = [cases.dense(600, k, 3), cases.dense(600, k, 8)]
    x = _t(xs[0], gpu)
    plan = build_stream_plan(d_rowptr, d_col, d_val, 600, 4, 3, None, 2, 300)
    mplan = build_stream_plan(d_rowptr, d_col, d_val, 600, 3, 3, None, 4, 300, minmax=True)
    assert plan.gens > 1 and plan.n_hub > 0
    ws, mws = plan.workspace(), mplan.workspace(minmax=True)
    z, zm, zv = (torch.empty((700, k), device=gpu) for _ in range(3))
    arg = torch.empty((700, k), dtype=torch.int64, device=gpu)

    def step():
        cabi.fusedMM_csr_stream_hip(cabi.MESSAGE["sum"], d_rowptr, col.size, plan, x, z, ws)
        cabi.fusedMM_csr_stream_minmax_hip(cabi.MESSAGE["max"], d_rowptr, col.size, mplan, x, zm, arg, mws)
        cabi.fusedMM_csr_stream_minmax_hip(cabi.MESSAGE["max"], d_rowptr, col.size, mplan, x, zv, None, mws)

    side = torch.cuda.Stream(device=gpu)
    side.wait_stream(torch.cuda.current_stream(gpu))
    with torch.cuda.stream(side):
        step()                                               # warm-up outside the capture (module load, first-call paths)
    torch.cuda.current_stream(gpu).wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for host_x in (xs[1], xs[0]):
        x.copy_(_t(host_x, gpu))
        z.zero_(); zm.zero_(); zv.zero_(); arg.zero_()
        graph.replay()
        torch.cuda.synchronize()
        got = (z.clone(), zm.clone(), zv.clone(), arg.clone())
        step()
        torch.cuda.synchronize()
        for a_, b_ in zip(got, (z, zm, zv, arg)):
            assert torch.equal(a_.view(torch.int32) if a_.dtype == torch.float32 else a_, b_.view(torch.int32) if b_.dtype == torch.float32 else b_)
        _check(oracle_mod, rowptr, col, val, host_x, "sum", got[0], None)
        _check(oracle_mod, rowptr, col, val, host_x, "max", got[1], got[3])
        assert torch.equal(got[1].view(torch.int32), got[2].view(torch.int32))


def test_long_row_of_repeated_terms_stays_within_the_bound_of_the_oracle_or_of_the_exact_sum(gpu, oracle_mod):
    """A row of 1,269 stored entries over TWO columns with unit weights: every term has the sign of its column's value, so any
    sequential fp32 sum -- the oracle's included -- is ~3x the 1e-5 x sum |a||x| bound away from the exact sum (found by
    scripts/fuzz_parity.py, profiles/r05_fuzz_long.txt).  What parity asks of a schedule on such a row: within the bound of the
    ORACLE (a plan that keeps the row in one piece adds in the oracle's order) or of the exact fp64 sum (a plan that cuts it)."""
    from isplib_amd import cabi
    from isplib_amd.plan import build_stream_plan
    m, n, k = 82, 2, 256
    rowptr, col = cases.random_csr(m, n, 2, seed=1, hub=(26, 1269), duplicates=True)
    x = cases.dense(n, k, 8, "uniform")
    ones = np.ones(col.size, np.float32)
    row_ids = np.repeat(np.arange(m), np.diff(rowptr))
    exact = np.zeros((m, k))
    np.add.at(exact, row_ids, x.astype(np.float64)[col])
    ref, _ = oracle_mod.spmm_fw(rowptr, col, ones, x, "sum")
    tol = cases.sum_tolerance(oracle_mod, rowptr, col, ones, x)
    assert np.max(np.abs(ref - exact) / tol) > 1.5, "the case must put the fp32 oracle itself outside the bound of the exact sum"
    d_rowptr, d_col, d_x = _t(rowptr, gpu), _t(col, gpu), _t(x, gpu)
    outs = {}
    for chunk in (2048, 300, 64):
        plan = build_stream_plan(d_rowptr, d_col, None, n, 5, 6, None, 4, chunk)
        outs[f"stream, chunk {chunk}"] = cabi.spmm_stream(d_rowptr, col.size, plan, d_x, "sum").cpu().numpy()
    plain = torch.empty((m, k), device=gpu)
    cabi.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, d_rowptr, d_col, None, d_x, plain, None)
    outs["plain"] = plain.cpu().numpy()
    for name, o in outs.items():
        near_exact = np.all(np.abs(o - exact) <= tol)
        near_oracle = np.all(np.abs(o - ref) <= tol)
        assert near_exact or near_oracle, (name, float(np.max(np.abs(o - exact) / tol)), float(np.max(np.abs(o - ref) / tol)))
