"""CPU tests of the oracle (no GPU): the C restatement of fusedMM_csr against
(1) the two inputs in the reference tree with derivable answers,
(2) an independent naive NumPy scan,
(3) scipy CSR@dense in fp64 and torch.sparse.mm(csr, X, reduce) on CPU,
(4) the committed golden vectors produced by the reference's own autograd layer
    (tests/golden/make_golden.py), which pin the NumPy restatements of the
    launcher and the four backward formulas.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from tests import cases

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "spmm_ref_layer.npz")


def test_readme_known_answer(oracle_mod):
    rowptr, col, val, x, e_sum, e_max, e_arg = cases.readme_case()
    out, _ = oracle_mod.spmm_fw(rowptr, col, val, x, "sum")
    assert np.array_equal(out, e_sum)
    out, arg = oracle_mod.spmm_fw(rowptr, col, val, x, "max")
    assert np.array_equal(out, e_max) and np.array_equal(arg, e_arg)


def test_gpu_toy_known_answer(oracle_mod):
    rowptr, col, val, x, e = cases.gpu_toy_case()
    out, _ = oracle_mod.spmm_fw(rowptr, col, val, x, "sum")
    assert np.array_equal(out, e)


def test_message_values_match_reference_header(oracle_mod):
    # csrc/fusedMM.h:20,33,47,56,58,67-69 -> the four words of csrc/fusedmm.cpp:168-186
    assert oracle_mod.message("sum") == 0x11102
    assert oracle_mod.message("mean") == 0x13102
    assert oracle_mod.message("max") == 0x21102
    assert oracle_mod.message("min") == 0x31102


def test_unsupported_message_returns_no_opt_impl(oracle_mod):
    rowptr, col, val, x, *_ = cases.readme_case()
    z = np.zeros((3, 3), np.float32)
    assert oracle_mod.fusedMM_csr(0x11103, 3, 3, 3, val, col, rowptr, x, z) == 128   # VOP_ADD
    assert oracle_mod.fusedMM_csr(0x23102, 3, 3, 3, val, col, rowptr, x, z) == 128   # MEAN with MAX


@pytest.mark.parametrize("kind", ("uniform", "integer", "constant", "signed_zero", "nonfinite", "denormal"))
@pytest.mark.parametrize("red", cases.REDUCES)
def test_c_oracle_equals_naive_scan(oracle_mod, red, kind):
    rowptr, col = cases.random_csr(60, 45, 7.0, seed=31, empty_rows=(0, 30, 59), duplicates=True)
    val = cases.weights(col.size, 4, "signed_int" if kind in ("integer", "signed_zero", "nonfinite", "denormal") else "uniform")
    x = cases.dense(45, 19, 3, kind)
    out, arg = oracle_mod.spmm_fw(rowptr, col, val, x, red)
    out2, arg2 = oracle_mod.scan_spmm(rowptr, col, val, x, red)
    if red in ("max", "min"):
        assert np.array_equal(out.view(np.uint32), out2.view(np.uint32))
        assert np.array_equal(arg, arg2)
    elif kind in ("integer", "constant", "signed_zero"):
        assert np.array_equal(out, out2)            # exactly representable sums
    else:
        fin = np.isfinite(out2)
        assert np.array_equal(np.isnan(out), np.isnan(out2))
        tol = cases.sum_tolerance(oracle_mod, rowptr, col, val, np.where(np.isfinite(x), x, 0).astype(np.float32))
        assert np.all(np.abs(out[fin] - out2[fin]) <= tol[fin] + 1e-12)


@pytest.mark.parametrize("k", (1, 16, 41, 128))
def test_sum_mean_against_scipy_fp64(oracle_mod, k):
    rowptr, col = cases.random_csr(200, 150, 12.0, seed=k, empty_rows=(7,))
    val = cases.weights(col.size, 4)
    x = cases.dense(150, k, 3)
    a = sp.csr_matrix((val.astype(np.float64), col, rowptr), shape=(200, 150))
    ref = a @ x.astype(np.float64)
    tol = cases.sum_tolerance(oracle_mod, rowptr, col, val, x)
    out, _ = oracle_mod.spmm_fw(rowptr, col, val, x, "sum")
    assert np.all(np.abs(out - ref) <= tol)
    deg = np.maximum(np.diff(rowptr), 1)[:, None]
    out, _ = oracle_mod.spmm_fw(rowptr, col, val, x, "mean")
    assert np.all(np.abs(out - ref / deg) <= tol / deg + 1e-12)


@pytest.mark.parametrize("red,tred", (("sum", "sum"), ("mean", "mean"), ("max", "amax"), ("min", "amin")))
def test_against_torch_sparse_mm_cpu(oracle_mod, red, tred):
    # no duplicates / no empty rows: torch's CSR reduce defines neither
    rowptr, col = cases.random_csr(50, 80, 6.0, seed=9)
    keep = np.ones(col.size, bool)
    for r in range(50):
        b, e = rowptr[r], rowptr[r + 1]
        _, first = np.unique(col[b:e], return_index=True)
        m = np.zeros(e - b, bool)
        m[first] = True
        keep[b:e] = m
    counts = np.array([keep[rowptr[r]:rowptr[r + 1]].sum() for r in range(50)])
    col, rowptr = col[keep], np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    if np.any(counts == 0):
        pytest.skip("empty row after dedup")
    val = cases.weights(col.size, 4)
    x = cases.dense(80, 12, 3)
    a = torch.sparse_csr_tensor(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val), size=(50, 80))
    ref = torch.sparse.mm(a, torch.from_numpy(x), tred).numpy()
    out, _ = oracle_mod.spmm_fw(rowptr, col, val, x, red)
    if red in ("max", "min"):
        assert np.array_equal(out, ref)
    else:
        assert np.allclose(out, ref, rtol=1e-5, atol=1e-6)


def test_empty_row_and_sentinel_conventions(oracle_mod):
    rowptr = np.array([0, 0, 2, 2], np.int64)
    col = np.array([1, 0], np.int64)
    val = np.array([1.0, 1.0], np.float32)
    x = np.array([[np.nan, 5.0], [np.nan, 5.0]], np.float32)
    out, arg = oracle_mod.spmm_fw(rowptr, col, val, x, "max")
    assert np.array_equal(out[0], [0, 0]) and np.array_equal(arg[0], [2, 2])          # empty row: 0 / nnz
    assert out[1, 0] == np.finfo(np.float32).min and arg[1, 0] == 2                    # all-NaN: nothing wins
    assert out[1, 1] == 5.0 and arg[1, 1] == 0                                         # tie -> lowest CSR position
    out, _ = oracle_mod.spmm_fw(rowptr, col, val, np.ones((2, 2), np.float32), "mean")
    assert np.array_equal(out, [[0, 0], [1, 1], [0, 0]])                               # divide by max(deg,1)


def test_sddmm_oracle(oracle_mod):
    rowptr, col = cases.random_csr(40, 30, 5.0, seed=3)
    x, g = cases.dense(30, 24, 3), cases.dense(40, 24, 5)
    row = np.repeat(np.arange(40), np.diff(rowptr))
    ref = np.einsum("ek,ek->e", x[col].astype(np.float64), g[row].astype(np.float64))
    assert np.allclose(oracle_mod.sddmm(rowptr, col, x, g), ref, rtol=1e-6, atol=1e-7)
    deg = np.maximum(np.diff(rowptr), 1)[row]
    assert np.allclose(oracle_mod.sddmm(rowptr, col, x, g, mean=True), ref / deg, rtol=1e-6, atol=1e-7)


# ---- golden vectors from the reference's own autograd layer --------------------------------

def _golden():
    assert os.path.exists(GOLDEN), "tests/golden/spmm_ref_layer.npz missing (tests/golden/make_golden.py)"
    z = np.load(GOLDEN)
    names = sorted({k.split("/")[0] for k in z.files})
    return z, names


def test_golden_file_is_complete():
    z, names = _golden()
    assert len(names) >= 7
    for n in names:
        for key in ("rowptr", "col", "val", "x", "g", "sum/out", "sum/dx", "mean/out", "mean/dx",
                    "max/out", "max/arg", "max/dx", "max/dval", "min/out", "min/arg", "min/dx", "min/dval"):
            assert f"{n}/{key}" in z.files


def test_numpy_launcher_matches_reference_layer_forward(oracle_mod):
    z, names = _golden()
    for n in names:
        a = [z[f"{n}/{k}"] for k in ("rowptr", "col", "val", "x")]
        for red in cases.REDUCES:
            out, arg = oracle_mod.spmm_fw(*a, red)
            assert np.array_equal(out.view(np.uint32), z[f"{n}/{red}/out"].view(np.uint32)), (n, red)
            if arg is not None:
                assert np.array_equal(arg, z[f"{n}/{red}/arg"]), (n, red)


def test_numpy_backward_matches_reference_layer(oracle_mod):
    z, names = _golden()
    for n in names:
        rowptr, col, val, x, g = (z[f"{n}/{k}"] for k in ("rowptr", "col", "val", "x", "g"))
        ncols = int(z[f"{n}/ncols"])
        assert np.array_equal(oracle_mod.spmm_sum_bw(rowptr, col, val, ncols, g), z[f"{n}/sum/dx"]), n
        assert np.array_equal(oracle_mod.spmm_mean_bw(rowptr, col, val, ncols, g), z[f"{n}/mean/dx"]), n
        for red in ("max", "min"):
            dval, dx = oracle_mod.spmm_minmax_bw(col, val, x, z[f"{n}/{red}/arg"], g)
            # ATen's CPU scatter_add_ and np.add.at both add in row-major order
            assert np.allclose(dx, z[f"{n}/{red}/dx"], rtol=1e-6, atol=1e-6), (n, red)
            assert np.allclose(dval, z[f"{n}/{red}/dval"], rtol=1e-6, atol=1e-6), (n, red)


def test_dx_of_sum_and_mean_against_dense_fp64(oracle_mod):
    """The intended mean-backward pairing proven against a dense fp64 reference."""
    z, names = _golden()
    for n in names:
        rowptr, col, val, g = (z[f"{n}/{k}"] for k in ("rowptr", "col", "val", "g"))
        ncols = int(z[f"{n}/ncols"])
        m = rowptr.size - 1
        a = sp.csr_matrix((val.astype(np.float64), col, rowptr), shape=(m, ncols)).toarray()
        assert np.allclose(z[f"{n}/sum/dx"], a.T @ g.astype(np.float64), rtol=1e-5, atol=1e-5)
        deg = np.maximum(np.diff(rowptr), 1)[:, None]
        assert np.allclose(z[f"{n}/mean/dx"], (a / deg).T @ g.astype(np.float64), rtol=1e-5, atol=1e-5)


# ---- generic FusedMM pipeline (words the reference never sends; unpinned by it) ----------------------------

def _word(vop, rop, sop, vsc, aop):
    return vop | (rop << 4) | (sop << 8) | (vsc << 12) | (aop << 16)


def test_generic_pipeline_named_patterns_against_closed_forms(oracle_mod):
    rowptr, col = cases.random_csr(50, 40, 6.0, seed=21, empty_rows=(3, 49), duplicates=True)
    k = 12
    x, y = cases.dense(50, k, 7) * 0.5, cases.dense(40, k, 8) * 0.5
    row = np.repeat(np.arange(50), np.diff(rowptr))
    xe, ye = x[row].astype(np.float64), y[col].astype(np.float64)
    # sigmoid embedding: z_i = sum_j sigmoid(<x_i, y_j>) y_j
    st, z, _ = oracle_mod.fusedmm_general(_word(2, 1, 0xF, 1, 1), rowptr, col, None, x, y, sop_udef=1)
    want = np.zeros((50, k))
    np.add.at(want, row, (1 / (1 + np.exp(-(xe * ye).sum(1))))[:, None] * ye)
    assert st == 0 and np.allclose(z, want, rtol=1e-5, atol=1e-6)
    # t-distribution embedding: z_i = sum_j (y_j - x_i) / (1 + |y_j - x_i|^2)
    st, z, _ = oracle_mod.fusedmm_general(_word(5, 5, 0xF, 1, 1), rowptr, col, None, x, y, sop_udef=3)
    d = ye - xe
    want = np.zeros((50, k))
    np.add.at(want, row, d / (1 + (d * d).sum(1))[:, None])
    assert st == 0 and np.allclose(z, want, rtol=1e-5, atol=1e-6)
    # the SpMM word through the generic path = the launcher's SpMM
    val = cases.weights(col.size, 4)
    st, z, _ = oracle_mod.fusedmm_general(0x11102, rowptr, col, val, None, y)
    ref, _ = oracle_mod.spmm_fw(rowptr, col, val, y, "sum")
    assert st == 0 and np.array_equal(z, ref)
    st, z, arg = oracle_mod.fusedmm_general(0x21102, rowptr, col, val, None, y)
    ref, ref_arg = oracle_mod.spmm_fw(rowptr, col, val, y, "max")
    assert st == 0 and np.array_equal(z, ref) and np.array_equal(arg, ref_arg)


def test_generic_pipeline_every_stage_against_numpy_statement(oracle_mod):
    from tests import fusedmm_ref
    rowptr, col = cases.random_csr(30, 25, 5.0, seed=4, empty_rows=(0, 17), duplicates=True)
    val = cases.weights(col.size, 4, "signed_int")
    for k in (5, 16):
        x, y = cases.dense(30, k, 3, "integer"), cases.dense(25, k, 5, "integer")      # exact arithmetic: ties are real
        for vop in range(1, 8):
            for rop in range(0, 6):
                for sop, kind, prm in ((0, 0, 0.0), (1, 0, 0.0), (0xF, 4, 0.25), (0xF, 1, 0.0), (0xF, 3, 0.0)):
                    if kind == 3 and rop not in (4, 5):
                        continue                                  # 1/(1+s) is only meaningful for s >= 0
                    for vsc in range(0, 4):
                        for aop in (1, 2, 3):
                            if vsc == 3 and aop != 1:
                                continue
                            w = _word(vop, rop, sop, vsc, aop)
                            st, z, arg = oracle_mod.fusedmm_general(w, rowptr, col, val, x, y, kind, prm)
                            want, want_arg = fusedmm_ref.fusedmm(w, rowptr, col, val, x, y, kind, prm)
                            assert st == 0, hex(w)
                            exact = kind in (0, 4)
                            if exact and aop != 1:
                                assert np.array_equal(z, want.astype(np.float32)) and np.array_equal(arg, want_arg), hex(w)
                            else:
                                lim = 1e-5 * np.abs(want).max() + 1e-6
                                assert np.all(np.abs(z - want) <= lim), hex(w)


def test_generic_pipeline_status_codes(oracle_mod):
    rowptr, col, val, x, *_ = cases.readme_case()
    call = lambda w, kind=0: oracle_mod.fusedmm_general(w, rowptr, col, val, x, x, kind)[0]  # noqa: E731
    assert call(_word(0xF, 0, 1, 1, 1)) == 64          # VOP_UDEF: no user functions on this path
    assert call(_word(2, 1, 0xF, 1, 1)) == 64          # SOP_UDEF without a menu entry
    assert call(_word(2, 1, 0xF, 1, 1), 1) == 0
    assert call(_word(8, 0, 1, 1, 1)) == 128           # a VOP value the header does not define
    assert call(_word(2, 0, 2, 1, 1)) == 128           # SOP 0x200 undefined
    assert call(_word(2, 0, 1, 3, 2)) == 128           # MEAN with MAX
    assert call(_word(0, 0, 1, 1, 1)) == 128           # VOP_NOOP: nothing to aggregate
    assert call(_word(2, 0, 1, 1, 0)) == 128           # AOP_NOOP


def test_timed_baseline_entry_computes_the_same_rows(oracle_mod):
    """bench.py's cpu_baseline leg (static nnz-balanced row blocks, first-touch placement) runs the oracle's own inner
    loop: bit-identical output, hub row and empty rows included."""
    rowptr, col = cases.random_csr(500, 300, 12.0, seed=8, empty_rows=(0, 499), hub=(17, 4000))
    val = cases.weights(col.size, 4)
    x = cases.dense(300, 24, 3)
    ref, _ = oracle_mod.spmm_fw(rowptr, col, val, x, "sum")
    secs, out = oracle_mod.spmm_sum_timed(rowptr, col, val, x, reps=2)
    assert secs.shape == (2,) and np.all(secs >= 0) and np.array_equal(out, ref)


def test_empty_row_switch_of_the_oracle():
    """The open convention (oracle/fusedmm_oracle.c header): an empty row of max / min is 0 by default and, with
    oracle.set_empty_row("init"), what the reference launcher pre-filled it with (csrc/fusedmm.cpp:147-150); positions
    are the launcher's sentinel nnz either way; every non-empty row and sum / mean are untouched by the switch."""
    import oracle
    rowptr = np.array([0, 0, 2, 2, 3], np.int64)
    col = np.array([1, 2, 0], np.int64)
    val = np.array([2.0, -1.0, 3.0], np.float32)
    x = np.array([[1, -2], [3, 4], [5, -6]], np.float32)
    try:
        for red, fill in (("max", -np.finfo(np.float32).max), ("min", np.finfo(np.float32).max)):
            oracle.set_empty_row("zero")
            z0, a0 = oracle.spmm_fw(rowptr, col, val, x, red)
            oracle.set_empty_row("init")
            z1, a1 = oracle.spmm_fw(rowptr, col, val, x, red)
            assert np.all(z0[[0, 2]] == 0) and np.all(z1[[0, 2]] == np.float32(fill))
            assert np.array_equal(z0[[1, 3]], z1[[1, 3]]) and np.array_equal(a0, a1) and np.all(a1[[0, 2]] == 3)
        oracle.set_empty_row("init")
        for red in ("sum", "mean"):
            assert np.all(oracle.spmm_fw(rowptr, col, val, x, red)[0][[0, 2]] == 0)
    finally:
        oracle.set_empty_row("zero")
