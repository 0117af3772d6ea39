"""GPU parity of the generic FusedMM pipeline (fusedMM_csr_udef_hip) against the oracle's generic path, which
tests/test_oracle.py pins against an independent NumPy statement.  Integer-valued operands make every stage
exact in fp32, so max/min values AND arg indices (tie rule: lowest CSR position) are compared bit for bit."""
import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu


def _word(vop, rop, sop, vsc, aop):
    return vop | (rop << 4) | (sop << 8) | (vsc << 12) | (aop << 16)


def _t(a, gpu):
    return None if a is None else torch.from_numpy(a).to(gpu)


@pytest.mark.parametrize("k", (5, 16, 41, 300))
def test_every_stage_combination_exact_on_integer_operands(gpu, oracle_mod, k):
    from isplib_amd import cabi
    rowptr, col = cases.random_csr(70, 55, 7.0, seed=k, empty_rows=(0, 33, 69), hub=(5, 300), duplicates=True)
    val = cases.weights(col.size, 4, "signed_int")
    x, y = cases.dense(70, k, 3, "integer"), cases.dense(55, k, 5, "integer")
    d = [_t(a, gpu) for a in (rowptr, col, val, x, y)]
    for vop in range(1, 8):
        for rop in range(0, 6):
            for sop, kind, prm in ((0, 0, 0.0), (1, 0, 0.0), (0xF, 4, 0.25)):
                for vsc in range(0, 4):
                    for aop in (1, 2, 3):
                        if vsc == 3 and aop != 1:
                            continue
                        w = _word(vop, rop, sop, vsc, aop)
                        st, ref, ref_arg = oracle_mod.fusedmm_general(w, rowptr, col, val, x, y, kind, prm)
                        st2, z, arg = cabi.fusedmm(w, *d, sop_udef=kind, sop_param=prm)
                        assert st == 0 and st2 == 0, hex(w)
                        got = z.cpu().numpy()
                        if vsc == 3:       # one division per element on both sides, but products this large round
                            assert np.allclose(got, ref, rtol=1e-6, atol=1e-6), hex(w)
                        else:
                            assert np.array_equal(got, ref), hex(w)
                        if arg is not None:
                            assert np.array_equal(arg.cpu().numpy(), ref_arg), hex(w)


@pytest.mark.parametrize("pattern", ("sigmoid_embedding", "tdist_embedding", "attention_sum"))
@pytest.mark.parametrize("k", (32, 128, 602))
def test_named_patterns_within_tolerance(gpu, oracle_mod, pattern, k):
    """The FusedMM paper's SDDMM-fused patterns on real-valued operands: 1e-4 of the largest |z| (the device uses
    the fast exponential and a tree-ordered dot product; the oracle sums sequentially with expf)."""
    from isplib_amd import cabi
    rowptr, col = cases.random_csr(400, 300, 20.0, seed=1, empty_rows=(7,), hub=(11, 2500))
    scale = 1.0 / np.sqrt(k)
    x, y = cases.dense(400, k, 3) * np.float32(scale), cases.dense(300, k, 5) * np.float32(scale)
    word, fn = cabi.PATTERNS[pattern]
    prm = 0.2
    st, ref, _ = oracle_mod.fusedmm_general(word, rowptr, col, None, x, y, cabi.SOP_UDEF[fn], prm)
    st2, z, _ = cabi.fusedmm(word, _t(rowptr, gpu), _t(col, gpu), None, _t(x, gpu), _t(y, gpu), sop_udef=fn, sop_param=prm)
    assert st == 0 and st2 == 0
    again = cabi.fusedmm(word, _t(rowptr, gpu), _t(col, gpu), None, _t(x, gpu), _t(y, gpu), sop_udef=fn, sop_param=prm)[1]
    assert torch.equal(z, again)                                   # no atomics: bitwise reproducible
    assert np.all(np.abs(z.cpu().numpy() - ref) <= 1e-4 * np.abs(ref).max() + 1e-7)


def test_spmm_words_agree_with_the_tuned_kernels_and_status_codes(gpu, oracle_mod):
    from isplib_amd import cabi
    rowptr, col = cases.random_csr(120, 120, 9.0, seed=2, empty_rows=(0,), duplicates=True)
    val = cases.weights(col.size, 4, "signed_int")
    y = cases.dense(120, 24, 3, "integer")
    d_rowptr, d_col, d_val, d_y = (_t(a, gpu) for a in (rowptr, col, val, y))
    for red in cases.REDUCES:
        fast, fast_arg = cabi.spmm(d_rowptr, d_col, d_val, d_y, red)
        _, z, arg = cabi.fusedmm(cabi.MESSAGE[red], d_rowptr, d_col, d_val, None, d_y)
        assert torch.equal(z, fast) and (arg is None or torch.equal(arg, fast_arg)), red
    bad = lambda w, kind="none": cabi.fusedmm(w, d_rowptr, d_col, d_val, d_y, d_y, sop_udef=kind, check=False)[0]  # noqa: E731
    assert bad(_word(0xF, 0, 1, 1, 1)) == cabi.UNDEFINED_USER_FUNCTION
    assert bad(_word(2, 1, 0xF, 1, 1)) == cabi.UNDEFINED_USER_FUNCTION and "built-in" in cabi.last_error()
    assert bad(_word(2, 1, 0xF, 1, 1), "sigmoid") == cabi.SUCCESS
    assert bad(_word(8, 0, 1, 1, 1)) == cabi.NO_OPT_IMPL
    assert bad(_word(2, 0, 1, 3, 2)) == cabi.NO_OPT_IMPL
    assert cabi.fusedmm(_word(3, 0, 1, 1, 1), d_rowptr, d_col, d_val, None, d_y, check=False)[0] == cabi.FAIL   # VOP_ADD without x


@pytest.mark.parametrize("k", (8, 41, 128, 300))
def test_task_form_of_the_generic_pipeline(gpu, oracle_mod, k):
    """fusedMM_csr_udef_tasks_hip over SpMM task plans: exact on integer operands for a spread of stage words
    (max/min incl. arg; hub rows chunked; odd slice counts), named patterns within tolerance on real operands."""
    from isplib_amd import cabi
    from isplib_amd.plan import build_task_plan
    rowptr, col = cases.random_csr(300, 280, 25.0, seed=k, empty_rows=(0, 150), hub=(9, 300), duplicates=True)   # sums stay < 2^24: exact
    val = cases.weights(col.size, 4, "signed_int")
    x, y = cases.dense(300, k, 3, "integer"), cases.dense(280, k, 5, "integer")
    d = [_t(a, gpu) for a in (rowptr, col, val, x, y)]
    words = [_word(v, r, s, c, a) for v in (1, 2, 4, 5, 6) for r in (0, 1, 3, 5) for s, c in ((0, 1), (1, 1), (0xF, 2), (1, 3))
             for a in (1, 2, 3) if not (c == 3 and a != 1)]
    for slices, chunk, short in ((1, 256, 0), (5, 128, 16), (8, 1024, 128)):
        plan = build_task_plan(d[0], d[1], 280, slices, chunk, short)
        for w in words:
            kind, prm = (4, 0.25) if ((w >> 8) & 0xF) == 0xF else (0, 0.0)
            st, ref, ref_arg = oracle_mod.fusedmm_general(w, rowptr, col, val, x, y, kind, prm)
            st2, z, arg = cabi.fusedmm(w, *d, sop_udef=kind, sop_param=prm, plan=plan)
            assert st == 0 and st2 == 0, hex(w)
            if ((w >> 12) & 0xF) == 3:
                assert np.allclose(z.cpu().numpy(), ref, rtol=1e-6, atol=1e-6), (hex(w), slices)
            else:
                assert np.array_equal(z.cpu().numpy(), ref), (hex(w), slices)
            if arg is not None:
                assert np.array_equal(arg.cpu().numpy(), ref_arg), (hex(w), slices)
    scale = np.float32(1.0 / np.sqrt(k))
    xr, yr = cases.dense(300, k, 13) * scale, cases.dense(280, k, 15) * scale
    plan = build_task_plan(d[0], d[1], 280, 6, 512, 64)
    for pattern in ("sigmoid_embedding", "tdist_embedding", "attention_sum"):
        word, fn = cabi.PATTERNS[pattern]
        _, ref, _ = oracle_mod.fusedmm_general(word, rowptr, col, None, xr, yr, cabi.SOP_UDEF[fn], 0.2)
        z = cabi.fusedmm(word, d[0], d[1], None, _t(xr, gpu), _t(yr, gpu), sop_udef=fn, sop_param=0.2, plan=plan)[1]
        assert np.all(np.abs(z.cpu().numpy() - ref) <= 1e-4 * np.abs(ref).max() + 1e-7), pattern
