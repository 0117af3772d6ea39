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


@pytest.mark.parametrize("k", (4, 8, 32, 48, 64, 100, 128))
def test_stream_form_of_the_sddmm_fused_words(gpu, oracle_mod, k):
    """fusedMM_csr_udef_stream_hip (round 5: the two graph-embedding words on the stream schedule's front end -- rows of x and
    of z resident in LDS, four steps' dot products per transposed butterfly) over plans of its own geometry, every slot
    width (k <= 32 / 64 / 128), several slices / waves per generation / hub-row chunks (a row of 2,500 entries is cut into
    virtual rows whose partial rows are folded), empty rows, duplicates:
      * with the SCALE menu function on small-integer operands every stage is exact in fp32 -> bit for bit the oracle;
      * the named patterns (sigmoid, 1 - sigmoid, t-distribution, leaky exp) on real operands within 1e-4 of the largest |z|
        (fast exponential, tree-ordered dots), and twice for bitwise reproducibility;
      * against the task-list form of the same word on the same operands: the same bound."""
    from isplib_amd import cabi
    rowptr, col = cases.random_csr(400, 300, 20.0, seed=k, empty_rows=(0, 7, 399), hub=(11, 2500), duplicates=True)
    xi, yi = cases.dense(400, k, 3, "integer"), cases.dense(300, k, 5, "integer")
    sc = np.float32(1.0 / np.sqrt(k))
    xr, yr = cases.dense(400, k, 13) * sc, cases.dense(300, k, 15) * sc
    d_rowptr, d_col = _t(rowptr, gpu), _t(col, gpu)
    streams = 8 if k <= 32 else (4 if k <= 64 else 2)
    rpw, _ = cabi.fusedmm_stream_geometry(streams)
    assert rpw == {2: 16, 4: 32, 8: 64}[streams]
    dot_word, norm_word = cabi.PATTERNS["sigmoid_embedding"][0], cabi.PATTERNS["tdist_embedding"][0]
    for (slices, wpg, chunk) in ((3, 4, 64), (1, 2, 4096), (7, 9, 300)):
        plan = cabi.NativeStreamPlan(d_rowptr, d_col, None, 300, streams, slices, chunk, wpg, fusedmm=True)
        assert plan.rows_per_wave == rpw and plan.streams == streams
        for word in (dot_word, norm_word):
            # exact: s = 0.25 * <x, y> (or 0.25 * |y - x|^2) on small integers, then s * T summed: all representable
            st, ref, _ = oracle_mod.fusedmm_general(word, rowptr, col, None, xi, yi, cabi.SOP_UDEF["scale"], 0.25)
            st2, z = cabi.fusedmm_stream(word, d_rowptr, col.size, plan, _t(xi, gpu), _t(yi, gpu), sop_udef="scale", sop_param=0.25)
            assert st == 0 and st2 == 0
            assert np.array_equal(z.cpu().numpy(), ref), (hex(word), slices)
            for fn in ("sigmoid", "one_minus_sigmoid", "tdist", "leaky_exp"):
                if fn == "tdist" and word == dot_word:
                    continue                                  # 1 / (1 + s) on a signed dot product has a pole
                _, ref, _ = oracle_mod.fusedmm_general(word, rowptr, col, None, xr, yr, cabi.SOP_UDEF[fn], 0.2)
                _, z = cabi.fusedmm_stream(word, d_rowptr, col.size, plan, _t(xr, gpu), _t(yr, gpu), sop_udef=fn, sop_param=0.2)
                _, again = cabi.fusedmm_stream(word, d_rowptr, col.size, plan, _t(xr, gpu), _t(yr, gpu), sop_udef=fn, sop_param=0.2)
                assert torch.equal(z, again), "bitwise reproducible"
                bound = 1e-4 * np.abs(ref).max() + 1e-7
                assert np.all(np.abs(z.cpu().numpy() - ref) <= bound), (hex(word), fn, slices)
                tasks = cabi.fusedmm(word, d_rowptr, d_col, None, _t(xr, gpu), _t(yr, gpu), sop_udef=fn, sop_param=0.2)[1]
                assert np.all(np.abs(z.cpu().numpy() - tasks.cpu().numpy()) <= bound), (hex(word), fn, slices)
        plan.close()


def test_stream_form_status_codes_and_the_plugin_route(gpu, oracle_mod, monkeypatch):
    from isplib_amd import cabi
    import isplib_amd
    rowptr, col = cases.random_csr(90, 80, 9.0, seed=4, empty_rows=(3,))
    d_rowptr, d_col = _t(rowptr, gpu), _t(col, gpu)
    x, y = cases.dense(90, 64, 3), cases.dense(80, 64, 5)
    word = cabi.PATTERNS["sigmoid_embedding"][0]
    plan = cabi.NativeStreamPlan(d_rowptr, d_col, None, 80, 4, 2, 64, 3, fusedmm=True)
    sum_plan = cabi.NativeStreamPlan(d_rowptr, d_col, None, 80, 4, 2, 64, 3)
    call = lambda w=word, p=plan, xx=x, yy=y, fn="sigmoid": cabi.fusedmm_stream(w, d_rowptr, col.size, p, _t(xx, gpu), _t(yy, gpu), sop_udef=fn, check=False)[0]  # noqa: E731
    assert call() == cabi.SUCCESS
    assert call(w=cabi.MSG_SPMM_SUM) == cabi.NO_OPT_IMPL                        # not one of the two words
    assert call(fn="none") == cabi.UNDEFINED_USER_FUNCTION
    assert call(p=sum_plan) == cabi.FAIL and "geometry" in cabi.last_error()     # a sum plan has other rows per wave
    assert call(xx=cases.dense(90, 6, 3), yy=cases.dense(80, 6, 5)) == cabi.FAIL   # k not a multiple of 4
    assert call(xx=cases.dense(90, 68, 3), yy=cases.dense(80, 68, 5)) == cabi.FAIL  # wider than the plan's slots
    assert cabi.suggest_fusedmm_stream(word, 90, 80, col.size, 64) is None        # far too small for the rule
    assert cabi.suggest_fusedmm_stream(word, 232965, 232965, 114615892, 128) is not None
    assert cabi.suggest_fusedmm_stream(word, 232965, 232965, 114615892, 256) is None
    # the plug-in's fusedmm() takes the stream form where the rule accepts the graph (forced here) and gives the oracle's answer
    monkeypatch.setattr(cabi, "suggest_fusedmm_stream", lambda *a_, **k_: (4, 2, 64))
    adj = isplib_amd.SparseTensor.from_csr(d_rowptr, d_col, None, (90, 80))
    z = isplib_amd.fusedmm(adj, _t(x, gpu), _t(y, gpu), "sigmoid_embedding")
    assert adj.storage._fusedmm_streams[(4, 2, 64)] is not None
    _, ref, _ = oracle_mod.fusedmm_general(word, rowptr, col, None, x, y, cabi.SOP_UDEF["sigmoid"], 0.0)
    assert np.all(np.abs(z.cpu().numpy() - ref) <= 1e-4 * np.abs(ref).max() + 1e-7)
    plan.close(); sum_plan.close()
