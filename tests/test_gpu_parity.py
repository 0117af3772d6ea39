"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs.  Bar: max/min values and arg indices BIT-EXACT; sum/mean
within 1e-5 * sum|val*x| per element (BASELINE.json north_star: 1e-5 relative fp32).
"""
import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _assert_sum_close(got, ref, tol):
    got = np.asarray(got)
    fin = np.isfinite(ref) & np.isfinite(tol)
    assert np.array_equal(np.isnan(got[~fin]), np.isnan(ref[~fin]))
    inf = ~fin & ~np.isnan(ref)
    assert np.array_equal(got[inf], ref[inf])
    err = np.abs(got[fin].astype(np.float64) - ref[fin].astype(np.float64))
    assert np.all(err <= tol[fin]), f"max err/tol = {np.max(err / tol[fin])}"


def _run_all(gpu, oracle, rowptr, col, val, x, unit=False, slices=(8, 16, 1, 5)):
    """Every reduction through both boundary entry points: fusedMM_csr_hip and, when the rows
    are column-sorted, fusedMM_csr_sliced_hip for each slice count."""
    from isplib_amd import cabi
    d_rowptr, d_col, d_x = _t(rowptr, gpu), _t(col, gpu), _t(x, gpu)
    d_val = None if unit else _t(val, gpu)
    tol = cases.sum_tolerance(oracle, rowptr, col, val, x)
    tables = []
    for s in slices:
        table, ok = cabi.spmm_slices(d_rowptr, d_col, x.shape[0], s)
        assert ok, "test graphs are column-sorted"
        tables.append((s, table))
    for red in cases.REDUCES:
        ref, ref_arg = oracle.spmm_fw(rowptr, col, val, x, red)
        results = [("plain", cabi.spmm(d_rowptr, d_col, d_val, d_x, red))]
        results += [(f"sliced{s}", cabi.spmm_sliced(d_rowptr, d_col, d_val, t, s, d_x, red)) for s, t in tables]
        torch.cuda.synchronize()
        for name, (out, arg) in results:
            out = out.cpu().numpy()
            if red in ("sum", "mean"):
                _assert_sum_close(out, ref, tol)
            else:
                assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), f"{red}/{name}: values not bit-exact"
                assert np.array_equal(arg.cpu().numpy(), ref_arg), f"{red}/{name}: arg indices differ"


@pytest.mark.parametrize("k", cases.WIDTHS)
def test_widths_weighted(gpu, oracle_mod, k):
    rowptr, col = cases.random_csr(300, 257, 9.0, seed=10 + k, empty_rows=(0, 150, 299))
    val = cases.weights(col.size, 4)
    x = cases.dense(257, k, 3)
    _run_all(gpu, oracle_mod, rowptr, col, val, x)


@pytest.mark.parametrize("k", (16, 41, 128))
def test_unit_weights_skip_value_stream(gpu, oracle_mod, k):
    rowptr, col = cases.random_csr(200, 200, 12.0, seed=77)
    val = cases.weights(col.size, 0, "unit")
    x = cases.dense(200, k, 3)
    _run_all(gpu, oracle_mod, rowptr, col, val, x, unit=True)


@pytest.mark.parametrize("kind", ("integer", "constant", "signed_zero", "nonfinite", "denormal"))
@pytest.mark.parametrize("k", (64, 100))
def test_ties_and_nonfinite(gpu, oracle_mod, kind, k):
    rowptr, col = cases.random_csr(128, 96, 20.0, seed=5, empty_rows=(3,), duplicates=True)
    val = cases.weights(col.size, 4, "signed_int" if kind != "constant" else "unit")
    x = cases.dense(96, k, 3, kind)
    _run_all(gpu, oracle_mod, rowptr, col, val, x)


@pytest.mark.parametrize("k", (32, 128, 602))
def test_hub_row_uses_workgroup_split(gpu, oracle_mod, k):
    # one row far above the long-row threshold (2048) + degree-1 and empty rows around it
    rowptr, col = cases.random_csr(64, 5000, 3.0, seed=9, empty_rows=(0, 63), hub=(17, 12345))
    val = cases.weights(col.size, 4)
    x = cases.dense(5000, k, 3, "integer" if k == 32 else "uniform")
    _run_all(gpu, oracle_mod, rowptr, col, val, x)


def test_rectangular_and_tiny(gpu, oracle_mod):
    for m, n in ((1, 1), (3, 1000), (1000, 3), (65, 63)):
        rowptr, col = cases.random_csr(m, n, 5.0, seed=m * 7 + n)
        val = cases.weights(col.size, 4)
        x = cases.dense(n, 20, 3)
        _run_all(gpu, oracle_mod, rowptr, col, val, x)


def test_all_rows_empty(gpu, oracle_mod):
    rowptr = np.zeros(11, np.int64)
    col = np.zeros(0, np.int64)
    val = np.zeros(0, np.float32)
    x = cases.dense(7, 16, 3)
    _run_all(gpu, oracle_mod, rowptr, col, val, x)


def test_reference_known_answers(gpu, oracle_mod):
    from isplib_amd import cabi
    rowptr, col, val, x, e_sum, e_max, e_arg = cases.readme_case()
    d = [_t(a, gpu) for a in (rowptr, col, val, x)]
    out, _ = cabi.spmm(*d, "sum")
    assert np.array_equal(out.cpu().numpy(), e_sum)
    out, arg = cabi.spmm(*d, "max")
    assert np.array_equal(out.cpu().numpy(), e_max) and np.array_equal(arg.cpu().numpy(), e_arg)
    rowptr, col, val, x, e = cases.gpu_toy_case()
    out, _ = cabi.spmm(*[_t(a, gpu) for a in (rowptr, col, val, x)], "sum")
    assert np.array_equal(out.cpu().numpy(), e)


def test_reference_known_answers_on_every_schedule(gpu):
    """The two inputs of the reference tree whose answers can be derived by hand through the planned
    schedules as well -- task list, sweep, stream (torch-built and native plan): the README case with its columns
    repeated four times (columns of an SpMM are independent, so the known answer repeats with them; k = 12 is inside
    every entry's domain), and gpu/fusedmm.cu's 16 x 16 case as it is.  All values are small integers: exact."""
    from isplib_amd import cabi
    from isplib_amd.plan import build_stream_plan, build_sweep_plan, build_task_plan
    rowptr, col, val, x, e_sum, e_max, e_arg = cases.readme_case()
    toy = cases.gpu_toy_case()
    inputs = [(rowptr, col, val, np.tile(x, (1, 4)), np.tile(e_sum, (1, 4)), np.tile(e_max, (1, 4)), np.tile(e_arg, (1, 4))),
              (toy[0], toy[1], toy[2], toy[3], toy[4], toy[4], np.tile(np.arange(16)[:, None], (1, 16)))]
    for rp, cl, vl, xx, w_sum, w_max, w_arg in inputs:
        n, k = xx.shape
        d_rp, d_cl, d_vl, d_x = _t(rp, gpu), _t(cl, gpu), _t(vl, gpu), _t(xx, gpu)
        tplan = build_task_plan(d_rp, d_cl, n, 2, 64, 0)
        out, _ = cabi.spmm_tasks(d_rp, d_cl, d_vl, tplan, d_x, "sum")
        assert np.array_equal(out.cpu().numpy(), w_sum)
        out, arg = cabi.spmm_tasks(d_rp, d_cl, d_vl, tplan, d_x, "max")
        assert np.array_equal(out.cpu().numpy(), w_max) and np.array_equal(arg.cpu().numpy(), w_arg)
        wplan = build_sweep_plan(d_rp, d_cl, n, 2, 2, 8, 2, 1)
        out, _ = cabi.spmm_sweep(d_rp, d_cl, d_vl, wplan, d_x, "sum")
        assert np.array_equal(out.cpu().numpy(), w_sum)
        out, arg = cabi.spmm_sweep(d_rp, d_cl, d_vl, wplan, d_x, "max")
        assert np.array_equal(out.cpu().numpy(), w_max) and np.array_equal(arg.cpu().numpy(), w_arg)
        for streams in (2, 4, 8):
            splan = build_stream_plan(d_rp, d_cl, d_vl, n, 2, 2, None, streams, 2)
            assert np.array_equal(cabi.spmm_stream(d_rp, cl.size, splan, d_x, "sum").cpu().numpy(), w_sum)
            nat = cabi.NativeStreamPlan(d_rp, d_cl, d_vl, n, streams, 2, 2, 2)
            assert np.array_equal(cabi.spmm_stream(d_rp, cl.size, nat, d_x, "sum").cpu().numpy(), w_sum)
            nat.close()


def test_strided_operands_through_leading_dimensions(gpu, oracle_mod):
    from isplib_amd import cabi
    rowptr, col = cases.random_csr(90, 80, 7.0, seed=21)
    val = cases.weights(col.size, 4)
    x = cases.dense(80, 48, 3)
    big_y = torch.zeros((80, 64), device=gpu)
    big_y[:, :48] = _t(x, gpu)
    big_z = torch.full((90, 56), -7.0, device=gpu)
    cabi.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, _t(rowptr, gpu), _t(col, gpu), _t(val, gpu), big_y[:, :48], big_z[:, :48])
    ref, _ = oracle_mod.spmm_fw(rowptr, col, val, x, "sum")
    _assert_sum_close(big_z[:, :48].cpu().numpy(), ref, cases.sum_tolerance(oracle_mod, rowptr, col, val, x))
    assert torch.all(big_z[:, 48:] == -7.0), "wrote outside ldz window"


def test_status_codes(gpu):
    from isplib_amd import cabi
    rowptr = torch.tensor([0, 1], device=gpu)
    col = torch.tensor([0], device=gpu)
    y = torch.ones((1, 4), device=gpu)
    z = torch.empty((1, 4), device=gpu)
    assert cabi.fusedMM_csr_hip(0x11108, rowptr, col, None, y, z, check=False) == cabi.NO_OPT_IMPL   # VOP 8: undefined
    assert cabi.fusedMM_csr_hip(0x1110F, rowptr, col, None, y, z, check=False) == cabi.UNDEFINED_USER_FUNCTION   # VOP_UDEF
    assert cabi.fusedMM_csr_hip(0x11103, rowptr, col, None, y, z, check=False) == cabi.FAIL          # VOP_ADD reads x: NULL here
    assert cabi.fusedMM_csr_hip(0x23102, rowptr, col, None, y, z, check=False) == cabi.NO_OPT_IMPL   # MEAN with MAX
    assert cabi.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, y, z, beta=1.0, check=False) == cabi.FAIL
    assert "beta" in cabi.last_error()
    assert cabi.fusedMM_csr_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, y, z, check=False) == cabi.SUCCESS
    cabi.perform_dummy_spmm(1)
    torch.cuda.synchronize()


def test_run_twice_bitwise_identical(gpu):
    from isplib_amd import cabi
    rowptr, col = cases.random_csr(500, 400, 30.0, seed=2, hub=(5, 9000))
    val = cases.weights(col.size, 4)
    x = cases.dense(400, 128, 3)
    d = [_t(a, gpu) for a in (rowptr, col, val, x)]
    a, _ = cabi.spmm(*d, "sum")
    b, _ = cabi.spmm(*d, "sum")
    assert torch.equal(a, b)


def test_slice_table_detects_unsorted_rows(gpu):
    from isplib_amd import cabi
    rowptr = torch.tensor([0, 3, 5], device=gpu)
    col = torch.tensor([0, 2, 1, 0, 1], device=gpu)
    _, ok = cabi.spmm_slices(rowptr, col, 3, 8)
    assert not ok
    col = torch.tensor([0, 1, 2, 0, 1], device=gpu)
    table, ok = cabi.spmm_slices(rowptr, col, 3, 8)
    assert ok
    t = table.view(2, 9).cpu()
    assert t[0, 0] == 0 and t[0, 8] == 3 and t[1, 0] == 3 and t[1, 8] == 5
    assert bool((t[:, 1:] >= t[:, :-1]).all())


def test_sliced_status_codes(gpu):
    from isplib_amd import cabi
    rowptr = torch.tensor([0, 1], device=gpu)
    col = torch.tensor([0], device=gpu)
    y = torch.ones((1, 4), device=gpu)
    z = torch.empty((1, 4), device=gpu)
    table, _ = cabi.spmm_slices(rowptr, col, 1, 8)
    small = torch.empty(16, dtype=torch.uint8, device=gpu)
    assert cabi.fusedMM_csr_sliced_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, table, 8, y, z, None, small,
                                       check=False) == cabi.NOT_ENOUGH_MEM
    ws = cabi.sliced_workspace("sum", 1, 4, 8, gpu)
    assert cabi.fusedMM_csr_sliced_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, table, 0, y, z, None, ws,
                                       check=False) == cabi.FAIL
    assert cabi.fusedMM_csr_sliced_hip(cabi.MSG_SPMM_SUM, rowptr, col, None, table, 8, y, z, None, ws) == cabi.SUCCESS
    torch.cuda.synchronize()
    assert torch.all(z == 1.0)


@pytest.mark.parametrize("k", (5, 7, 41, 100, 128, 602))
def test_generic_64bit_addressing_path(gpu, oracle_mod, k):
    """isplib_hip_tune(1, 0) forces the 64-bit-address kernels (the path taken when the dense operand
    is too large for a buffer descriptor); results must not change."""
    from isplib_amd import cabi
    rowptr, col = cases.random_csr(150, 140, 10.0, seed=k, empty_rows=(0, 149), hub=(70, 3000))
    val = cases.weights(col.size, 4)
    x = cases.dense(140, k, 3, "integer" if k % 2 else "uniform")
    assert cabi.lib().isplib_hip_tune(1, 0) == 0
    try:
        _run_all(gpu, oracle_mod, rowptr, col, val, x, slices=(8,))
    finally:
        cabi.lib().isplib_hip_tune(1, 1)
    _run_all(gpu, oracle_mod, rowptr, col, val, x, slices=(8,))


def test_dword_aligned_rows_take_the_vector_path(gpu, oracle_mod):
    """Rows that are only 4-byte aligned (odd K, odd leading dimensions, offset base) through fusedMM_csr_hip."""
    from isplib_amd import cabi
    rowptr, col = cases.random_csr(77, 66, 6.0, seed=3, empty_rows=(5,))
    val = cases.weights(col.size, 4)
    for k, ldy, ldz, shift in ((41, 41, 41, 0), (41, 45, 47, 1), (12, 13, 12, 3), (4, 5, 7, 1), (9, 9, 9, 2)):
        x = cases.dense(66, k, 3)
        ybuf = torch.zeros(66 * ldy + 8, device=gpu)
        y = ybuf[shift: shift + 66 * ldy].view(66, ldy)[:, :k]
        y.copy_(_t(x, gpu))
        zbuf = torch.full((77 * ldz + 8,), -7.0, device=gpu)
        z = zbuf[shift: shift + 77 * ldz].view(77, ldz)[:, :k]
        abuf = torch.full((77 * ldz + 8,), -7, dtype=torch.int64, device=gpu)
        for red in cases.REDUCES:
            ref, ref_arg = oracle_mod.spmm_fw(rowptr, col, val, x, red)
            arg = abuf[shift: shift + 77 * ldz].view(77, ldz)[:, :k] if red in ("max", "min") else None
            cabi.fusedMM_csr_hip(cabi.MESSAGE[red], _t(rowptr, gpu), _t(col, gpu), _t(val, gpu), y, z, arg)
            if red in ("sum", "mean"):
                _assert_sum_close(z.cpu().numpy(), ref, cases.sum_tolerance(oracle_mod, rowptr, col, val, x))
            else:
                assert np.array_equal(z.cpu().numpy(), ref) and np.array_equal(arg.cpu().numpy(), ref_arg)
        if ldz > k:
            pad = zbuf[shift: shift + 77 * ldz].view(77, ldz)[:, k:]
            assert torch.all(pad == -7.0), "wrote outside the k columns"


@pytest.mark.parametrize("world", (2, 4, 8))
def test_phased_sliced_spmm_equals_single_call(gpu, oracle_mod, world):
    """The overlap schedule of the row-partitioned path (local column slices from the rank's own shard
    first, remote slices after the all-gather, then the fold) is bitwise the one-call sliced SpMM."""
    from isplib_amd import cabi
    from isplib_amd.dist import RowPartition
    rowptr, col = cases.random_csr(500, 500, 40.0, seed=world, empty_rows=(0, 250), hub=(77, 3000))
    val = cases.weights(col.size, 4, "signed_int")      # integer weights and features: every sum is exact
    x = cases.dense(500, 24, 3, "integer")
    d_rowptr, d_col, d_val, d_x = (_t(a, gpu) for a in (rowptr, col, val, x))
    for red in ("sum", "max"):
        ref, ref_arg = oracle_mod.spmm_fw(rowptr, col, val, x, red)
        for rank in range(world):
            part = RowPartition(d_rowptr, d_col, d_val, 500, rank, world)
            assert part.max_rows % 192 == 0
            buf = part.gather_buffer(24)
            buf.zero_()
            for p in range(world):                     # what the all-gather leaves behind
                r0, r1 = part.x_cuts[p], part.x_cuts[p + 1]
                buf[p * part.max_rows: p * part.max_rows + (r1 - r0)] = d_x[r0:r1]
            plan = part.plan(24, red, slices=16)
            s, table, work = plan
            one, one_arg = cabi.spmm_sliced(part.rowptr, part.col_padded, part.val, table, s, buf, red, work)
            out = torch.empty_like(one)
            arg = torch.empty_like(one_arg) if one_arg is not None else None
            part.spmm_overlapped(part.shard(d_x), buf, out, plan, red, arg, gather=False)
            assert torch.equal(out, one)
            r0, r1 = part.row_cuts[rank], part.row_cuts[rank + 1]
            assert np.array_equal(out.cpu().numpy(), ref[r0:r1])          # integer data: exact
            if arg is not None:
                assert torch.equal(arg, one_arg)
                garg = torch.where(arg == part.nnz, arg.new_full((), part.total_nnz), arg + part.edge0)
                assert np.array_equal(garg.cpu().numpy(), ref_arg[r0:r1])


@pytest.mark.parametrize("k", (4, 41, 64, 128, 300))
def test_task_list_schedule(gpu, oracle_mod, k):
    """fusedMM_csr_tasks_hip with plans that exercise chunked hub rows, unsliced short rows, empty rows."""
    from isplib_amd import cabi
    from isplib_amd.plan import build_task_plan
    rowptr, col = cases.random_csr(400, 900, 30.0, seed=k, empty_rows=(0, 200, 399), hub=(123, 7000), duplicates=True)
    val = cases.weights(col.size, 4, "signed_int" if k % 2 else "uniform")
    x = cases.dense(900, k, 3, "integer" if k % 2 else "uniform")
    d_rowptr, d_col, d_val, d_x = (_t(a, gpu) for a in (rowptr, col, val, x))
    tol = cases.sum_tolerance(oracle_mod, rowptr, col, val, x)
    for slices, chunk, short in ((8, 512, 256), (16, 64, 16), (8, 100, 0), (24, 512, 10 ** 9), (1, 128, 64), (3, 256, 0), (13, 512, 128)):
        # small operands would always run in one pass: every other plan is forced through the column-panel path
        cabi.lib().isplib_hip_tune(8, 0 if slices in (8, 3, 13) else 9216)
        plan = build_task_plan(d_rowptr, d_col, 900, slices, chunk, short)
        assert plan is not None and plan.lane_off[0] == 0 and plan.lane_off[8] == plan.n_tasks
        assert int(plan.task_len.sum()) == col.size and int(plan.task_len.max()) <= chunk
        for red in cases.REDUCES:
            for dv, hv in ((d_val, val), (None, np.ones_like(val))):
                ref, ref_arg = oracle_mod.spmm_fw(rowptr, col, hv, x, red)
                out, arg = cabi.spmm_tasks(d_rowptr, d_col, dv, plan, d_x, red)
                again, _ = cabi.spmm_tasks(d_rowptr, d_col, dv, plan, d_x, red)
                assert torch.equal(out, again)
                if red in ("sum", "mean"):
                    _assert_sum_close(out.cpu().numpy(), ref, tol if dv is not None else cases.sum_tolerance(oracle_mod, rowptr, col, hv, x))
                else:
                    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32)), (red, slices)
                    assert np.array_equal(arg.cpu().numpy(), ref_arg), (red, slices)
    cabi.lib().isplib_hip_tune(8, 9216)


@pytest.mark.parametrize("k", (16, 41, 256))
def test_ordered_plain_kernel_is_bitwise_the_plain_kernel(gpu, oracle_mod, k):
    """fusedMM_csr_ordered_hip: the rows taken in a community order (isplib_amd/reorder.py on a block-structured graph), in
    a random order and in none: every reduction bit for bit the plain kernel -- values, arg, empty rows, a hub row that
    the whole workgroup shares -- and the plain kernel within the oracle's bound."""
    from isplib_amd import cabi, reorder, synth
    rowptr, col = synth.sbm_csr(3000, 120000, 6, 0.8, 400, 1.0, 5, device=gpu)
    n = 3000
    hub = torch.arange(0, n, 2, device=gpu)                                 # row 7 becomes a hub of 1500 entries (> long_row / 4 waves)
    deg = (rowptr[1:] - rowptr[:-1]).clone()
    row = torch.repeat_interleave(torch.arange(n, device=gpu), deg)
    keep = row != 7
    col2 = torch.cat([col[keep & (row < 7)], hub, col[keep & (row > 7)]])
    deg[7] = hub.numel()
    rowptr2 = torch.zeros(n + 1, dtype=torch.int64, device=gpu)
    torch.cumsum(deg, 0, out=rowptr2[1:])
    val = synth.edge_weights(col2.numel(), device=gpu)
    x = synth.features(n, k, device=gpu, integer=True)
    order = reorder.community_order(rowptr2, col2)
    assert torch.equal(torch.sort(order.long()).values, torch.arange(n, device=gpu)), "a permutation of the rows"
    ident = torch.arange(n, device=gpu, dtype=torch.int32)
    assert reorder.ordered_gather_locality(rowptr2, col2, order, 300) > 2 * reorder.ordered_gather_locality(rowptr2, col2, ident, 300)
    shuffled = torch.randperm(n, device=gpu).to(torch.int32)
    for red in ("sum", "mean", "max", "min"):
        want, want_arg = cabi.spmm(rowptr2, col2, val, x, red)
        for o in (order, shuffled, None):
            got, got_arg = cabi.spmm_ordered(rowptr2, col2, val, o, x, red)
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (red, k)
            if want_arg is not None:
                assert torch.equal(got_arg, want_arg), (red, k)
    # the handle with an order given by the caller (and forced onto the plain kernel), forward and backward
    h = cabi.GraphHandle(rowptr2, col2, val, n)
    try:
        h.set_slices(0)
        h.set_row_order(shuffled, order)
        for red in ("sum", "max"):
            got, got_arg = h.spmm(x, red)
            want, want_arg = cabi.spmm(rowptr2, col2, val, x, red)
            assert torch.equal(got, want) and (want_arg is None or torch.equal(got_arg, want_arg)), red
        colptr, _, row_t, val_t = cabi.csr2csc(rowptr2, col2, val, n, want_perm=False)
        assert torch.equal(h.spmm_backward(x, mean=False), cabi.spmm(colptr, row_t, val_t, x, "sum")[0])
        # an order that is not a permutation of the rows is refused (checked once, on the device) and changes nothing
        bad_range, twice = shuffled.clone(), shuffled.clone()
        bad_range[3] = n
        twice[5] = twice[6]
        for bad in (bad_range, twice):
            with pytest.raises(cabi.IsplibError, match="order"):
                h.set_row_order(bad, None)
            with pytest.raises(cabi.IsplibError, match="order"):
                h.set_row_order(shuffled, bad)
        got, _ = h.spmm(x, "sum")
        assert torch.equal(got, cabi.spmm(rowptr2, col2, val, x, "sum")[0])
    finally:
        h.close()
    ref, _ = oracle_mod.spmm_fw(rowptr2.cpu().numpy(), col2.cpu().numpy(), val.cpu().numpy(), x.cpu().numpy(), "max")
    got, _ = cabi.spmm_ordered(rowptr2, col2, val, order, x, "max")
    assert np.array_equal(got.cpu().numpy().view(np.uint32), ref.view(np.uint32))


def test_native_label_propagation_equals_the_torch_statement(gpu):
    """isplib_community_order_hip (rocPRIM sorts + kernels) against isplib_amd/reorder.py (torch ops): the same labels after
    the same number of rounds, the same order; on a graph with blocks the order groups them, on one without it finds
    nothing worth keeping."""
    from isplib_amd import cabi, reorder, synth
    rowptr, col, member = synth.sbm_csr(20000, 600000, 20, 0.8, 800, 1.0, 9, device=gpu, return_membership=True)
    labels_t, rounds_t = reorder.label_propagation(rowptr, col)
    order, labels, rounds = cabi.community_order(rowptr, col)
    assert rounds == rounds_t and torch.equal(labels.long(), labels_t)
    assert torch.equal(order, reorder.community_order(rowptr, col, native=False))
    # most rows of a block share one label
    agree = 0
    for b in range(20):
        lab = labels_t[member == b]
        agree += int((lab == torch.mode(lab).values).sum())
    assert agree >= 0.9 * 20000
    assert cabi.order_locality(rowptr, col, order, 1024) == pytest.approx(reorder.ordered_gather_locality(rowptr, col, order, 1024), abs=1e-12)
    assert reorder.useful_order(rowptr, col) is not None
    rowptr2, col2 = synth.chung_lu_csr(20000, 600000, 800, 1.0, 9, device=gpu)
    assert reorder.useful_order(rowptr2, col2) is None


def test_plugin_and_handle_take_the_community_order_where_the_operand_is_beyond_the_caches(gpu):
    """A square graph with blocks, mean degree 20 (the slice rule says: plain kernel) and a dense operand of 307 MB (beyond
    the Infinity Cache): the plug-in passes the community order to the operators, forward and backward, and the C handle
    finds one itself -- results bit for bit those of the plain kernel."""
    import isplib_amd
    from isplib_amd import cabi, synth
    n, k = 300000, 256
    rowptr, col = synth.sbm_csr(n, 6000000, 300, 0.8, 2000, 1.0, 4, device=gpu)
    x = synth.features(n, k, device=gpu)
    # the index-order rows in the one-pass form a community order runs (isplib_hip_tune(0, 64): the index-order DEFAULT for an
    # operand beyond the Infinity Cache is two 128-column panels since round 5, whose sums are associated differently)
    cabi.lib().isplib_hip_tune(0, 64)
    try:
        want, _ = cabi.spmm(rowptr, col, None, x, "sum")
    finally:
        cabi.lib().isplib_hip_tune(0, 0)
    panels, _ = cabi.spmm(rowptr, col, None, x, "sum")
    assert torch.allclose(panels, want, rtol=1e-5, atol=1e-4) and not torch.equal(panels, want), "the default index-order launch runs panels here"
    adj = isplib_amd.SparseTensor.from_csr(rowptr, col, None, (n, n))
    xg = x.clone().requires_grad_(True)
    out = isplib_amd.matmul(adj, xg, "sum")
    assert adj.storage._row_orders[False] is not None, "the plug-in did not look for / keep a row order"
    assert torch.equal(out.detach(), want)
    out.backward(x)
    assert adj.storage._row_orders[True] is not None
    assert torch.equal(xg.grad, want), "A is symmetric: A^T x == A x, through the transposed side's order"
    mx = isplib_amd.matmul(adj, x, "max")
    assert torch.equal(mx, cabi.spmm(rowptr, col, None, x, "max")[0])
    h = cabi.GraphHandle(rowptr, col, None, n)
    try:
        got, _ = h.spmm(x, "sum")
        assert torch.equal(got, want)
        assert torch.equal(h.spmm_backward(x, mean=False), want)
    finally:
        h.close()


def test_empty_row_convention_switch_on_every_schedule(gpu, oracle_mod):
    """the reference launcher pre-fills max / min outputs with lowest() / max() and nothing in its
    tree rewrites an empty row (csrc/fusedmm.cpp:147-150); oracle and kernels write 0 by default.  With
    isplib_hip_set_empty_row(1) ("init") every schedule -- plain, column-sliced, task list, stream, generic pipeline --
    returns the pre-fill instead (-FLT_MAX / +FLT_MAX), positions nnz as before; every other row is bit for bit what it was,
    and the oracle's own switch gives the same answer.  sum / mean are 0 either way."""
    from isplib_amd import cabi
    from isplib_amd.plan import build_stream_plan, build_task_plan
    rowptr, col = cases.random_csr(150, 130, 9.0, seed=77, empty_rows=(0, 17, 76, 149))
    val = cases.weights(col.size, 4)
    x = cases.dense(130, 48, 3)
    d = [torch.from_numpy(a).to(gpu) for a in (rowptr, col, val, x)]
    empty = np.diff(rowptr) == 0
    assert empty.sum() >= 4
    table, ok = cabi.spmm_slices(d[0], d[1], 130, 4)
    tplan = build_task_plan(d[0], d[1], 130, 4, chunk=64, short_row=4)
    splan = build_stream_plan(d[0], d[1], d[2], 130, 3, 4, None, 4, 64, minmax=True)
    sum_plan = build_stream_plan(d[0], d[1], d[2], 130, 3, 4, None, 4, 64)
    assert ok and tplan is not None and splan is not None

    def run_all(red):
        outs = {"plain": cabi.spmm(d[0], d[1], d[2], d[3], red),
                "sliced": cabi.spmm_sliced(d[0], d[1], d[2], table, 4, d[3], red),
                "tasks": cabi.spmm_tasks(d[0], d[1], d[2], tplan, d[3], red)}
        if red in ("max", "min"):
            outs["stream"] = cabi.spmm_stream_minmax(d[0], col.size, splan, d[3], red)
            outs["stream, values only"] = cabi.spmm_stream_minmax(d[0], col.size, splan, d[3], red, want_arg=False)
        else:
            outs["stream"] = (cabi.spmm_stream(d[0], col.size, sum_plan, d[3], red), None)
        return outs

    try:
        for red in ("max", "min", "sum", "mean"):
            cabi.set_empty_row("zero")
            oracle_mod.set_empty_row("zero")
            assert cabi.get_empty_row() == "zero"
            ref0, arg0 = oracle_mod.spmm_fw(rowptr, col, val, x, red)
            zero = run_all(red)
            cabi.set_empty_row("init")
            oracle_mod.set_empty_row("init")
            assert cabi.get_empty_row() == "init"
            ref1, arg1 = oracle_mod.spmm_fw(rowptr, col, val, x, red)
            init = run_all(red)
            if red in ("max", "min"):
                fill = np.float32(-np.finfo(np.float32).max if red == "max" else np.finfo(np.float32).max)
                assert np.all(ref0[empty] == 0) and np.all(ref1[empty] == fill) and np.array_equal(ref0[~empty], ref1[~empty])
                assert np.array_equal(arg0, arg1) and np.all(arg1[empty] == col.size)
            else:
                assert np.array_equal(ref0, ref1) and np.all(ref1[empty] == 0)
            for name in zero:
                z0, a0 = zero[name]
                z1, a1 = init[name]
                z0, z1 = z0.cpu().numpy(), z1.cpu().numpy()
                if red in ("max", "min"):
                    assert np.array_equal(z0, ref0) and np.array_equal(z1, ref1), (red, name)
                    if a0 is not None:
                        assert np.array_equal(a0.cpu().numpy(), arg0) and np.array_equal(a1.cpu().numpy(), arg1), (red, name)
                else:
                    assert np.array_equal(z0, z1) and np.all(z1[empty] == 0), (red, name)
        # the generic pipeline's max word follows the same switch
        word = cabi.VOP["add"] | cabi.ROP["noop"] | cabi.SOP["noop"] | cabi.VSC["noop"] | cabi.AOP["max"]      # z_i = max_j (x_i + y_j)
        st, z, zarg = cabi.fusedmm(word, d[0], d[1], d[2], torch.zeros((150, 48), device=gpu), d[3])
        assert st == 0 and np.all(z.cpu().numpy()[empty] == -np.finfo(np.float32).max) and np.all(zarg.cpu().numpy()[empty] == col.size)
    finally:
        cabi.set_empty_row("zero")
        oracle_mod.set_empty_row("zero")
