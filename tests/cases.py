"""Seeded input cases shared by the CPU (oracle) and GPU (parity) tests."""
from __future__ import annotations

import numpy as np

REDUCES = ("sum", "mean", "max", "min")
# feature widths the callers actually produce (32/41 GCN, 602/608 SAGE) + vector-width edge cases
WIDTHS = (1, 2, 3, 16, 32, 41, 64, 100, 128, 256, 602)


def random_csr(m, n, avg_deg, seed, empty_rows=(), hub=None, duplicates=False, sort_cols=True):
    """CSR with controllable pathologies.  hub = (row, degree)."""
    rng = np.random.default_rng(seed)
    deg = rng.poisson(avg_deg, m).astype(np.int64)
    for r in empty_rows:
        deg[r] = 0
    if hub is not None:
        deg[hub[0]] = hub[1]
    rowptr = np.zeros(m + 1, np.int64)
    np.cumsum(deg, out=rowptr[1:])
    col = rng.integers(0, n, rowptr[-1]).astype(np.int64)
    if duplicates and col.size > 4:
        # force equal neighbours inside rows (legal: duplicates simply add / tie)
        for r in range(m):
            b, e = rowptr[r], rowptr[r + 1]
            if e - b >= 2:
                col[b + 1] = col[b]
    if sort_cols:
        for r in range(m):
            b, e = rowptr[r], rowptr[r + 1]
            col[b:e] = np.sort(col[b:e], kind="stable")
    return rowptr, col


def dense(n, k, seed, kind="uniform"):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return (rng.random((n, k), np.float32) * 2 - 1).astype(np.float32)
    if kind == "integer":      # forces max/min ties; sums exact in fp32
        return rng.integers(-3, 4, (n, k)).astype(np.float32)
    if kind == "constant":     # all-ties
        return np.full((n, k), 0.5, np.float32)
    if kind == "signed_zero":
        x = rng.integers(-1, 2, (n, k)).astype(np.float32)
        x[x == 0] = np.where(rng.random(np.count_nonzero(x == 0)) < 0.5, np.float32(0.0), np.float32(-0.0))
        return x
    if kind == "denormal":     # products and sums in the fp32 subnormal range: no flush-to-zero anywhere
        return ((rng.random((n, k), np.float32) * 2 - 1) * np.float32(3e-38)).astype(np.float32)
    if kind == "nonfinite":
        x = (rng.random((n, k), np.float32) * 2 - 1).astype(np.float32)
        flat = x.reshape(-1)
        idx = rng.choice(flat.size, max(3, flat.size // 50), replace=False)
        flat[idx[0::3]] = np.nan
        flat[idx[1::3]] = np.inf
        flat[idx[2::3]] = -np.inf
        return x
    raise ValueError(kind)


def weights(nnz, seed, kind="uniform"):
    rng = np.random.default_rng(seed)
    if kind == "unit":
        return np.ones(nnz, np.float32)
    if kind == "uniform":
        return rng.random(nnz, np.float32).astype(np.float32)
    if kind == "signed_int":
        w = rng.integers(-2, 3, nnz).astype(np.float32)
        return w
    raise ValueError(kind)


# the two inputs in the reference tree with derivable answers
def readme_case():
    """README.md:105-116 -- COO with a duplicate (0,0); CSR in stable torch_sparse order."""
    rowptr = np.array([0, 3, 4, 5], np.int64)
    col = np.array([0, 0, 2, 0, 1], np.int64)
    val = np.array([3, -2, 2, 4, 3], np.float32)
    x = np.array([[1, 0, 2], [4, 0, 0], [0, 3, 0]], np.float32)
    expect_sum = np.array([[1, 6, 2], [4, 0, 8], [12, 0, 0]], np.float32)
    expect_max = np.array([[3, 6, 6], [4, 0, 8], [12, 0, 0]], np.float32)
    expect_argmax = np.array([[0, 2, 0], [3, 3, 3], [4, 4, 4]], np.int64)
    return rowptr, col, val, x, expect_sum, expect_max, expect_argmax


def gpu_toy_case():
    """gpu/fusedmm.cu:60-118 -- 16x16 diagonal, val 2.0, mat 10 with 10(i+1) on the diagonal."""
    m = 16
    rowptr = np.arange(m + 1, dtype=np.int64)
    col = np.arange(m, dtype=np.int64)
    val = np.full(m, 2.0, np.float32)
    x = np.full((m, m), 10.0, np.float32)
    x[np.arange(m), np.arange(m)] = 10.0 * (np.arange(m) + 1)
    return rowptr, col, val, x, 2.0 * x


def sum_tolerance(oracle, rowptr, col, val, x, rel=1e-5):
    """Per-element bound rel * sum_j |val_j * x_j| (BASELINE.md section 3 parity rule) + 1 ulp-ish floor."""
    mag, _ = oracle.spmm_fw(rowptr, col, np.abs(val), np.abs(x), "sum")
    return rel * mag + 1e-30


def rowwise_relative_error(got, ref):
    """||got_i - ref_i||_2 / ||ref_i||_2 per row (fp64 arithmetic), rows with ||ref_i|| = 0 skipped after checking that
    got_i is zero there too: the conventional reading of BASELINE.json's "1e-5 relative fp32" for a row-producing op."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    den = np.sqrt((ref * ref).sum(1))
    num = np.sqrt(((got - ref) ** 2).sum(1))
    zero = den == 0
    assert np.all(num[zero] == 0), "rows whose reference is exactly zero must be exactly zero"
    return num[~zero] / den[~zero]
