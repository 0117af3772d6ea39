"""Synthetic graphs with the shapes of the benchmark datasets (no datasets or
network on either machine).  Shapes from the reference tree: Reddit nnz
``tests/cpu/tmp/error.log:56``, node count ``tests/cpu/dataset_tester.ipynb:496``;
ogbn-products ``tests/cpu/dataset_tester.ipynb:983``.

Generator: Chung-Lu style -- undirected edges drawn with endpoint probability
proportional to a heavy-tailed (log-normal) weight per node, self-loops and
duplicates removed, topped up to the exact edge count, symmetrised, rows and
in-row columns sorted (torch_sparse order).  Runs on whatever device the
``torch.Generator`` lives on, so the full-size graphs are built on the GPU.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

SHAPES = {
    # name: (nodes, nnz, max_degree_cap, lognormal_sigma, seed)
    "cora": (2708, 10556, 168, 0.8, 0),
    "reddit": (232965, 114615892, 21657, 1.0, 1),
    "products": (2449029, 123718280, 17481, 1.1, 2),
}


def _weights(n: int, mean_deg: float, max_deg: int, sigma: float, gen: torch.Generator) -> torch.Tensor:
    w = torch.exp(torch.randn(n, generator=gen, device=gen.device, dtype=torch.float64) * sigma)
    w = w * (mean_deg / w.mean())
    for _ in range(8):   # clamp the tail, re-centre the mean
        w = w.clamp(min=0.5, max=float(max_deg))
        w = w * (mean_deg / w.mean())
    return w.clamp(max=float(max_deg))


def chung_lu_csr(n: int, nnz: int, max_deg: int, sigma: float, seed: int, device="cpu",
                 weights: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Symmetric, loop-free, duplicate-free CSR (rowptr[n+1], col[nnz]) with exactly ``nnz`` entries."""
    assert nnz % 2 == 0, "symmetric graph needs an even nnz"
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    und = nnz // 2
    w = _weights(n, nnz / n, max_deg, sigma, gen) if weights is None else weights.to(device=device, dtype=torch.float64)
    cdf = torch.cumsum(w, 0)
    cdf = cdf / cdf[-1]
    keys = torch.empty(0, dtype=torch.int64, device=device)
    need = und
    for _ in range(64):
        if need <= 0:
            break
        draw = int(need * 1.15) + 1024
        a = torch.searchsorted(cdf, torch.rand(draw, generator=gen, device=device, dtype=torch.float64)).clamp_(max=n - 1)
        b = torch.searchsorted(cdf, torch.rand(draw, generator=gen, device=device, dtype=torch.float64)).clamp_(max=n - 1)
        keep = a != b
        lo, hi = torch.minimum(a, b)[keep], torch.maximum(a, b)[keep]
        keys = torch.unique(torch.cat([keys, lo * n + hi]))
        del a, b, lo, hi, keep
        need = und - keys.numel()
    if keys.numel() < und:
        raise RuntimeError("chung_lu_csr: could not reach the requested edge count (graph too dense)")
    if keys.numel() > und:
        drop = torch.randperm(keys.numel(), generator=gen, device=device)[: keys.numel() - und]
        mask = torch.ones(keys.numel(), dtype=torch.bool, device=device)
        mask[drop] = False
        keys = keys[mask]
        del mask, drop
    lo, hi = keys // n, keys % n
    del keys
    full = torch.sort(torch.cat([lo * n + hi, hi * n + lo])).values
    del lo, hi
    row, col = full // n, full % n
    del full
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(torch.bincount(row, minlength=n), 0, out=rowptr[1:])
    return rowptr, col.contiguous()


def _csr_from_pairs(a: torch.Tensor, b: torch.Tensor, n: int, und: int, gen: torch.Generator):
    """Symmetric, loop-free, duplicate-free CSR from candidate undirected pairs (truncated to `und` edges)."""
    keep = a != b
    lo, hi = torch.minimum(a, b)[keep], torch.maximum(a, b)[keep]
    keys = torch.unique(lo * n + hi)
    if keys.numel() > und:
        keys = keys[torch.randperm(keys.numel(), generator=gen, device=keys.device)[:und]]
    lo, hi = keys // n, keys % n
    full = torch.sort(torch.cat([lo * n + hi, hi * n + lo])).values
    row, col = full // n, full % n
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=full.device)
    torch.cumsum(torch.bincount(row, minlength=n), 0, out=rowptr[1:])
    return rowptr, col.contiguous()


def rmat_csr(n: int, nnz: int, seed: int = 7, device="cpu", abc=(0.57, 0.19, 0.19)):
    """R-MAT (a,b,c = 0.57,0.19,0.19): strong community / locality structure -- the friendly case for
    caches.  Returns a symmetric CSR with ~nnz entries (duplicates dropped)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    und = nnz // 2
    draw = int(und * 1.6)
    bits = max(1, (n - 1).bit_length())
    a_, b_, c_ = abc
    r = torch.zeros(draw, dtype=torch.int64, device=device)
    c = torch.zeros(draw, dtype=torch.int64, device=device)
    for _ in range(bits):
        u = torch.rand(draw, generator=gen, device=device)
        down = (u >= a_ + b_)                    # quadrants c, d: lower half
        right = ((u >= a_) & (u < a_ + b_)) | (u >= a_ + b_ + c_)   # quadrants b, d: right half
        r = r * 2 + down.long()
        c = c * 2 + right.long()
    r, c = r % n, c % n
    return _csr_from_pairs(r, c, n, und, gen)


def uniform_csr(n: int, nnz: int, seed: int = 8, device="cpu"):
    """Uniformly random endpoints: no hubs, no locality -- the cache-hostile case."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    und = nnz // 2
    draw = int(und * 1.02) + 1024
    a = torch.randint(0, n, (draw,), generator=gen, device=device)
    b = torch.randint(0, n, (draw,), generator=gen, device=device)
    return _csr_from_pairs(a, b, n, und, gen)


def dataset_like(name: str, device="cpu", scale: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """(rowptr, col, n) of a Cora- / Reddit- / products-shaped graph.  ``scale`` < 1
    shrinks nodes and edges together (same mean degree) for CPU-sized tests."""
    n, nnz, max_deg, sigma, seed = SHAPES[name]
    if scale != 1.0:
        n = max(64, int(n * scale))
        nnz = max(2, int(nnz * scale)) // 2 * 2
        max_deg = max(8, min(max_deg, n // 4))
    rowptr, col = chung_lu_csr(n, nnz, max_deg, sigma, seed, device)
    return rowptr, col, n


def features(n: int, k: int, seed: int = 3, device="cpu", integer: bool = False) -> torch.Tensor:
    """X ~ U(-1,1) fp32 (seed 3) or integer-valued in {-3..3} (forces max/min ties)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    if integer:
        return torch.randint(-3, 4, (n, k), generator=gen, device=device).to(torch.float32)
    return torch.rand((n, k), generator=gen, device=device, dtype=torch.float32) * 2 - 1


def edge_weights(nnz: int, seed: int = 4, device="cpu") -> torch.Tensor:
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    return torch.rand(nnz, generator=gen, device=device, dtype=torch.float32)


def algorithmic_bytes(m: int, n: int, nnz: int, k: int, with_arg: bool = False) -> int:
    """B_alg of BASELINE.md section 3: every array once, reference dtypes (int64 index, fp32 value)."""
    b = nnz * (8 + 4) + (m + 1) * 8 + n * k * 4 + m * k * 4
    return b + (m * k * 8 if with_arg else 0)


def gather_bytes(m: int, nnz: int, k: int) -> int:
    """Traffic with zero cache reuse of the gathered rows."""
    return nnz * (12 + 4 * k) + m * k * 4 + (m + 1) * 8


def sbm_csr(n: int, nnz: int, communities: int, p_in: float, max_deg: int, sigma: float, seed: int, device="cpu",
            return_membership: bool = False):
    """Degree-corrected stochastic block model with the SAME degree law as `chung_lu_csr` (log-normal weights): every
    node belongs to one of `communities` blocks (assigned at random, so vertex ids carry no locality at all -- what a
    locality ordering has to find, it has to find in the edges); an undirected edge picks its first endpoint by weight
    and its second, with probability `p_in`, by weight inside the first one's block, else by weight anywhere.  Symmetric,
    loop-free, duplicate-free CSR with exactly `nnz` entries.  The structure real co-purchase / social graphs have and a
    Chung-Lu graph lacks: for SpMM operands larger than every cache it is the only reuse there is to plan for."""
    assert nnz % 2 == 0 and 1 <= communities <= n and 0.0 <= p_in <= 1.0
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    und = nnz // 2
    w = _weights(n, nnz / n, max_deg, sigma, gen)
    member = torch.randint(0, communities, (n,), generator=gen, device=device)
    by_block = torch.sort(member, stable=True).indices                    # nodes grouped by block
    w_sorted = w[by_block]
    cdf_b = torch.cumsum(w_sorted, 0)                                      # running weight in block order
    size = torch.bincount(member, minlength=communities)
    end = torch.cumsum(size, 0)
    start = end - size
    zero = torch.zeros(1, dtype=cdf_b.dtype, device=device)
    cum_before = torch.cat([zero, cdf_b])                                  # weight before sorted position i
    cdf = torch.cumsum(w, 0)
    cdf = cdf / cdf[-1]
    keys = torch.empty(0, dtype=torch.int64, device=device)
    need = und
    for _ in range(64):
        if need <= 0:
            break
        draw = int(need * 1.15) + 1024
        a = torch.searchsorted(cdf, torch.rand(draw, generator=gen, device=device, dtype=torch.float64)).clamp_(max=n - 1)
        inside = torch.rand(draw, generator=gen, device=device) < p_in
        u = torch.rand(draw, generator=gen, device=device, dtype=torch.float64)
        blk = member[a]
        lo_w, hi_w = cum_before[start[blk]], cum_before[end[blk]]
        pos = torch.searchsorted(cdf_b, lo_w + u * (hi_w - lo_w)).clamp_(max=n - 1)
        pos = torch.minimum(torch.maximum(pos, start[blk]), end[blk] - 1)
        b_in = by_block[pos]
        b_out = torch.searchsorted(cdf, u).clamp_(max=n - 1)
        b = torch.where(inside, b_in, b_out)
        keep = a != b
        lo, hi = torch.minimum(a, b)[keep], torch.maximum(a, b)[keep]
        keys = torch.unique(torch.cat([keys, lo * n + hi]))
        del a, b, lo, hi, keep, inside, u, blk, pos, b_in, b_out, lo_w, hi_w
        need = und - keys.numel()
    if keys.numel() < und:
        raise RuntimeError("sbm_csr: could not reach the requested edge count (blocks too small for their degrees)")
    if keys.numel() > und:
        drop = torch.randperm(keys.numel(), generator=gen, device=device)[: keys.numel() - und]
        mask = torch.ones(keys.numel(), dtype=torch.bool, device=device)
        mask[drop] = False
        keys = keys[mask]
        del mask, drop
    lo, hi = keys // n, keys % n
    del keys
    full = torch.sort(torch.cat([lo * n + hi, hi * n + lo])).values
    del lo, hi
    row, col = full // n, full % n
    del full
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(torch.bincount(row, minlength=n), 0, out=rowptr[1:])
    if return_membership:
        return rowptr, col.contiguous(), member
    return rowptr, col.contiguous()


def sbm_like(name: str, device="cpu", scale: float = 1.0, communities: Optional[int] = None, p_in: float = 0.8):
    """(rowptr, col, n) with the node / edge counts and degree law of `dataset_like(name)` and block structure on top:
    blocks of ~1,000 nodes unless `communities` says otherwise (ogbn-products shape: 2,449 blocks)."""
    n, nnz, max_deg, sigma, seed = SHAPES[name]
    if scale != 1.0:
        n = max(64, int(n * scale))
        nnz = max(2, int(nnz * scale)) // 2 * 2
        max_deg = max(8, min(max_deg, n // 4))
    c = max(1, n // 1000) if communities is None else communities
    rowptr, col = sbm_csr(n, nnz, c, p_in, max_deg, sigma, seed + 100, device)
    return rowptr, col, n
