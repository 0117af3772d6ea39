"""Locality ordering of the rows of a graph for SpMM operands larger than every cache (the ogbn-products shape: X is
2.5 GB at K=256, ten times the Infinity Cache).

No schedule of this library reuses a gathered row of such an operand; the only reuse there is lies in the graph's own
structure: rows of one community gather mostly each other.  `fusedMM_csr_ordered_hip` takes the rows in a given order,
every XCD walking a contiguous range of positions, so rows that sit next to each other in the order are worked on at the
same time behind the same 4 MiB L2 -- nothing is moved, the result is bit for bit that of the plain kernel.  This module
finds the order: label propagation (Raghavan et al. 2007), synchronous, on the device, with plain torch ops -- one
sort of nnz keys per round, a handful of rounds, once per graph (the cost class of the plan builders in plan.py).
The reference has no counterpart (its CPU kernel walks rows in index order, csrc/fusedmm.cpp:198).
"""
from __future__ import annotations

import torch


def label_propagation(rowptr: torch.Tensor, col: torch.Tensor, rounds: int = 8, seed: int = 0, min_change: float = 0.01):
    """Community labels [n] (int64; label = a node id of the community) of a square graph by synchronous label
    propagation: every node takes the label most of its neighbours carry, ties broken by a per-round hash of the label
    (a fixed seed: the same graph gives the same labels).  Stops after `rounds` rounds or when fewer than `min_change`
    of the nodes changed.  Returns (labels, rounds run)."""
    n = rowptr.numel() - 1
    dev = col.device
    nnz = col.numel()
    labels = torch.arange(n, dtype=torch.int64, device=dev)
    if nnz == 0 or n == 0:
        return labels, 0
    deg = rowptr[1:] - rowptr[:-1]
    row = torch.repeat_interleave(torch.arange(n, dtype=torch.int64, device=dev), deg)
    big = 1 << 20                                           # tie-break range below one count
    done = 0
    for r in range(rounds):
        key = torch.sort(row * n + labels[col]).values      # (row, neighbour label) runs
        ukey, cnt = torch.unique_consecutive(key, return_counts=True)
        del key
        urow, ulab = ukey // n, ukey % n
        del ukey
        tie = ((ulab * 2654435761 + (seed + r) * 40503 + 12345) >> 7) % big
        # most frequent label of a row; equal count: the larger hash; equal hash: the larger label (reorder.hip: ro_mode_kernel).
        # Two stages, so that nothing is packed beyond 63 bits: (count, hash) first -- count < 2^31 edges, hash < 2^20 --
        # then the label among the winners (one packed key (cnt * 2^20 + tie) * n + label overflows int64 once a row has
        # 2^22 neighbours under one label at n ~ 2^21)
        score = cnt * big + tie
        best = torch.full((n,), -1, dtype=torch.int64, device=dev)
        best.scatter_reduce_(0, urow, score, reduce="amax", include_self=True)
        first = score == best[urow]
        lab_best = torch.full((n,), -1, dtype=torch.int64, device=dev)
        lab_best.scatter_reduce_(0, urow[first], ulab[first], reduce="amax", include_self=True)
        win = first & (ulab == lab_best[urow])
        del first, lab_best
        new = labels.clone()
        new[urow[win]] = ulab[win]
        changed = int((new != labels).sum())
        labels = new
        done = r + 1
        del urow, ulab, cnt, tie, score, best, win, new
        if changed < min_change * n:
            break
    return labels, done


def community_order(rowptr: torch.Tensor, col: torch.Tensor, rounds: int = 8, seed: int = 0, native: bool = True) -> torch.Tensor:
    """int32 [n]: position -> row, rows of one community (label propagation) next to each other, communities in the
    order of their labels, rows of a community in index order.  A permutation of [0, n).  native: the library's own
    implementation (isplib_community_order_hip: rocPRIM sorts + three kernels, no torch op) -- the same labels as the
    torch statement above (tests/test_gpu_parity.py holds the two together)."""
    if native and col.is_cuda:
        from . import cabi
        return cabi.community_order(rowptr, col, rounds, seed)[0]
    labels, _ = label_propagation(rowptr, col, rounds, seed)
    return torch.sort(labels, stable=True).indices.to(torch.int32)


def useful_order(rowptr: torch.Tensor, col: torch.Tensor, rounds: int = 8, window: int = 1024):
    """The community order of a square graph if it found structure, else None: kept when at least a fifth of the stored
    entries AND twice the share of the index order lie within `window` positions (the rows an XCD has in flight) of their
    row.  A structure-free graph (Chung-Lu: 0.1 % -> 0.1 %) keeps the index order and loses nothing but the one-off
    ~0.2 s; the result of the SpMM is the same bits either way."""
    from . import cabi
    order, _, _ = cabi.community_order(rowptr, col, rounds)
    before = cabi.order_locality(rowptr, col, None, window)
    after = cabi.order_locality(rowptr, col, order, window)
    return order if after >= 0.2 and after >= 2.0 * before else None


def ordered_gather_locality(rowptr: torch.Tensor, col: torch.Tensor, order: torch.Tensor, window: int = 1024) -> float:
    """Share of the stored entries whose column lies within `window` positions of its row in `order` (1.0 = every
    neighbour is worked on at about the same time): a quick, kernel-free figure of what an order found."""
    n = rowptr.numel() - 1
    pos = torch.empty(n, dtype=torch.int64, device=col.device)
    pos[order.to(torch.int64)] = torch.arange(n, dtype=torch.int64, device=col.device)
    deg = rowptr[1:] - rowptr[:-1]
    row = torch.repeat_interleave(torch.arange(n, dtype=torch.int64, device=col.device), deg)
    near = (pos[row] - pos[col]).abs() <= window
    return float(near.sum()) / max(1, col.numel())
