"""Projection for "reuse across output rows without a workgroup barrier" (round-3 review, item 8): how many of the gathers
of the stream schedule are REPEAT references -- a column of A (= row of y) that the same slot / wave / CU has already
gathered inside the same column slice -- on the Reddit-shaped graph?  That share is the ceiling of what any slot-, wave-
or CU-private hot-column pass can take out of the gather stream (each repeat served from registers / LDS instead), before
the price of serving it (an LDS read-modify-write of the accumulator row per application: DESIGN.md section 4.2c).
Rows are dealt to groups the way the plan deals them (by length, i.e. at random with respect to columns); hub rows are
not split here, which makes the figures an UPPER bound.  Runs on the CPU (numpy); ~2 minutes.
usage: exp_dup_projection.py [slices=31]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from isplib_amd import synth

slices = int(sys.argv[1]) if len(sys.argv) > 1 else 31
rowptr, col, n = synth.dataset_like("reddit", device="cpu")
rowptr, col = rowptr.numpy(), col.numpy()
nnz = col.size
deg = np.diff(rowptr)
row = np.repeat(np.arange(n, dtype=np.int64), deg)
rng = np.random.default_rng(0)
perm = rng.permutation(n)                     # position of a row in the deal
print(f"N={n} nnz={nnz} slices={slices} (a column lies in exactly one slice, so (group, column) pairs are per slice)")
for name, rows_per_group in (("slot (16 rows, one stream)", 16), ("wave (64 rows, 4 slots)", 64), ("workgroup (4 waves, 256 rows)", 256),
                             ("CU (8 waves, 512 rows)", 512)):
    group = perm[row] // rows_per_group
    key = group * n + col
    key.sort()
    distinct = 1 + int(np.count_nonzero(key[1:] != key[:-1]))
    # references beyond the first of a (group, column): what a private hot-column pass could serve without a gather
    repeats = nnz - distinct
    # the same restricted to columns referenced >= 3 times by the group (a pass has per-column overhead: cheap ones only)
    starts = np.flatnonzero(np.concatenate(([True], key[1:] != key[:-1])))
    counts = np.diff(np.concatenate((starts, [nnz])))
    r3 = int((counts[counts >= 3] - 1).sum())
    print(f"{name:34s}: {repeats / nnz * 100:5.2f} % of the edges are repeat references; {r3 / nnz * 100:5.2f} % on columns met >= 3 times; "
          f"hot columns per group and slice: {np.count_nonzero(counts >= 2) / (n / rows_per_group) / slices:.1f}")
    del group, key, starts, counts
