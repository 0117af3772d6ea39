#!/usr/bin/env python3
"""Generates tests/golden/spmm_ref_layer.npz with the REFERENCE's own operator +
autograd layer: /root/reference/csrc/fusedmm.cpp compiled in place into
oracle/_ref/_fusedmm_ref.so (oracle/Makefile), its two undefined externs
resolved by the C restatement oracle/fusedmm_oracle.c.

So the forward values come from the restated kernel body (the reference's body
is an absent third-party library), while everything above it -- output
initialisation, arg sentinel, message selection, saved tensors and all four
backward formulas (csrc/fusedmm.cpp:113-518) -- is the reference's compiled code.

Run in THIS container only (needs /root/reference for the build; the .so also
travels to the GPU box).  Must not import isplib_amd: both libraries register
the same torch.ops.isplib.* names.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from tests import cases  # noqa: E402


def build_cases():
    out = {}
    rp, col, val, x, *_ = cases.readme_case()
    out["readme"] = (rp, col, val, x, 3)
    rp, col, val, x, _ = cases.gpu_toy_case()
    out["gputoy"] = (rp, col, val, x, 16)
    rp, col = cases.random_csr(40, 37, 6.0, seed=101, empty_rows=(0, 19, 39), duplicates=True)
    out["rand_k5"] = (rp, col, cases.weights(col.size, 4), cases.dense(37, 5, 3), 37)
    out["rand_k16_int"] = (rp, col, cases.weights(col.size, 4, "signed_int"), cases.dense(37, 16, 3, "integer"), 37)
    rp, col = cases.random_csr(23, 61, 9.0, seed=202, empty_rows=(5,))
    out["rect_k33"] = (rp, col, cases.weights(col.size, 4), cases.dense(61, 33, 3), 61)
    rp, col = cases.random_csr(12, 300, 2.0, seed=303, hub=(4, 2500))
    out["hub_k32"] = (rp, col, cases.weights(col.size, 4), cases.dense(300, 32, 3), 300)
    rp, col = cases.random_csr(30, 30, 8.0, seed=404)
    out["ties_k8"] = (rp, col, cases.weights(col.size, 0, "unit"), cases.dense(30, 8, 3, "constant"), 30)
    return out


def main():
    ref_so = oracle.REF_LIB_PATH
    if not os.path.exists(ref_so):
        raise SystemExit(f"{ref_so} missing: run `make -C oracle all` where /root/reference exists")
    torch.ops.load_library(ref_so)
    ops = torch.ops.isplib
    blobs = {}
    for name, (rowptr, col, val, x, ncols) in build_cases().items():
        row, rowcount, colptr, csr2csc = oracle.csr_transpose(rowptr, col, ncols)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
        g = cases.dense(rowptr.size - 1, x.shape[1], 5)
        blobs[f"{name}/rowptr"], blobs[f"{name}/col"], blobs[f"{name}/val"] = rowptr, col, val
        blobs[f"{name}/x"], blobs[f"{name}/g"], blobs[f"{name}/ncols"] = x, g, np.int64(ncols)

        # sum: operands cached by the wrapper at isplib/__init__.py:79-80
        xs = t(x).requires_grad_(True)
        out = ops.fusedmm_spmm(t(row), t(rowptr), t(col), t(val), t(colptr), t(csr2csc), xs,
                               t(val[csr2csc]), t(row[csr2csc]))
        out.backward(t(g))
        blobs[f"{name}/sum/out"], blobs[f"{name}/sum/dx"] = out.detach().numpy(), xs.grad.numpy()

        # mean: intended weights (csrc/fusedmm.cpp:357-364)
        _, new_row, new_rowcount = oracle.mean_bw_weights(rowptr, col, val, ncols)
        xs = t(x).requires_grad_(True)
        out = ops.fusedmm_spmm_mean(t(row), t(rowptr), t(col), t(val), t(rowcount), t(colptr), t(csr2csc), xs,
                                    t(new_row), t(new_rowcount))
        out.backward(t(g))
        blobs[f"{name}/mean/out"], blobs[f"{name}/mean/dx"] = out.detach().numpy(), xs.grad.numpy()

        for red, fn in (("max", ops.fusedmm_spmm_max), ("min", ops.fusedmm_spmm_min)):
            xs = t(x).requires_grad_(True)
            vs = t(val).requires_grad_(True)
            out, arg = fn(t(rowptr), t(col), vs, xs)
            out.backward(t(g))
            blobs[f"{name}/{red}/out"], blobs[f"{name}/{red}/arg"] = out.detach().numpy(), arg.numpy()
            blobs[f"{name}/{red}/dx"], blobs[f"{name}/{red}/dval"] = xs.grad.numpy(), vs.grad.numpy()
    path = os.path.join(HERE, "spmm_ref_layer.npz")
    np.savez_compressed(path, **blobs)
    print(f"wrote {path}: {len(blobs)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
