#!/usr/bin/env python3
"""Experiment (round 5): the SDDMM-fused FusedMM words at the Reddit shape -- task-list form
(fusedMM_csr_udef_tasks_hip, round 3) against the stream front end (fusedMM_csr_udef_stream_hip) over a sweep of slice
counts and hub-row chunks; the SpMM-sum that gathers the same rows beside them.  Both forms against each other within
1e-4 of the largest |z|."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import isplib_amd  # noqa: E402
from isplib_amd import cabi, synth  # noqa: E402


def clock(fn, reps=5):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    dev = torch.device("cuda:0")
    ks = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "128,64,32").split(",")]
    rowptr, col, n = synth.dataset_like("reddit", device=dev)
    nnz = col.numel()
    adj = isplib_amd.SparseTensor.from_csr(rowptr, col, None, (n, n), validate=False)
    for k in ks:
        sc = 1.0 / k ** 0.5
        x, y = synth.features(n, k, seed=3, device=dev) * sc, synth.features(n, k, seed=5, device=dev) * sc
        print(f"== K={k}: SpMM-sum through the plug-in {clock(lambda: isplib_amd.matmul(adj, y)):.3f} ms", flush=True)
        for pat in ("sigmoid_embedding", "tdist_embedding"):
            word, fn = cabi.PATTERNS[pat]
            os.environ["ISPLIB_STREAM"] = "0"
            ref = isplib_amd.fusedmm(adj, x, y, pat)
            t_tasks = clock(lambda: isplib_amd.fusedmm(adj, x, y, pat))
            os.environ.pop("ISPLIB_STREAM")
            rule = cabi.suggest_fusedmm_stream(word, n, n, nnz, k)
            print(f"   {pat}: task list {t_tasks:.3f} ms; rule {rule}", flush=True)
            if rule is None:
                continue
            st, sl, ch = rule
            for slices, chunk in [(sl, ch)] + [(s_, ch) for s_ in (sl // 4, sl // 2, sl * 3 // 2, sl * 2) if s_ >= 1] + [(sl, ch // 2), (sl, ch * 2)]:
                plan = cabi.NativeStreamPlan(rowptr, col, None, n, st, slices, chunk, 0, fusedmm=True)
                _, z = cabi.fusedmm_stream(word, rowptr, nnz, plan, x, y, sop_udef=fn)
                err = float((z - ref).abs().max() / ref.abs().max())
                ms = clock(lambda: cabi.fusedmm_stream(word, rowptr, nnz, plan, x, y, sop_udef=fn))
                print(f"      stream {st} streams {slices:4d} slices chunk {chunk:6d} gens {plan.gens} parts {plan.n_parts:6d}: {ms:.3f} ms"
                      f"   max |diff| / max |z| = {err:.2e}", flush=True)
                plan.close()
        del x, y


if __name__ == "__main__":
    main()
