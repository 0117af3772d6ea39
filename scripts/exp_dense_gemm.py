"""The dense half of config 5's epoch: X W1 (232,965 x 602 by 602 x 32) is 561 MB of reads for 9 GFLOP -- memory-bound, ~0.12 ms at HBM
rate -- and takes 0.585 ms in the epoch's kernel trace (a 32x32x256 macro-tile of the BLAS library's heuristic).  Which of the two BLAS
back ends torch can call serves the epoch's skinny shapes better?  usage: exp_dense_gemm.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda:0")
n, f, h, c = 232965, 602, 32, 41
x = torch.randn(n, f, device=dev)
w1 = torch.randn(h, f, device=dev)           # nn.Linear keeps [out, in]
w2 = torch.randn(c, h, device=dev)
hid = torch.randn(n, h, device=dev)
g1 = torch.randn(n, h, device=dev)
g2 = torch.randn(n, c, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(reps):
        fn()
    e_.record()
    torch.cuda.synchronize()
    return s_.elapsed_time(e_) / reps


shapes = {
    "X W1^T      (n x 602 by 602 x 32)": lambda: torch.nn.functional.linear(x, w1),
    "H W2^T      (n x 32 by 32 x 41)": lambda: torch.nn.functional.linear(hid, w2),
    "dW1 = G1^T X (32 x n by n x 602)": lambda: g1.t().mm(x),
    "dW2 = G2^T H (41 x n by n x 32)": lambda: g2.t().mm(hid),
    "dH = G2 W2   (n x 41 by 41 x 32)": lambda: g2.mm(w2),
}
for lib in ("default", "cublas", "cublaslt"):
    if lib != "default":
        try:
            torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e:  # noqa: BLE001
            print(f"{lib}: not selectable ({e})")
            continue
    print(f"== preferred_blas_library = {lib} ({torch.backends.cuda.preferred_blas_library()})", flush=True)
    for name, fn in shapes.items():
        print(f"   {name}: {timed(fn):.3f} ms", flush=True)
