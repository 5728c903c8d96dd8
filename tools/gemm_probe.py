"""Times the fp32 GEMM shapes of the update phase under the available BLAS back-ends (diagnostic)."""
import os, sys, time
import torch
shapes = [  # (M, K, N, description)   y = x @ W^T
    (1476000, 128, 128, "AGG_vertex fwd  (3*R*P x 128 -> 128)"),
    (492000, 384, 128, "semantic fwd    (R*P x 384 -> 128)"),
    (492000, 128, 384, "GRU input proj  (T*B x 128 -> 384)"),
    (3280, 128, 384, "GRU recurrent step (B x 128 -> 384)"),
    (3280, 384, 128, "GRU recurrent bwd  (B x 384 -> 128)"),
]
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for lib in ("hipblaslt", "hipblas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as ex:
        print("cannot select", lib, ex); continue
    for M, K, N, desc in shapes:
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); g = torch.randn(M, N, device="cuda")
        b = torch.zeros(N, device="cuda")
        t_f = bench(lambda: torch.addmm(b, x, w.t()))
        t_dx = bench(lambda: torch.mm(g, w))
        t_dw = bench(lambda: torch.mm(g.t(), x))
        fl = 2.0 * M * K * N
        print(f"{lib:10s} {desc:42s} fwd {t_f*1e6:9.1f} us {fl/t_f/1e12:6.1f} TF | dX {t_dx*1e6:9.1f} us {fl/t_dx/1e12:6.1f} TF | dW {t_dw*1e6:9.1f} us {fl/t_dw/1e12:6.1f} TF")
