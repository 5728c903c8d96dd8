"""Weight-gradient GEMM (a^T b over ~5e5 rows): split-K MFMA kernel vs the BLAS library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_multi_agent_reinforcement_learning_amd import ops, trainer
trainer.enable_tuned_gemms()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for K, M, N in ((492000, 384, 128), (492000, 128, 384), (492000, 128, 128), (1476000, 128, 128), (492000, 256, 128), (492000, 128, 256)):
    a = torch.randn((K, M), device="cuda"); b = torch.randn((K, N), device="cuda")
    t_lib = timeit(lambda: torch.mm(a.t(), b)); t_own = timeit(lambda: ops.wgrad(a, b))
    fl = 2.0 * K * M * N
    print(f"K={K} M={M} N={N}: library {t_lib:8.1f} us ({fl/t_lib/1e6:6.1f} TFLOP/s)   wgrad {t_own:8.1f} us ({fl/t_own/1e6:6.1f} TFLOP/s, {(M+N)*K*4/t_own/1e6:5.2f} TB/s)")
