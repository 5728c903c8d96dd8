"""Timing of the update's streaming helper kernels at the benchmark's shapes against the torch ops they replace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_multi_agent_reinforcement_learning_amd import ops


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


R = 1476000
g = torch.randn((R, 128), device="cuda"); y = torch.relu(torch.randn((R, 128), device="cuda"))
t0 = timeit(lambda: torch.ops.aten.threshold_backward(g, y, 0.0)); t1 = timeit(lambda: g.sum(0)); t2 = timeit(lambda: ops.relu_bwd_colsum(g, y))
print(f"relu backward {R}x128: threshold_backward {t0:.1f} us + sum(0) {t1:.1f} us; fused {t2:.1f} us ({3 * R * 128 * 4 / t2 / 1e6:.2f} TB/s)")
R = 492000
x = torch.randn((R, 128), device="cuda")
for ns in (1, 4, 9):
    s = torch.randn((R, ns), device="cuda")
    t0 = timeit(lambda: torch.mm(s.t(), x)); t1 = timeit(lambda: x.sum(0)); t2 = timeit(lambda: ops.wgrad_skinny(s, x, colsum_x=True, colsum_s=True))
    print(f"skinny {R}x{ns} ^T {R}x128: mm {t0:.1f} us, sum(0) {t1:.1f} us; k_wgrad_skinny {t2:.1f} us ({R * 128 * 4 / t2 / 1e6:.2f} TB/s)")
