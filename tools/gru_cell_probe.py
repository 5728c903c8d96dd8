"""Rollout GRU step: fused cell kernel vs GEMM + GEMM + gate kernel, over batch sizes (prologue vs per-tile cost)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_multi_agent_reinforcement_learning_amd import ops, trainer
trainer.enable_tuned_gemms()
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.nn.GRU(128, 128, 1).cuda()
for B in (4096, 8192, 16384, 32768, 65536, 131072):
    x = torch.randn(1, B, 128, device="cuda"); h = torch.randn(1, B, 128, device="cuda")
    with torch.no_grad():
        ops.FUSED_CELL_MIN_ROWS = 1024
        t_f = timeit(lambda: ops.gru(x, h, g))
        ops.FUSED_CELL_MIN_ROWS = 1 << 30
        t_u = timeit(lambda: ops.gru(x, h, g))
    fl = 2.0 * B * 128 * 768
    print(f"B={B}: fused cell {t_f:.1f} us ({fl/t_f/1e6:.1f} TFLOP/s)   GEMM+GEMM+gates {t_u:.1f} us")
