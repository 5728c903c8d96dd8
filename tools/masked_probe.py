"""ops.input_grad_masked at the update's mini-batch size: plain input gradient, ReLU backward in the epilogue with the fp32 activations as
mask, and with the sign bits the producer wrote -- beside the separate relu_bwd_colsum pass they replace."""
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


R = 492000
ops.set_matmul_mode("split_bf16")
with torch.no_grad():
    for n_in, n_out in ((256, 128), (384, 128), (128, 384)):
        g = torch.randn(R, n_out, device="cuda")
        y = torch.relu(torch.randn(R, n_in, device="cuda"))
        W = torch.randn(n_out, n_in, device="cuda") * 0.1
        bits = ((y > 0).view(R, n_in // 8, 8).to(torch.int32) * (1 << torch.arange(8, device="cuda", dtype=torch.int32))).sum(-1).to(torch.uint8)
        t0 = timeit(lambda: ops.input_grad(g, W))
        t1 = timeit(lambda: ops.input_grad_masked(g, W, y, n_in))
        t2 = timeit(lambda: ops.input_grad_masked(g, W, y, n_in, bits=bits))
        d = torch.randn(R, n_in, device="cuda")
        t3 = timeit(lambda: ops.relu_bwd_colsum(d, y))
        print(f"{n_in:4d} <- {n_out:4d}, {R} rows: plain {t0:7.1f} us | masked by activations {t1:7.1f} us | masked by sign bits {t2:7.1f} us | "
              f"separate relu' + bias-sum pass over ({R}, {n_in}) {t3:7.1f} us")
        del g, y, W, bits, d
