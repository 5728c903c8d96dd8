"""The rollout's paired GRU cell launch, fp32-MFMA kernel against the split-bf16 kernel: error against an f64 torch reference and
time per launch (both networks, one layer).  python tools/gru_cell_probe2.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_multi_agent_reinforcement_learning_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = "cuda"
torch.manual_seed(0)
mods = [torch.nn.GRU(128, 128, 1).to(dev) for _ in range(2)]
xs = [torch.randn(B, 128, device=dev) for _ in range(2)]
h0 = [torch.randn(1, B, 128, device=dev) * 0.5 for _ in range(2)]
with torch.no_grad():
    ref = []
    for x, h, m in zip(xs, h0, mods):
        m64 = torch.nn.GRU(128, 128, 1).to(dev).double()
        m64.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
        ref.append(m64(x.double().unsqueeze(0), h.double())[1])
    for mode in ("fp32", "split_bf16"):
        ops.set_cell_mode(mode)
        outs = [torch.empty_like(h) for h in h0]
        run = lambda: ops.gru_step_multi(xs, h0, mods, hiddens_out=outs)
        run()
        torch.cuda.synchronize()
        err = max(float((o.double() - r).abs().max()) for o, r in zip(outs, ref))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            run()
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"{mode:10s}: {us:7.1f} us per launch (2 cells, {B} rows)  {2 * 2.0 * B * 128 * 768 / us / 1e6:6.1f} TFLOP/s (algorithmic fp32)   max |h - h_f64| = {err:.2e}")
    o_f = [torch.empty_like(h) for h in h0]; o_s = [torch.empty_like(h) for h in h0]
    ops.set_cell_mode("fp32"); ops.gru_step_multi(xs, h0, mods, hiddens_out=o_f)
    ops.set_cell_mode("split_bf16"); ops.gru_step_multi(xs, h0, mods, hiddens_out=o_s)
    print("max |split - fp32 kernel| =", max(float((a - b).abs().max()) for a, b in zip(o_f, o_s)))
