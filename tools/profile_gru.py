"""Runs the hand-written fp32-MFMA kernels of the update / rollout at the benchmark's shapes (cfg2: mini-batch 410 episodes,
T = 150, P = 8 -> B = 3280 sequence rows; rollout B = 32768 rows) -- the target of the rocprofv3 passes whose MFMA-utilisation
counters are committed under profiles/ (north star: "rocprof-reported MFMA utilisation for the GRU GEMMs"):
  rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/profile_gru.py
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES --output-format csv -- python3 tools/profile_gru.py
`--grouped`: the form the update launches since round 3 -- the layers of all ten mini-batches and both networks (20 x 205 workgroups,
the last mini-batch 203) in ONE launch per direction (ops.gru_multi(grouped=True), MAPPO._train_grouped).
Summarise with tools/mfma_summary.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from distributed_multi_agent_reinforcement_learning_amd import ops

dev = "cuda"
T, B, Br, Kr = 150, 3280, 32768, 492000
torch.manual_seed(0)
w = lambda *s: torch.randn(*s, device=dev) * 0.08
gm = SimpleNamespace(num_layers=1, weight_ih_l0=w(384, 128), weight_hh_l0=w(384, 128), bias_ih_l0=torch.zeros(384, device=dev), bias_hh_l0=torch.zeros(384, device=dev))
for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
    getattr(gm, k).requires_grad_(True)
if "--grouped" in sys.argv:
    Bs = [3280] * 18 + [3248] * 2          # 9 mini-batches of 410 episodes and one of 406, actor and critic each
    xs = [torch.randn(T, b, 128, device=dev, requires_grad=True) for b in Bs]
    h0s = [torch.zeros(1, b, 128, device=dev) for b in Bs]
    for mode in ("fp32", "split_bf16"):    # k_gru_seq_fwd2 / bwd2, then k_gru_seq_fwd_sb / bwd_sb (the default)
        ops.SEQ_MODE = mode
        for rep in range(3):
            outs = ops.gru_multi(xs, h0s, [gm] * len(Bs), grouped=True)
            torch.autograd.backward([o.sum() for o in outs])
    with torch.no_grad():                  # the update's 384-input product (dgi W_ih: 492 000 rows) on the split-bf16 GEMM
        dg, W3 = torch.randn(Kr, 384, device=dev), w(128, 384)
        for rep in range(10):
            ops.split_linear(dg, W3)
    torch.cuda.synchronize()
    print("done (grouped)")
    sys.exit(0)
x = torch.randn(T, B, 128, device=dev, requires_grad=True)
h0 = torch.zeros(1, B, 128, device=dev)
for mode in ("fp32", "split_bf16"):
    ops.SEQ_MODE = mode
    for rep in range(4):   # k_gru_seq_fwd + k_gru_seq_bwd + k_wgrad (the sequence mode of the update)
        out, _ = ops.gru(x, h0, gm)
        out.backward(torch.randn_like(out))
xr, hr = torch.randn(1, Br, 128, device=dev), torch.randn(1, Br, 128, device=dev)
with torch.no_grad():
    for rep in range(20):  # k_gru_cell (one rollout step, fp32 MFMA: the evaluator's path and runtime.matmul: fp32)
        ops.gru(xr, hr, gm)
    # what the tick launches by default: actor's and critic's cell in one launch on exact bf16 operand splits (k_gru_cell_sb)
    xs, hs, ho = [xr[0], xr[0].clone()], [hr, hr.clone()], [torch.empty_like(hr), torch.empty_like(hr)]
    for rep in range(20):
        ops.gru_step_multi(xs, hs, [gm, gm], hiddens_out=ho)
    # the rollout's Linear layers (k_sb_gemm_n128) at the tick's shapes, and a mini-batch's GRU input projection (384 outputs)
    m3 = torch.randn(6 * Br, 128, device=dev); e3 = torch.randn(2 * Br, 384, device=dev); o2 = torch.randn(2 * Br, 128, device=dev)
    W1, W3, b1 = w(128, 128), w(128, 384), torch.zeros(128, device=dev)
    xg, Wg, bg = torch.randn(Kr, 128, device=dev), w(384, 128), torch.zeros(384, device=dev)
    for rep in range(10):
        ops.split_linear(m3, W1, b1, True)
        ops.split_linear(e3, W3, None, False, out=o2, addend=o2)
        ops.split_linear(xg, Wg, bg)
torch.cuda.synchronize()
print("done")
