"""ops.fcra_mean (k_nbr_mean) at the update's mini-batch size and the rollout's tick size: time and bytes moved."""
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


P, E, T, d = 8, 128, 150, 3
for label, n in (("update mini-batch (410 episodes x 150 steps)", 410), ("4096 episodes x 150 steps", 4096)):
    hist = torch.randn(n, T + d, P, E, device="cuda")
    adj = (torch.rand(n * T, P, P, device="cuda") < 0.4).float()
    R = n * T
    for k in (0, 2):
        z = hist[:, d - 1 - k: d - 1 - k + T]
        ta = timeit(lambda: ops.fcra_mean(z_actor=z, adj=adj))
        tc = timeit(lambda: ops.fcra_mean(z_critic=z))
        by_a, by_c = R * P * E * 4 * 2 + R * P * P * 4, R * P * E * 4 * 2
        print(f"{label}, hop {k}: actor {ta:7.1f} us ({by_a / ta / 1e6:5.2f} TB/s)   critic {tc:7.1f} us ({by_c / tc / 1e6:5.2f} TB/s)")
    del hist, adj
R = 65536 // 2
za, zc = torch.randn(R, P, E, device="cuda"), torch.randn(R, P, E, device="cuda")
adj = (torch.rand(R, P, P, device="cuda") < 0.4).float()
out = torch.empty(2, R, P, E, device="cuda")
t = timeit(lambda: ops.fcra_mean(z_actor=za, z_critic=zc, adj=adj, out=out))
print(f"rollout tick pair, {R} environments: {t:7.1f} us ({(4 * R * P * E * 4 + R * P * P * 4) / t / 1e6:5.2f} TB/s)")
