"""Per-expansion latency of the wave-per-problem A* (k_astar): one long search alone, 4096 copies of it, 4096 trivial ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distributed_multi_agent_reinforcement_learning_amd import pe_env

W = H = 40
L = pe_env.load_library()
def run(obs, sg, reps=5):
    n = obs.shape[0]
    obs_d = torch.as_tensor(obs).cuda(); sg_d = torch.as_tensor(sg).cuda()
    path = torch.zeros((n, 256, 2), dtype=torch.int16, device="cuda"); lens = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.pe_astar_batch(W, H, n, pe_env._ptr(obs_d), pe_env._ptr(sg_d), pe_env._ptr(path), pe_env._ptr(lens), 256, pe_env._stream())
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts), lens.cpu().numpy()

wall = np.zeros((W + 1, H + 1), np.uint8); wall[20, 0:39] = 1
for name, n, grid, sg in (("long x1", 1, wall, (5, 5, 35, 5)), ("long x256", 256, wall, (5, 5, 35, 5)), ("long x1280", 1280, wall, (5, 5, 35, 5)),
                          ("long x4096", 4096, wall, (5, 5, 35, 5)), ("trivial x4096", 4096, wall, (5, 5, 6, 6)), ("open40 x4096", 4096, np.zeros_like(wall), (0, 0, 40, 40))):
    t, lens = run(np.broadcast_to(grid, (n, W + 1, H + 1)).copy(), np.tile(np.asarray(sg, np.int32), (n, 1)))
    print(f"{name}: {t:.1f} us, path len {lens[0,0]}, expansions {lens[0,1]}, us/expansion {t/max(1,lens[0,1]):.3f}")
