"""Runs only the environment kernels (one episode of fused ticks with random actions) -- the target of the rocprofv3
kernel-trace / PMC passes whose summaries are committed under profiles/.
  rocprofv3 --kernel-trace --stats -- python3 tools/profile_tick.py
  rocprofv3 --pmc FETCH_SIZE -- python3 tools/profile_tick.py      (and a second pass with WRITE_SIZE)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config  # noqa: E402
from distributed_multi_agent_reinforcement_learning_amd.pursuit_env import Pursuit_Env  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfg2")
ap.add_argument("--num-envs", type=int, default=4096)
ap.add_argument("--episodes", type=int, default=1)
args = ap.parse_args()
cfg = baseline_config(args.config, **{"runtime.num_envs": args.num_envs})
env = Pursuit_Env(cfg, num_envs=args.num_envs)
env.sim.overlap_replan = False  # one stream: per-kernel durations
N, P, T = env.num_envs, env.num_defender, env.max_steps
obs = env.sim.new_obs()
reward = torch.zeros((N, P), dtype=torch.float32, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
for ep in range(args.episodes):
    env.reset()
    acts = torch.randint(0, 9, (T, N, P), dtype=torch.int32, device="cuda", generator=g)
    env.observe(obs)
    env.attacker_step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(T - 1):
        env.tick(acts[t], obs, reward)
    e1.record()
    torch.cuda.synchronize()
    print(f"episode {ep}: {e0.elapsed_time(e1) / (T - 1) * 1e3:.2f} us per tick launch (avg incl. replan ticks), "
          f"status bits {int(env.sim.status().max())}, mean A* expansions {float(env.sim.meta[:, 5].float().mean()):.1f}")
