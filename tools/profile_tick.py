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
ap.add_argument("--float-obs", action="store_true", help="request the fp32 (N, P, O) o_adj rows instead of the packed rows the rollout uses")
args = ap.parse_args()
cfg = baseline_config(args.config, **{"runtime.num_envs": args.num_envs})
env = Pursuit_Env(cfg, num_envs=args.num_envs)
env.sim.overlap_replan = False  # one stream: per-kernel durations
N, P, T = env.num_envs, env.num_defender, env.max_steps
obs = env.sim.new_obs(packed=not args.float_obs)
reward = torch.zeros((N, P), dtype=torch.float32, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
for ep in range(args.episodes):
    env.reset()
    acts = torch.randint(0, 9, (T, N, P), dtype=torch.int32, device="cuda", generator=g)
    env.observe(obs)
    env.attacker_step()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(T)]
    ev[0].record()
    kinds = []
    for t in range(T - 1):
        env.tick(acts[t], obs, reward)
        kinds.append(env.sim.t_host % env.pe_cfg.difficulty == 0)
        ev[t + 1].record()
    torch.cuda.synchronize()
    dur = [ev[t].elapsed_time(ev[t + 1]) * 1e3 for t in range(T - 1)]
    reg = sorted(d for d, k in zip(dur, kinds) if not k)
    rep = [d for d, k in zip(dur, kinds) if k]
    print(f"episode {ep}: regular tick {sum(reg) / len(reg):.2f} us avg, {reg[len(reg) // 2]:.2f} median, {reg[0]:.2f} min; replan tick "
          f"{sum(rep) / max(1, len(rep)):.1f} us avg, {max(rep) if rep else 0:.1f} max; all ticks {sum(dur) / len(dur):.2f} us avg; "
          f"status bits {int(env.sim.status().max())}, mean A* expansions {float(env.sim.meta[:, 5].float().mean()):.1f}")

# regular ticks replayed from a hipGraph: back-to-back launches without the Python launch path in between
env.reset(); env.observe(obs); env.attacker_step(); torch.cuda.synchronize()
D = env.pe_cfg.difficulty
t0, ts0 = env.sim.t_host, env.time_step
g9 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g9):
    for k in range(D - 1):
        env.tick(acts[k], obs, reward)
env.sim.t_host, env.time_step = t0, ts0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0.0
for rep in range(10):
    e0.record(); g9.replay(); e1.record(); torch.cuda.synchronize()
    tot += e0.elapsed_time(e1)
    env.sim.t_host += D - 1
    env.tick(acts[D - 1], obs, reward)
print(f"graph-replayed regular ticks: {tot / 10 / (D - 1) * 1e3:.2f} us per launch (kernel + same-stream boundary)")

# per-phase launches (each pays its own prologue): step only, observe only, evader only (no replan)
def timeit(fn, n=40):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
env.reset()
env.observe(obs); env.attacker_step()
a0 = acts[0]
env.sim.t_host = 1  # not a replan tick
t_step = timeit(lambda: (env.sim.step(a0, reward), setattr(env.sim, "t_host", 1)))
t_obs = timeit(lambda: env.sim.observe(obs))
t_eva = timeit(lambda: env.sim.evader_step())
print(f"phase launches: step {t_step:.2f} us, observe {t_obs:.2f} us, evader(no replan) {t_eva:.2f} us")
