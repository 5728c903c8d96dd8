"""Micro-benchmark of the batched env_n2n tick (BASELINE config 4 geometry: 16 pursuers, 8192 envs): env-steps/s and the
achieved fraction of the HBM roofline.  Algorithmic bytes per env-step: state read + write (5 f64 per agent), actions,
evader command, reward / active / done, fp32 observations (p_state, e_state, pp_adj, pe_adj)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_multi_agent_reinforcement_learning_amd.n2n_env import ParticleEnv
P, E, N, T = 16, 1, 8192, 100
env = ParticleEnv(num_envs=N, episode_limit=T)
env.initialize(P, E)
env.reset()
acts = torch.randint(0, 9, (T, N, P), dtype=torch.int32, device="cuda")
cmds = torch.rand(T, N, E, dtype=torch.float64, device="cuda") * 2 - 1
for t in range(5):
    env.evader_step(cmds[t]); env.step(acts[t])
env.reset()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for t in range(T):
    env._cmd = cmds[t]
    env.step(acts[t])
e1.record(); torch.cuda.synchronize()
dt = e0.elapsed_time(e1) / T * 1e-3
B = 2 * (P + E) * 5 * 8 + 4 * P + 8 * E + 4 * P + P + 1 + 4 * (3 * P + 3 * E + P * P + P * E)
print(json.dumps({"kernel": "k_n2n<tick>", "envs": N, "pursuers": P, "us_per_launch": round(dt * 1e6, 2), "env_steps_per_s": round(N / dt),
                  "bytes_per_env_step": B, "achieved_GBps": round(N * B / dt / 1e9, 1), "frac_of_8TBps": round(N * B / dt / 8e12, 4)}))
