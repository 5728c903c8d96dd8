"""Launch durations of the single-phase variants of the tick kernel (each includes the launch + prologue floor)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
from distributed_multi_agent_reinforcement_learning_amd.pursuit_env import Pursuit_Env
cfg = baseline_config("cfg2")
env = Pursuit_Env(cfg, num_envs=4096)
env.sim.overlap_replan = False
env.reset()
obs = env.sim.new_obs()
acts = torch.randint(0, 9, (4096, 8), dtype=torch.int32, device="cuda")
rew = torch.zeros(4096, 8, device="cuda")
env.observe(obs); env.attacker_step()
def timeit(fn, n=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def keep_t(fn):
    def g():
        env.sim.t_host = 1
        fn()
        env.sim.t_host = 1
    return g
print(f"observe only  {timeit(lambda: env.sim.observe(obs)):7.2f} us")
print(f"step only     {timeit(keep_t(lambda: env.sim.step(acts, rew))):7.2f} us")
print(f"evader only   {timeit(keep_t(lambda: env.sim.evader_step())):7.2f} us")
print(f"fused tick    {timeit(keep_t(lambda: env.sim.tick(acts, obs, rew))):7.2f} us")
x = torch.zeros(4096 * 64, device="cuda")
print(f"torch add_ on 256k floats (launch floor) {timeit(lambda: x.add_(1.0)):7.2f} us")
