"""Kernel-by-kernel listing of (a) one eager rollout tick program (policy_step + record + env tick) and (b) one update pass,
with the aten operator each kernel was launched from -- the attribution the rocprofv3 summaries lack.
  python3 tools/trace_programs.py [cfg2|cfg3] > gpurun_out/trace_programs.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer
from distributed_multi_agent_reinforcement_learning_amd import ops

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg = baseline_config(name)
tr = Trainer(cfg)
ag, env = tr.agent, tr.env
_, buffer, _ = ag.explore_env(env, 1)   # fills the buffer, builds the rollout state
st = ag._rstate
torch.cuda.synchronize()


def kernels_in_order(prof):
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    evs.sort(key=lambda e: e.time_range.start)
    return evs


print(f"== {name}: one eager rollout tick program ==")
with torch.no_grad():
    for _ in range(2):
        st.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        st.step()
        torch.cuda.synchronize()
tot = 0.0
for e in kernels_in_order(prof):
    tot += e.device_time
    print(f"{e.device_time:8.1f} us  {e.name[:140]}")
print(f"total kernel time {tot:.1f} us")

print(f"== {name}: one update pass (10 mini-batches), kernels grouped by launching operator ==")
with torch.enable_grad():
    ag.train(buffer, tr.total_steps, return_grads=False)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        ag.train(buffer, tr.total_steps, return_grads=False)
        torch.cuda.synchronize()
rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.self_device_time_total > 0]
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
for k, c, t in rows[:60]:
    print(f"{t / 1e3:9.2f} ms {c:6d} calls  {k[:150]}")
print(f"total {tot / 1e3:.1f} ms")
