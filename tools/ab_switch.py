"""A/B of one boolean switch of ops (RELU_LINK: the ReLU backward in the input-gradient GEMMs' epilogues / separate passes;
PPO_FROM_PROB: Categorical inside the loss launch / torch.distributions) in ONE process, alternating iteration by iteration: boxes of the
pool differ by a few percent, two runs on two boxes cannot resolve a 1 % change.   python tools/ab_switch.py cfg3 RELU_LINK"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_multi_agent_reinforcement_learning_amd import ops
from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
switch = sys.argv[2] if len(sys.argv) > 2 else "RELU_LINK"
assert isinstance(getattr(ops, switch), bool)
tr = Trainer(baseline_config(name))
for _ in range(3):
    tr.iterate()
torch.cuda.synchronize()
acc = {True: [], False: []}
for i in range(14):
    setattr(ops, switch, i % 2 == 0)
    on = getattr(ops, switch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.iterate()
    t_host = 1e3 * (time.perf_counter() - t0)
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0)
    if i >= 2:      # the first iteration of either setting may re-capture tick programs or re-plan the update group: not counted
        acc[on].append((dt,) + tuple(tr.last_breakdown_ms()))
    print(f"iter {i}: {switch} {on}: {dt:7.1f} ms (host returned after {t_host:7.1f})  rollout / update {tuple(round(x, 1) for x in tr.last_breakdown_ms())}", flush=True)
for k, v in acc.items():
    n = len(v)
    print(f"{name} {switch}={k}: iteration {sum(x[0] for x in v) / n:.2f} ms, rollout {sum(x[1] for x in v) / n:.2f}, update {sum(x[2] for x in v) / n:.2f}")
