"""A/B of ops.RELU_LINK in one process: iteration and update time with the ReLU backward in the input-gradient GEMMs' epilogues and
with the separate passes, alternating (same box, same clocks).  Also how long the host needs to ISSUE the update (returns from train())
against when the device finishes it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_multi_agent_reinforcement_learning_amd import ops
from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
tr = Trainer(baseline_config(name))
for _ in range(3):
    tr.iterate()
torch.cuda.synchronize()
acc = {True: [], False: []}
for i in range(12):
    ops.RELU_LINK = i % 2 == 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.iterate()
    t_host = 1e3 * (time.perf_counter() - t0)
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0)
    acc[ops.RELU_LINK].append((dt,) + tuple(tr.last_breakdown_ms()))
    print(f"iter {i}: link {ops.RELU_LINK}: {dt:7.1f} ms (host returned after {t_host:7.1f})  rollout / update {tuple(round(x, 1) for x in tr.last_breakdown_ms())}", flush=True)
for k, v in acc.items():
    n = len(v)
    print(f"{name} RELU_LINK={k}: iteration {sum(x[0] for x in v) / n:.2f} ms, rollout {sum(x[1] for x in v) / n:.2f}, update {sum(x[2] for x in v) / n:.2f}")
