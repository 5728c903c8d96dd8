"""Rollout only (no update): target for rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer
cfg = baseline_config(sys.argv[1] if len(sys.argv) > 1 else "cfg2")
tr = Trainer(cfg)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    tr.agent.explore_env(tr.env, 1)
    torch.cuda.synchronize(); print(f"rollout {it}: {(time.time()-t0)*1e3:.1f} ms", flush=True)
