"""A/B on one box: update with the critic branch on a second stream vs single stream (same process, alternating)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer
cfg = baseline_config(sys.argv[1] if len(sys.argv) > 1 else "cfg2")
tr = Trainer(cfg)
tr.iterate()
for rep in range(3):
    for ov in (True, False):
        tr.agent.overlap_actor_critic = ov
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.iterate(); torch.cuda.synchronize()
        print(f"overlap={ov}: {1e3*(time.perf_counter()-t0):.1f} ms, breakdown {tr.last_breakdown_ms()}", flush=True)
