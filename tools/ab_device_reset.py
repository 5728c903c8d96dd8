"""A/B in one process: iteration time with the host resetter (prefetched behind the update) vs the on-device reset."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
for dev in (False, True, False, True):
    tr = Trainer(baseline_config(name, **{"runtime.device_reset": dev}))
    tr.iterate()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); tr.iterate(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print(f"device_reset={dev}: {min(ts):.1f} ms / iteration (min of 3), breakdown {tuple(round(x, 1) for x in tr.last_breakdown_ms())}", flush=True)
    del tr
    torch.cuda.empty_cache()
