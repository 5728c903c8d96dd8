"""Wall time of consecutive iterations + caching-allocator activity (device mallocs between iterations mean stalls)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
tr = Trainer(baseline_config(name))
for i in range(7):
    st0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.iterate(); torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0)
    st1 = torch.cuda.memory_stats()
    print(f"iter {i}: {dt:7.1f} ms  events {tuple(round(x,1) for x in tr.last_breakdown_ms())}  segments +{st1['segment.all.allocated']-st0['segment.all.allocated']} "
          f"-{st1['segment.all.freed']-st0['segment.all.freed']}  reserved {st1['reserved_bytes.all.current']/2**30:.1f} GiB  peak alloc {st1['allocated_bytes.all.peak']/2**30:.1f} GiB  retries {st1['num_alloc_retries']}", flush=True)
