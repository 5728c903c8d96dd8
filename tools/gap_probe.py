"""GPU idle gaps of one training iteration (torch profiler timeline): where the device waits for the host."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer

tr = Trainer(baseline_config(sys.argv[1] if len(sys.argv) > 1 else "cfg2"))
for _ in range(3):
    tr.iterate()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.iterate()
    torch.cuda.synchronize()
evs = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA), key=lambda e: e.time_range.start)
t0, t1 = evs[0].time_range.start, max(e.time_range.end for e in evs)
busy_end = evs[0].time_range.end
gaps = []
for prev, e in zip(evs, evs[1:]):
    if e.time_range.start > busy_end:
        gaps.append((e.time_range.start - busy_end, busy_end - t0, prev.name[:60], e.name[:60]))
    busy_end = max(busy_end, e.time_range.end)
tot = sum(g[0] for g in gaps)
print(f"iteration span {(t1 - t0) / 1e3:.1f} ms, idle {tot / 1e3:.1f} ms in {len(gaps)} gaps")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"  {g[0]:8.1f} us idle at +{g[1] / 1e3:7.1f} ms  after {g[2]}  before {g[3]}")
small = sum(g[0] for g in gaps if g[0] < 20)
print(f"gaps < 20 us: {small / 1e3:.1f} ms in {sum(1 for g in gaps if g[0] < 20)} gaps")
