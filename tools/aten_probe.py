"""Which torch (aten) operators still launch kernels inside one training iteration, and how much device time they take."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer

tr = Trainer(baseline_config(sys.argv[1] if len(sys.argv) > 1 else "cfg3"))
for _ in range(3):
    tr.iterate()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.iterate()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = e.self_cuda_time_total
    if t > 0 and e.key.startswith("aten::"):
        rows.append((t, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"aten operators with device time: {tot / 1e3:.2f} ms in one iteration")
for t, c, k, sh in rows[:45]:
    print(f"{t / 1e3:8.3f} ms {c:5d} x {k:34s} {sh}")
