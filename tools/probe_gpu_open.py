"""Which steps of a CPU-only worker open the GPU device nodes (the box's process guard counts processes holding them)?"""
import os, sys
def fds(tag):
    out = []
    for f in os.listdir("/proc/self/fd"):
        try:
            t = os.readlink(f"/proc/self/fd/{f}")
        except OSError:
            continue
        if "kfd" in t or "dri" in t:
            out.append(t)
    print(tag, sorted(set(out)), flush=True)
fds("start")
import numpy
fds("numpy")
import torch
fds("import torch")
torch.set_num_threads(1); torch.manual_seed(0)
fds("seed")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
from distributed_multi_agent_reinforcement_learning_amd.model import build_actor_critic
fds("package import")
cfg = baseline_config("cfg2")
a, c = build_actor_critic(cfg, "cpu")
fds("build model")
from oracle import model_oracle, pe_oracle, reset_oracle
pe_oracle.lib()
fds("oracle")
x = torch.multinomial(torch.ones(4, 9) / 9, 1)
fds("multinomial")
w = torch.nn.Parameter(torch.ones(3)); (w * 2).sum().backward()
fds("first backward()")
