"""Does the first backward() of a CPU-only process open the GPU device node?  (run with and without LD_PRELOAD=oracle/libnogpu_shim.so)"""
import os, torch
w = torch.nn.Parameter(torch.ones(3)); (w * 2).sum().backward()
out = []
for f in os.listdir("/proc/self/fd"):
    try:
        t = os.readlink(f"/proc/self/fd/{f}")
    except OSError:
        continue
    if "kfd" in t or "dri" in t:
        out.append(t)
print("GPU device nodes open after backward():", sorted(set(out)))
