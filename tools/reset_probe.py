"""Host reset of 4096 environments (csrc/pe_reset.cpp): wall time vs threads (it runs behind the PPO update)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
from distributed_multi_agent_reinforcement_learning_amd import pe_env
cfg = baseline_config("cfg2")
pc = pe_env.make_pe_config(cfg, tape_len=16, max_path=128)
print("cpu_count", os.cpu_count())
for th in (1, 2, 4, 8, 16):
    r = pe_env.HostResetter(pc, cfg, list(range(4096)), n_threads=th)
    t0 = time.perf_counter(); r.reset(None); t1 = time.perf_counter()
    cons = np.zeros(4096, np.int32)
    r.reset(cons); t2 = time.perf_counter()
    print(f"threads {th:2d}: first reset {1e3*(t1-t0):7.1f} ms, next reset {1e3*(t2-t1):7.1f} ms")

try:
    import torch
    if torch.cuda.is_available():
        from distributed_multi_agent_reinforcement_learning_amd.pursuit_env import Pursuit_Env
        for name in ("cfg2",):
            env = Pursuit_Env(baseline_config(name, **{"runtime.device_reset": True}), num_envs=4096)
            env.reset(); torch.cuda.synchronize()
            for rep in range(3):
                t0 = time.perf_counter(); env.reset(); torch.cuda.synchronize()
                print(f"device reset of 4096 envs ({name}): {1e3*(time.perf_counter()-t0):.2f} ms")
            envh = Pursuit_Env(baseline_config(name), num_envs=4096)
            envh.reset(); torch.cuda.synchronize()
            t0 = time.perf_counter(); envh.reset(); torch.cuda.synchronize()
            print(f"host reset + upload of 4096 envs ({name}): {1e3*(time.perf_counter()-t0):.2f} ms")
except ImportError:
    pass
