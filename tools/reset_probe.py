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
