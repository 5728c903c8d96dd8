"""Tunes the library GEMMs (rocBLAS / hipBLASLt solution selection through PyTorch TunableOp) for the shapes of the
benchmark configurations and writes distributed_multi_agent_reinforcement_learning_amd/tunableop_gfx950.csv.
The trainer loads that file with tuning disabled (trainer.enable_tuned_gemms); run this once per torch/ROCm version:
    python tools/tune_gemms.py [--configs cfg2 cfg3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.cuda.tunable as tunable  # noqa: E402

from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config  # noqa: E402
from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer, TUNED_GEMM_FILE  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--configs", nargs="+", default=["cfg2", "cfg3"])
ap.add_argument("--max-ms", type=int, default=8)
args = ap.parse_args()
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_max_tuning_duration(args.max_ms)
tunable.set_max_tuning_iterations(6)
tunable.set_filename(TUNED_GEMM_FILE, insert_device_ordinal=False)
if os.path.exists(TUNED_GEMM_FILE):
    tunable.read_file(TUNED_GEMM_FILE)
import threading


def heartbeat():
    t0 = time.time()
    while True:
        time.sleep(45)
        print(f"[tune] {time.time() - t0:.0f}s elapsed, {len(tunable.get_results())} GEMM shapes tuned so far", flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
for name in args.configs:
    cfg = baseline_config(name)
    t0 = time.time()
    tr = Trainer(cfg, tuned_gemms=False)
    tr.iterate()
    torch.cuda.synchronize()
    print(f"{name}: tuned in {time.time() - t0:.0f}s, {len(tunable.get_results())} entries", flush=True)
    pass  # TunableOp writes the file at interpreter exit (write_file_on_exit)
    del tr
    torch.cuda.empty_cache()
print("wrote", TUNED_GEMM_FILE)
