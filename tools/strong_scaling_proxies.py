"""Single-GPU proxies of strong scaling: `bench.py --num-envs n` for the per-rank share of 4096 environments on 1 / 2 / 4 / 8 GPUs (no
collective in the proxy: the 2.4 MB all-reduce is ~30 us over xGMI).  One JSON line per (config, n) -> profiles/r04_strong_scaling_proxies.jsonl
  python tools/strong_scaling_proxies.py [out.jsonl]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "strong_scaling_proxies.jsonl")
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "w") as f:
    for cfg in ("cfg2", "cfg3"):
        base = None
        for n in (4096, 2048, 1024, 512):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--num-envs", str(n), "--steps", "6", "--warmup", "2",
                                "--no-cpu-baseline", "--no-secondary", "--no-kernel-probes"], capture_output=True, text=True, timeout=400)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode or not line:
                print(r.stderr[-2000:]); raise SystemExit(1)
            j = json.loads(line[-1])
            base = base or j["ms_per_step"]
            row = dict(config=cfg, envs_per_rank=n, ranks_of_4096=4096 // n, ms_per_step=j["ms_per_step"], breakdown_ms=j["breakdown_ms"],
                       vs_4096=round(j["ms_per_step"] / base, 3), whole_job_env_steps_per_s=round(4096 * 150 / (j["ms_per_step"] * 1e-3)))
            f.write(json.dumps(row) + "\n"); f.flush()
            print(row, flush=True)
