"""Copies what tools/profile_all.sh left under gpurun_out/r4final into profiles/ (the newest file of every pass), regenerates the
derived records (r04_tick_pmc.json, r04_tick_pmc_fp32.json, r04_gru_mfma_counters.json) and prints the per-category GPU time of the
bench profiles.
  python tools/collect_profiles.py [gpurun_out/r4final]"""
import csv, glob, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r4final")
P = os.path.join(ROOT, "profiles")


def newest(pattern):
    files = sorted(glob.glob(os.path.join(R, pattern), recursive=True), key=os.path.getmtime)
    if not files:
        raise SystemExit(f"nothing matches {pattern} under {R}")
    return files[-1]


for sub, dst in (("bench_cfg2", "r04_bench_cfg2_4096env_kernel_stats.csv"), ("bench_cfg3", "r04_bench_cfg3_4096env_kernel_stats.csv"),
                 ("tick", "r04_tick_only_4096env_kernel_stats.csv"), ("tick_f32", "r04_tick_only_fp32_rows_4096env_kernel_stats.csv"),
                 ("gru_t", "r04_gru_kernels_kernel_stats.csv")):
    src = newest(f"{sub}/**/*kernel_stats.csv")
    shutil.copy(src, os.path.join(P, dst))
    print(dst, "<-", os.path.relpath(src, ROOT))
shutil.copy(os.path.join(R, "envs.log"), os.path.join(P, "r04_envs_n2n_e3d.jsonl"))
rows = list(csv.DictReader(open(os.path.join(P, "r04_tick_only_4096env_kernel_stats.csv"))))
us = [float(r["AverageNs"]) / 1e3 for r in rows if "k_tick<true, true, true, false" in r["Name"]][0]
print(f"regular tick: {us:.2f} us per launch")
rows = list(csv.DictReader(open(os.path.join(P, "r04_tick_only_fp32_rows_4096env_kernel_stats.csv"))))
us32 = [float(r["AverageNs"]) / 1e3 for r in rows if "k_tick<true, true, true, false" in r["Name"]][0]
print(f"regular tick, fp32 LiDAR rows: {us32:.2f} us per launch")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), R, "--us-per-launch", f"{us:.2f}", "--passes", "pmc1,pmc2,pmc3"],
                      stdout=subprocess.DEVNULL)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), R, "--us-per-launch", f"{us32:.2f}", "--passes", "pmcf1,pmcf2",
                       "--layout", "fp32"], stdout=subprocess.DEVNULL)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "mfma_summary.py"), R], stdout=subprocess.DEVNULL)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "mfma_summary.py"), R, "--grouped"], stdout=subprocess.DEVNULL)
shutil.copy(newest("grug_t/**/*kernel_stats.csv"), os.path.join(P, "r04_gru_kernels_grouped_kernel_stats.csv"))
for cfg in ("cfg2", "cfg3"):
    cat = {}
    for r in csv.DictReader(open(os.path.join(P, f"r04_bench_{cfg}_4096env_kernel_stats.csv"))):
        n, t = r["Name"], float(r["TotalDurationNs"]) / 1e6 / 4   # 2 set-up + 2 timed iterations in the profiled run
        key = ("gru_cell" if "k_gru_cell" in n else "gru_seq" if "k_gru_seq" in n or "k_gru_bias" in n else "skinny" if "skinny" in n
               else "wgrad" if "k_wgrad" in n or "k_sb_wgrad" in n else "split_gemm" if "k_sb_gemm" in n else "msg" if "k_msg" in n
               else "lib_gemm" if "Cijk" in n or "rocblas" in n
               else "env" if "k_tick" in n or "k_build" in n or "k_reset" in n
               else "aten" if "at::" in n or "elementwise" in n or "reduce_kernel" in n or "rocclr" in n else "small_own")
        cat[key] = cat.get(key, 0.0) + t
    print(cfg, {k: round(v, 1) for k, v in sorted(cat.items(), key=lambda x: -x[1])}, "total", round(sum(cat.values()), 1), "ms GPU time / iteration")
