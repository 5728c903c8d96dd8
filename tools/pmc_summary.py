"""rocprofv3 --pmc csv outputs -> profiles/r04_tick_pmc.json (packed LiDAR rows: the launch the rollout issues) or, with
--layout fp32, profiles/r04_tick_pmc_fp32.json (profile_tick.py --float-obs: the reference's fp32 rows); bench.py attaches them
as roofline_packed_rows.traffic / roofline.traffic.

  python tools/pmc_summary.py <dir-with-counter-csvs> [--us-per-launch X] [--num-envs N] [--layout packed|fp32]
Looks for *counter_collection.csv files (one rocprofv3 pass each; FETCH_SIZE and WRITE_SIZE need separate passes on gfx950),
averages every counter over the launches of the regular fused tick kernel k_tick<true, true, true, false, ...> and applies the
gfx950 corrections of MI355X_MICROARCH.md (HBM section): FETCH_SIZE is reported in KB and counts HALF of the bytes of wide
coalesced reads -> x2; WRITE_SIZE (KB) as read.  The kernel-source hash ties the record to csrc/pe_env.hip + include/pe_env.h.
"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import algorithmic_bytes_per_env_step, tick_kernel_hash  # noqa: E402

d = sys.argv[1]
us = float(sys.argv[sys.argv.index("--us-per-launch") + 1]) if "--us-per-launch" in sys.argv else None
N = int(sys.argv[sys.argv.index("--num-envs") + 1]) if "--num-envs" in sys.argv else 4096
layout = sys.argv[sys.argv.index("--layout") + 1] if "--layout" in sys.argv else "packed"
passes = sys.argv[sys.argv.index("--passes") + 1].split(",") if "--passes" in sys.argv else None   # sub-directories of <dir> to read
acc = {}
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
if passes:
    files = [f for f in files if os.path.relpath(f, d).split(os.sep)[0] in passes]
for f in files:
    for r in csv.DictReader(open(f)):
        if "k_tick<true, true, true, false" not in r["Kernel_Name"]:
            continue
        a = acc.setdefault(r["Counter_Name"], [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
avg = {k: v[1] / v[0] for k, v in acc.items()}
n = {k: v[0] for k, v in acc.items()}
out = {"num_envs": N, "layout": layout,
       "config": "cfg2 (P=8, 40x40, O=176), " + ("packed LiDAR rows (the launch the rollout issues)" if layout == "packed" else "fp32 (N, P, O) LiDAR rows (reference layout)"),
       "kernel": "k_tick<true, true, true, false, true>", "kernel_hash": tick_kernel_hash(), "launches_sampled": n,
       "counters_avg_per_launch": {k: round(v, 2) for k, v in sorted(avg.items())},
       "algorithmic_bytes_per_launch": N * algorithmic_bytes_per_env_step(8, 40, 40, 176), "us_per_launch": us,
       "method": "separate rocprofv3 --pmc passes of tools/profile_tick.py; FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE doubled (gfx950 "
                 "reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as read"}
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    out["fetch_bytes_corrected_x2"] = int(avg["FETCH_SIZE"] * 1024 * 2)
    out["write_bytes"] = int(avg["WRITE_SIZE"] * 1024)
    out["traffic_bytes_per_launch"] = out["fetch_bytes_corrected_x2"] + out["write_bytes"]
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_tick_pmc.json" if layout == "packed" else "r04_tick_pmc_fp32.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
