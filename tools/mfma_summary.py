"""rocprofv3 outputs of tools/profile_gru.py -> profiles/r04_gru_mfma_counters.json: per hand-written MFMA kernel the average
duration (kernel trace), MFMA busy cycles, and MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE x #SIMDs) -- the
derived-counter formula rocprofv3 lists for MfmaUtil -- next to the algorithmic fp32 rate against the 157.3 TFLOP/s peak.
  python tools/mfma_summary.py <dir>"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
GROUPED = "--grouped" in sys.argv    # <dir> holds the passes of `tools/profile_gru.py --grouped`: 18 x 3280 + 2 x 3248 sequences per launch
ROWS = 18 * 3280 + 2 * 3248 if GROUPED else 3280
KERNELS = {"k_gru_seq_fwd2": 2.0 * 150 * ROWS * 128 * 384, "k_gru_seq_bwd2": 2.0 * 150 * ROWS * 384 * 128,
           "k_gru_seq_fwd_sb": 2.0 * 150 * ROWS * 128 * 384, "k_gru_seq_bwd_sb": 2.0 * 150 * ROWS * 384 * 128}
if GROUPED:
    KERNELS["k_sb_gemm_n128<12, 1,"] = 2.0 * 492000 * 384 * 128       # the update's 384-input product at mini-batch size
else:   # substring -> algorithmic fp32 flops per launch ("k_gru_cell(" so that it does not match k_gru_cell_sb)
    KERNELS.update({"k_gru_cell(": 2.0 * 32768 * 128 * 768, "k_gru_cell_sb": 2 * 2.0 * 32768 * 128 * 768, "k_wgrad<3, 1>": 2.0 * 492000 * 384 * 128,
                    "k_sb_wgrad<3, 1>": 2.0 * 492000 * 384 * 128, "k_sb_gemm_n128<4, 1,": 2.0 * 196608 * 128 * 128,
                    "k_sb_gemm_n128<12, 1,": 2.0 * 65536 * 384 * 128, "k_sb_gemm_n128<4, 3,": 2.0 * 492000 * 128 * 384})
cnt = {}
SUB = "grug" if GROUPED else "gru"    # the passes of tools/profile_all.sh: <dir>/gru_c, gru_t (single layer), grug_c, grug_t (grouped)
for f in glob.glob(os.path.join(d, SUB + "_c", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                a = cnt.setdefault(k, {}).setdefault(r["Counter_Name"], [0, 0.0])
                a[0] += 1; a[1] += float(r["Counter_Value"])
dur = {}
for f in glob.glob(os.path.join(d, SUB + "_t", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        for k in KERNELS:
            if k in r["Name"]:
                dur[k] = float(r["AverageNs"]) * 1e-9
out = {"method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES (one pass) and "
                 "--kernel-trace --stats (second pass) of tools/profile_gru.py; MfmaUtil = MFMA busy cycles / (GRBM_GUI_ACTIVE * 1024 SIMDs), "
                 "GRBM_GUI_ACTIVE as reported (summed over the 8 XCDs -> divided by 8)", "kernels": {}}
for k, fl in KERNELS.items():
    c = {n: v[1] / v[0] for n, v in cnt.get(k, {}).items()}
    e = {"counters_avg_per_launch": {n: round(v, 1) for n, v in c.items()}}
    if k in dur:
        e["avg_duration_us"] = round(dur[k] * 1e6, 1)
        e["algorithmic_TFLOPs"] = round(fl / dur[k] / 1e12, 2)
        e["frac_of_157.3_TFLOPs"] = round(fl / dur[k] / 157.3e12, 4)
        if "_sb" in k:   # the split-bf16 kernels run six bf16 MFMAs per product: their pipe peaks at 2500 / 6 TFLOP/s fp32-equivalent
            e["frac_of_bf16_pipe_div_6"] = round(fl / dur[k] / (2500e12 / 6), 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        e["MfmaUtil_percent"] = round(100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024), 2)
    tag = ""
    if GROUPED:
        tag = " (492 000 rows)" if "gemm" in k else " (grouped: 20 layers, 65 536 sequences, 4 096 workgroups per launch)"
    elif "k_sb_gemm_n128<12" in k:
        tag = " (65 536 rows, in place)"
    out["kernels"][k + tag] = e
path = os.path.join(ROOT, "profiles", "r04_gru_mfma_counters.json")
if GROUPED and os.path.exists(path):   # appended to the single-layer record
    base = json.load(open(path))
    base["kernels"].update(out["kernels"])
    out = base
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
