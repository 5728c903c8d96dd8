"""HBM traffic of the split GEMMs of the update (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/masked_probe.py, separate passes) beside
their algorithmic bytes -> profiles/r04_gemm_traffic.json.
  (on the GPU box)  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/gemm_pmc && mkdir -p $O &&
      rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 tools/masked_probe.py && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 tools/masked_probe.py
  (here)            python tools/pmc_gemm_traffic.py gpurun_out/gemm_pmc
gfx950 corrections as in tools/pmc_summary.py (MI355X_MICROARCH.md, HBM section): both counters in KB, FETCH_SIZE counts half of wide reads -> x 2."""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
R = 492000
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_sb_gemm_n128<(\d+), (\d+), (\d+)>", r["Kernel_Name"])
        if not m:
            continue
        a = acc.setdefault((tuple(int(x) for x in m.groups()), r["Counter_Name"]), [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
rows = []
for (kc, nt, opt) in sorted({k[0] for k in acc}):
    K, N = 32 * kc, 128 * nt
    fetch = acc.get(((kc, nt, opt), "FETCH_SIZE")); write = acc.get(((kc, nt, opt), "WRITE_SIZE"))
    if not fetch or not write:
        continue
    mask = (N // 8 if opt & 16 else 4 * N) if opt & 4 else 0
    algo = R * (4 * K + 4 * N + mask)
    fb, wb = fetch[1] / fetch[0] * 1024 * 2, write[1] / write[0] * 1024
    rows.append({"kernel": f"k_sb_gemm_n128<{kc}, {nt}, {opt}>", "what": f"{N} <- {K}" + (", relu' mask from sign bits" if opt & 16 else ", relu' mask from the activations" if opt & 4 else ""),
                 "rows": R, "launches_sampled": fetch[0], "algorithmic_bytes": algo, "fetch_bytes_x2": int(fb), "write_bytes": int(wb),
                 "traffic_over_algorithmic": round((fb + wb) / algo, 3)})
out = {"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/masked_probe.py; KB units, FETCH_SIZE x 2 (gfx950)", "kernels": rows}
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_gemm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
