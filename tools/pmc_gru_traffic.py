"""HBM traffic of the grouped GRU recurrences and the weight-gradient kernels (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over
tools/profile_gru.py --grouped, separate passes) beside their algorithmic bytes -> profiles/r04_gru_traffic.json.
  (on the GPU box)  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/gru_pmc && mkdir -p $O &&
      rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 tools/profile_gru.py --grouped &&
      rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 tools/profile_gru.py --grouped
  (here)            python tools/pmc_gru_traffic.py gpurun_out/gru_pmc
gfx950 corrections as in tools/pmc_summary.py: both counters in KB, FETCH_SIZE counts half of wide reads -> x 2."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
T, rows = 150, 18 * 3280 + 2 * 3248                   # sequence rows of the 20 layers of one grouped launch
algo = {"k_gru_seq_fwd_sb": ("gi in, out + four saved gate planes out: 4 096 B per row and step", 4096 * rows * T),
        "k_gru_seq_bwd_sb": ("gates, h_prev, dout in, dgi + dnr out: 5 120 B per row and step", 5120 * rows * T),
        "k_gru_seq_fwd2": ("(runtime.matmul: fp32) the same bytes", 4096 * rows * T),
        "k_gru_seq_bwd2": ("(runtime.matmul: fp32) the same bytes", 5120 * rows * T)}
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].split("::")[-1].strip()
        if name.startswith("void "):
            name = name[5:]
        if name not in algo:
            continue
        a = acc.setdefault((name, r["Counter_Name"]), [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
out = []
for name, (what, ab) in algo.items():
    fe, wr = acc.get((name, "FETCH_SIZE")), acc.get((name, "WRITE_SIZE"))
    if not fe or not wr:
        continue
    fb, wb = fe[1] / fe[0] * 1024 * 2, wr[1] / wr[0] * 1024
    out.append({"kernel": name, "what": what, "launches_sampled": fe[0], "algorithmic_bytes": ab, "fetch_bytes_x2": int(fb), "write_bytes": int(wb),
                "traffic_over_algorithmic": round((fb + wb) / ab, 3)})
res = {"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/profile_gru.py --grouped (20 layers, 65 536 sequence rows, T = 150); KB units, FETCH_SIZE x 2 (gfx950)",
       "kernels": out}
json.dump(res, open(os.path.join(ROOT, "profiles", "r04_gru_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
