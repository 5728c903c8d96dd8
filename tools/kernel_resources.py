"""Per-kernel register / scratch / LDS report of a HIP source (hipcc -Rpass-analysis=kernel-resource-usage), and optionally
the static instruction count per kernel from the gfx950 ISA.   python tools/kernel_resources.py pe_env.hip [-- extra flags]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
if not os.path.exists(src):
    src = os.path.join(ROOT, "distributed_multi_agent_reinforcement_learning_amd", "csrc", src)
extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else (["-ffp-contract=off"] if "pe_env" in src or "n2n" in src else [])
with tempfile.TemporaryDirectory() as td:
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I" + os.path.join(ROOT, "include"), *extra,
           "-Rpass-analysis=kernel-resource-usage", "-save-temps=obj", src, "-o", os.path.join(td, "x.o")]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=td).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: (?:\S+: )?\s*(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|VGPR Spill|SGPR Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\])[: ]+(\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:90]}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" ")[0] + ("Spill" if "Spill" in k else "")] = v
    isa = [f for f in os.listdir(td) if f.endswith(".s") and "gfx950" in f]
    counts = {}
    if isa:
        name = None
        for line in open(os.path.join(td, isa[0])):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                name = m.group(1); counts[name] = 0
            elif name and re.match(r"^\s+[sv]_|^\s+(ds|global|buffer|flat|scratch)_", line):
                counts[name] += 1
            if line.startswith("\t.end_amdhsa_kernel") or ".Lfunc_end" in line:
                name = None
    for r in rows:
        print(f"{r['name']:90s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>3s} SGPR {r.get('SGPRs','?'):>4s} scratch {r.get('ScratchSize','?'):>4s} "
              f"vspill {r.get('VGPRSpill','?'):>3s} sspill {r.get('SGPRSpill','?'):>3s} occ {r.get('Occupancy','?'):>2s}")
    if counts:
        print("static instruction counts:")
        for k, v in counts.items():
            d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:100]
            print(f"  {v:6d}  {d}")
