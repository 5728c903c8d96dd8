"""Builds the in-tree HIP libraries for gfx950 with hipcc (cross-compiles without a GPU).

Every source is compiled to its own object (cached by modification time, the objects of all libraries side by side in a thread
pool) and the objects of a library are linked into the shared object: touching one kernel file recompiles that file only
(csrc/mappo_ops.hip alone takes ~2.5 min, csrc/mappo_split.hip ~1 min)."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, ".obj")
INCLUDE = os.path.join(ROOT, "include")

# name -> (sources, extra flags).  -ffp-contract=off: the f64 simulator must not be FMA-contracted (see pe_env.hip).
LIBS = {
    "libpe_env.so": (["pe_env.hip", "pe_reset.cpp"], ["-ffp-contract=off", "-pthread"]),
    # mappo_ops.hip: message / GAE / heads / fp32-MFMA GRU / weight-gradient kernels; mappo_split.hip: the split-bf16 GEMM and GRU
    # sequence kernels (csrc/sb_*.hpp) -- one library, two translation units
    "libmappo_ops.so": (["mappo_ops.hip", "mappo_split.hip"], []),
    "libn2n_env.so": (["n2n_env.hip"], ["-ffp-contract=off", "-pthread"]),
    "libe3d_env.so": (["e3d_env.hip"], ["-ffp-contract=off", "-pthread"]),
    # host-only: the strided-output GEMM + epilogue on hipBLASLt (include/mappo_gemm.h)
    "libmappo_gemm.so": (["mappo_gemm.cpp"], ["-lhipblaslt"]),
}
LINK_ONLY = {"-lhipblaslt", "-pthread"}


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP libraries of this package cannot be built")


def lib_path(name):
    return os.path.join(PKG, name)


def _headers():
    hs = [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    hs += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    return hs


def _obj_path(lib, src):
    return os.path.join(OBJ, f"{os.path.splitext(lib)[0]}__{src}.o")


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def needs_build(name):
    srcs = [os.path.join(CSRC, s) for s in LIBS[name][0]]
    return _stale(lib_path(name), srcs + _headers())


def _compile(lib, src, force, verbose):
    extra = [f for f in LIBS[lib][1] if f not in LINK_ONLY or f == "-pthread"]
    out, path = _obj_path(lib, src), os.path.join(CSRC, src)
    if not force and not _stale(out, [path] + _headers()):
        return out
    os.makedirs(OBJ, exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I" + INCLUDE, "-I" + CSRC, *extra, "-o", out, path]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def _link(lib, objs, verbose):
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path(lib), *objs, *[f for f in LIBS[lib][1] if f in LINK_ONLY]]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib_path(lib)


def build_lib(name, force=False, verbose=False):
    srcs = LIBS[name][0]
    if not all(os.path.exists(os.path.join(CSRC, s)) for s in srcs):
        return None
    if not force and not needs_build(name):
        return lib_path(name)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(srcs)) as pool:
        objs = list(pool.map(lambda s: _compile(name, s, force, verbose), srcs))
    return _link(name, objs, verbose)


def build_all(force=False, verbose=False):
    """all libraries side by side (hipcc is one process per source)"""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(LIBS)) as pool:
        return list(pool.map(lambda n: build_lib(n, force=force, verbose=verbose), LIBS))


if __name__ == "__main__":
    print(build_all(force=True, verbose=True))
