"""Builds the in-tree HIP libraries for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")

# name -> (sources, extra flags).  -ffp-contract=off: the f64 simulator must not be FMA-contracted (see pe_env.hip).
LIBS = {
    "libpe_env.so": (["pe_env.hip", "pe_reset.cpp"], ["-ffp-contract=off", "-pthread"]),
    "libmappo_ops.so": (["mappo_ops.hip"], []),
    "libn2n_env.so": (["n2n_env.hip"], ["-ffp-contract=off", "-pthread"]),
    "libe3d_env.so": (["e3d_env.hip"], ["-ffp-contract=off", "-pthread"]),
    # host-only: the strided-output GEMM + epilogue on hipBLASLt (include/mappo_gemm.h)
    "libmappo_gemm.so": (["mappo_gemm.cpp"], ["-lhipblaslt"]),
}


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP libraries of this package cannot be built")


def lib_path(name):
    return os.path.join(PKG, name)


def needs_build(name):
    out = lib_path(name)
    if not os.path.exists(out):
        return True
    srcs = [os.path.join(CSRC, s) for s in LIBS[name][0]] + [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    return any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs if os.path.exists(s))


def build_lib(name, force=False, verbose=False):
    srcs, extra = LIBS[name]
    srcs = [os.path.join(CSRC, s) for s in srcs]
    if not all(os.path.exists(s) for s in srcs):
        return None
    if not force and not needs_build(name):
        return lib_path(name)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + INCLUDE, *extra,
           "-o", lib_path(name), *srcs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib_path(name)


def build_all(force=False, verbose=False):
    """the four libraries side by side (hipcc is one process per library; mappo_ops.hip alone takes ~2 min)"""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(LIBS)) as pool:
        return list(pool.map(lambda n: build_lib(n, force=force, verbose=verbose), LIBS))


if __name__ == "__main__":
    print(build_all(force=True, verbose=True))
