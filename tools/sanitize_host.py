"""CPU sanitizer job: the product's host code and the oracle's C under ASan / UBSan / TSan.

  python tools/sanitize_host.py            # builds into a temp dir, runs, exits non-zero on any report
1. csrc/pe_reset.cpp (multi-threaded host resetter, product code) + tools/sanitize/reset_driver.cpp
   with -fsanitize=address,undefined, then with -fsanitize=thread.
2. oracle/pe_env_oracle.c, oracle/n2n_oracle.c and oracle/e3d_oracle.c with -fsanitize=address,undefined; tests/test_oracle_env.py,
   tests/test_oracle_n2n.py and tests/test_oracle_e3d.py run against those builds (libasan preloaded into the interpreter).
GPU code cannot be sanitized on this pool (no GPU ASan, no XNACK); the HIP kernels are covered by the parity tests.
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "distributed_multi_agent_reinforcement_learning_amd", "csrc")
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
           "TSAN_OPTIONS": "halt_on_error=1"}


def run(cmd, **kw):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, **kw)


def main():
    tmp = tempfile.mkdtemp(prefix="dmarl_san_")
    env = dict(os.environ, **SAN_ENV)
    src = [os.path.join(CSRC, "pe_reset.cpp"), os.path.join(ROOT, "tools", "sanitize", "reset_driver.cpp")]
    common = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    for name, flags in (("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"]), ("tsan", ["-fsanitize=thread"])):
        exe = os.path.join(tmp, "reset_driver_" + name)
        run(common + flags + src + ["-o", exe])
        run([exe], env=env)
    libs = {}
    for stem in ("pe_env_oracle", "n2n_oracle", "e3d_oracle"):
        out = os.path.join(tmp, f"lib{stem}_asan.so")
        run(["gcc", "-O1", "-g", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-fno-omit-frame-pointer",
             "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-shared", "-o", out, os.path.join(ROOT, "oracle", stem + ".c"), "-lm"])
        libs[stem] = out
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    ubsan_rt = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    tenv = dict(env, LD_PRELOAD=f"{asan_rt}:{ubsan_rt}", DMARL_PE_ORACLE_LIB=libs["pe_env_oracle"], DMARL_N2N_ORACLE_LIB=libs["n2n_oracle"], DMARL_E3D_ORACLE_LIB=libs["e3d_oracle"])
    run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_env.py"),
         os.path.join(ROOT, "tests", "test_oracle_n2n.py"), os.path.join(ROOT, "tests", "test_oracle_e3d.py")], env=tenv, cwd=ROOT)
    print("sanitize_host: clean (ASan + UBSan + TSan on the host resetter, ASan + UBSan on the oracle C)")


if __name__ == "__main__":
    main()
