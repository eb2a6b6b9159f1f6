"""Build libhideseek.so (HIP, gfx950) in-tree.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libhideseek.so")

# -ffp-contract=off / no fast-math: results are compared bit for bit with the CPU oracle.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-Wno-pass-failed", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fno-fast-math", "-Wno-unused-value"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + [
        os.path.join(os.path.dirname(HERE), "include", "hideseek.h")]


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(s) <= t for s in sources())


def build_lib(force=False, verbose=False):
    if not force and up_to_date():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libhideseek.so (there is no CPU fallback)")
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc] + HIPCC_FLAGS + (["-Rpass-analysis=kernel-resource-usage"] if verbose else []) + [
        "-o", LIB, os.path.join(CSRC, "hideseek.hip")]
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


def build_headless():
    """C++ headless driver over the C ABI (reference: src/headless.cpp)."""
    out = os.path.join(HERE, "lib", "headless")
    src = os.path.join(HERE, "tools", "headless.cpp")
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(LIB)):
        return out
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(os.path.dirname(HERE), "include"), src,
                           "-L", os.path.join(HERE, "lib"), "-lhideseek", "-Wl,-rpath,$ORIGIN", "-o", out])
    return out


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose="-v" in sys.argv))
    print(build_headless())
