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


def up_to_date(lib=LIB):
    if not os.path.exists(lib):
        return False
    t = os.path.getmtime(lib)
    return all(os.path.getmtime(s) <= t for s in sources())


def build_lib(force=False, verbose=False, out=LIB, defines=()):
    if not force and up_to_date(out):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libhideseek.so (there is no CPU fallback)")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [hipcc] + HIPCC_FLAGS + ["-D" + d for d in defines] + (
        ["-Rpass-analysis=kernel-resource-usage"] if verbose else []) + ["-o", out, os.path.join(CSRC, "hideseek.hip")]
    subprocess.check_call(cmd, cwd=CSRC)
    return out


# Test-only build with tiny broadphase capacities, so that the overflow counters of hs_get_device_status can be seen
# to count (tests/test_gpu_status.py loads it through HS_LIB_PATH in a child process).
SMALLCAP_LIB = os.path.join(HERE, "lib", "libhideseek_smallcap.so")


def build_smallcap(force=False):
    return build_lib(force=force, out=SMALLCAP_LIB, defines=("HS_MAX_DD_CAND=1", "HS_MAX_S_CAND=1"))


# Development build with per-phase timers in the kernels (tools/phase_timing.py, tools/phase_tail.py).
TIMING_LIB = os.path.join(HERE, "lib", "libhideseek_timing.so")


def build_timing(force=False, counters=False):
    return build_lib(force=force, out=TIMING_LIB, defines=("HS_PHASE_TIMING",) + (("HS_SAT_COUNTERS",) if counters else ()))


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
    print(build_smallcap(force="--force" in sys.argv))
    if "--timing" in sys.argv:
        print(build_timing(force="--force" in sys.argv, counters="--sat-counters" in sys.argv))
