"""Build the native library in-tree.

`python -m slowfast_amd.build_ext` compiles slowfast_amd/csrc/sf_api.hip for gfx950 with hipcc into
slowfast_amd/libsfamd.so (cross-compiles without a GPU).  `--hostsim` additionally builds the host
functional simulator used by the CPU test-suite (tests/hostsim/libsfamd_sim.so) from the SAME sources.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "slowfast_amd", "csrc")
SRC = os.path.join(CSRC, "sf_api.hip")
LIB = os.path.join(ROOT, "slowfast_amd", "libsfamd.so")
LIB_BF16 = os.path.join(ROOT, "slowfast_amd", "libsfamd_bf16.so")      # same sources, -DSF_ACT_BF16 (bfloat16 storage)
SIM_LIB = os.path.join(ROOT, "tests", "hostsim", "libsfamd_sim.so")
SIM_LIB_BF16 = os.path.join(ROOT, "tests", "hostsim", "libsfamd_sim_bf16.so")


def _sources():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(ROOT, "include", "sfamd.h"))
    return deps


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False, act="fp16"):
    """act = "fp16" -> libsfamd.so, "bf16" -> libsfamd_bf16.so (the 16-bit storage type, lib.ACT_MODE)."""
    out = LIB if act == "fp16" else LIB_BF16
    if not force and not _stale(out, _sources()):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-shared",
           "-Wno-comment", "-I" + os.path.join(ROOT, "include")] + (["-DSF_ACT_BF16"] if act == "bf16" else []) + [SRC, "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_hostsim(force=False, verbose=False, act="fp16"):
    out = SIM_LIB if act == "fp16" else SIM_LIB_BF16
    deps = _sources() + [os.path.join(ROOT, "tests", "hostsim", "include", "hip", "hip_runtime.h")]
    if not force and not _stale(out, deps):
        return out
    cxx = os.environ.get("SF_HOST_CXX", "/opt/rocm/lib/llvm/bin/clang++")
    cmd = [cxx, "-x", "c++", "-std=c++20", "-O2", "-fPIC", "-shared", "-Wno-unknown-attributes", "-Wno-comment",
           "-I" + os.path.join(ROOT, "tests", "hostsim", "include"), "-I" + os.path.join(ROOT, "include")] + \
          (["-DSF_ACT_BF16"] if act == "bf16" else []) + [SRC, "-o", out, "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    force = "--force" in sys.argv
    for act in ("fp16", "bf16"):
        print(build_hip(force=force, verbose=True, act=act))
        if "--hostsim" in sys.argv:
            print(build_hostsim(force=force, verbose=True, act=act))
