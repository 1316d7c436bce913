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
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    deps.append(os.path.join(ROOT, "include", "sfamd.h"))
    return deps


SIM_SHIM = os.path.join(ROOT, "tests", "hostsim", "include", "hip", "hip_runtime.h")


def sources_present():
    """False in a deployment that ships the binaries without csrc/ (nothing to compare a build id against)."""
    return os.path.isfile(SRC) and os.path.isfile(os.path.join(ROOT, "include", "sfamd.h"))


def source_id(sim=False):
    """sha256 (first 16 hex digits) over the sources of ONE target, in name order: the kernel sources + the C header for the
    gfx950 libraries; the host simulator's HIP shim on top of them for the simulator builds only (``sim=True``) -- an edit of
    the test shim must not make the GPU library unloadable.  Compiled into every build as sf_build_id() and recomputed by
    lib.SfLibrary from the files shipped beside the binary.  A listed file that is missing raises (a silent skip would hash a
    different set of files into the same id)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(_sources() + ([SIM_SHIM] if sim else [])):
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _built_id(target):
    """Build id embedded in a binary ('' when it has none / cannot be read) without loading it into this process."""
    try:
        with open(target, "rb") as fh:
            blob = fh.read()
    except OSError:
        return ""
    tag = b"sfamd-build-id:"
    i = blob.find(tag)
    return blob[i + len(tag):i + len(tag) + 16].decode("ascii", "replace") if i >= 0 else ""


def _stale(target, sim=False):
    """A binary is current when it carries the hash of today's sources of its target (mtimes are not trusted: a checkout or a
    copy to the GPU box resets them)."""
    return not os.path.exists(target) or _built_id(target) != source_id(sim)


LIB_DIAG = os.path.join(ROOT, "slowfast_amd", "libsfamd_diag.so")    # -DSF_DIAG: tuning knobs / ablation bits read the environment


def build_hip(force=False, verbose=False, act="fp16", diag=False):
    """act = "fp16" -> libsfamd.so, "bf16" -> libsfamd_bf16.so (the 16-bit storage type, lib.ACT_MODE).  ``diag``: the fp16
    library compiled with -DSF_DIAG (libsfamd_diag.so; point SFAMD_LIBRARY at it): A/B knobs and the kernels' ablation
    switches are live there, compile-time constants everywhere else (csrc/sf_api.hip: tune_knob)."""
    if diag and act != "fp16":
        raise ValueError("build_hip: the diagnostic build exists for float16 storage only (diag=True with act=%r)" % (act,))
    out = LIB_DIAG if diag else (LIB if act == "fp16" else LIB_BF16)
    if not force and not _stale(out):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-shared",
           "-Wno-comment", "-I" + os.path.join(ROOT, "include"), '-DSF_BUILD_ID="%s"' % source_id()] + \
          (["-DSF_ACT_BF16"] if act == "bf16" and not diag else []) + (["-DSF_DIAG"] if diag else []) + [SRC, "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_hostsim(force=False, verbose=False, act="fp16"):
    out = SIM_LIB if act == "fp16" else SIM_LIB_BF16
    if not force and not _stale(out, sim=True):
        return out
    cxx = os.environ.get("SF_HOST_CXX", "/opt/rocm/lib/llvm/bin/clang++")
    cmd = [cxx, "-x", "c++", "-std=c++20", "-O2", "-fPIC", "-shared", "-Wno-unknown-attributes", "-Wno-comment",
           "-I" + os.path.join(ROOT, "tests", "hostsim", "include"), "-I" + os.path.join(ROOT, "include"),
           '-DSF_BUILD_ID="%s"' % source_id(sim=True)] + \
          (["-DSF_ACT_BF16"] if act == "bf16" else []) + [SRC, "-o", out, "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    force = "--force" in sys.argv
    if "--diag" in sys.argv:
        print(build_hip(force=force, verbose=True, diag=True))
        sys.exit(0)
    for act in ("fp16", "bf16"):
        print(build_hip(force=force, verbose=True, act=act))
        if "--hostsim" in sys.argv:
            print(build_hostsim(force=force, verbose=True, act=act))
