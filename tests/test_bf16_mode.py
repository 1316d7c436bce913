"""bfloat16 storage mode (SF_ACT_DTYPE=bf16 -> libsfamd_bf16.so: the same kernel sources compiled with -DSF_ACT_BF16, activations /
packed weights / MFMA operands in bfloat16, v_mfma_f32_16x16x32_bf16, fp32 everywhere else).

The reference's mixed-precision path is torch.cuda.amp.autocast (tools/train_net.py:101-118), which admits float16 and bfloat16;
the north star names both.  The storage type is fixed per PROCESS (slowfast_amd.lib.ACT_MODE), so these tests re-run the existing
kernel / block / model checks in a child process with SF_ACT_DTYPE=bf16: every tolerance of those checks is written in units of
the storage type's epsilon (tests/kernel_checks.F16_EPS = 2^-10 | 2^-7, EPS_SCALE = 1 | 8) and the oracle's storage-model
yardstick rounds to the same type.  Model level: the ``*_wc`` cases against the fp32 oracle at max(8 x the fp16 bar, 1.5 x what
the reference graph loses under torch.autocast(bfloat16) on an MI355X -- tests/golden/autocast_yardstick_bf16.json)."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_pytest_bf16(args, timeout, extra_env=None):
    env = dict(os.environ, SF_ACT_DTYPE="bf16")
    env.pop("SFAMD_LIBRARY", None)
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert " passed" in tail, tail
    return tail


def test_bf16_builds_identify_themselves():
    """Both bf16 builds export the whole C ABI (lib._SIGNATURES is checked symbol by symbol on load) and report bfloat16;
    a process in one mode refuses the other mode's library."""
    from slowfast_amd import build_ext, lib
    for path in (build_ext.build_hip(act="bf16"), build_ext.build_hostsim(act="bf16")):
        cdll = ctypes.CDLL(path)
        assert cdll.sf_act_dtype() == 1 and cdll.sf_abi_version() == lib.ABI_VERSION
        for name in lib._SIGNATURES:
            getattr(cdll, name)
    assert ctypes.CDLL(build_ext.build_hip()).sf_act_dtype() == 0
    if lib.ACT_MODE == "fp16":
        with pytest.raises(lib.SfError, match="bf16 build"):
            lib.SfLibrary(build_ext.SIM_LIB_BF16)


def test_bf16_kernels_and_blocks_hostsim():
    """Kernel, token-kernel, implicit-GEMM-variant and block-level checks on the host simulator's bf16 build."""
    _run_pytest_bf16(["tests/test_kernels_hostsim.py", "tests/test_tokens_hostsim.py", "tests/test_igemm2_hostsim.py",
                      "tests/test_blocks_hostsim.py", "-m", "not gpu"], timeout=1500)


def test_bf16_models_hostsim():
    """Whole drop-in models (ResNet / SlowFast / X3D / MViT wiring cases, eval path) in bf16 storage on the host simulator."""
    _run_pytest_bf16(["tests/test_model_hostsim.py", "-m", "not gpu", "-k",
                      "slowfast_tiny or mvit_tiny or x3d_tiny"], timeout=2400)


@pytest.mark.gpu
def test_bf16_kernels_on_gpu():
    """Every kernel check of the GPU suite on libsfamd_bf16.so (tolerances in bf16 epsilons: 2 * 2^-7 forward / data gradient)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    print(_run_pytest_bf16(["tests/test_kernels_gpu.py", "tests/test_tokens_gpu.py", "-m", "gpu"], timeout=1500)[-300:])


@pytest.mark.gpu
def test_bf16_models_on_gpu():
    """Well-conditioned model cases in bf16 storage against the fp32 oracle: max(8e-3, 1.5 x reference under autocast(bfloat16));
    block-level strict checks; the MViT / X3D wiring cases."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    rep = os.environ.get("SF_PARITY_REPORT")
    extra = {"SF_PARITY_REPORT": rep[:-6] + "_bf16.jsonl"} if rep and rep.endswith(".jsonl") else {}
    print(_run_pytest_bf16(["tests/test_model_gpu.py", "-m", "gpu", "-k",
                            "c2d_wc or slowfast_wc or x3d_wc or blocks_strict or mvit_tiny"], timeout=2400, extra_env=extra)[-300:])
