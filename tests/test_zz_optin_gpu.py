"""GPU (-m gpu), opt-in code paths and option families written after the round's GPU budget ended (green on the host
simulator, not yet run on hardware): kept in a file that sorts last, so that with ``pytest -x`` a failure here cannot hide
the results of the default-path tests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_depthwise_version2_kernels_gpu(gpu):
    """The version-2 depthwise stencils (SF_DW_*_V2=1, profiles/r1_isa_dwconv_v2.md) on the real kernels: narrow (fp32 LDS
    weights) and wide (fp16) layers, stride 1 and 2, whole and ragged 4-column groups, cls rows; then X3D end to end."""
    code = ("import torch; from tests import token_checks as tc, model_checks as mc; d=torch.device('cuda:0');"
            "tc.check_dwconv(d,2,2,96,(4,14,14),(3,3,3),(1,2,2),cls=1);"
            "tc.check_dwconv(d,1,4,96,(4,7,7),(3,3,3),(1,1,1),cls=1);"
            "tc.check_dwconv(d,2,1,24,(8,12,12),(5,1,1),(1,1,1),cls=0);"
            "tc.check_dwconv(d,2,1,216,(4,14,14),(3,3,3),(1,2,2),cls=0);"
            "tc.check_dwconv(d,2,1,56,(4,28,28),(3,3,3),(1,1,1),cls=0);"
            "tc.check_dwconv(d,2,1,56,(4,56,56),(3,3,3),(1,2,2),cls=0);"
            "tc.check_dwconv(d,1,1,432,(4,8,8),(3,3,3),(1,1,1),cls=0);"
            "mc.check_engine('x3d_tiny', d, loss_scale=1.0, tol_logits=4e-3, tol_loss=1e-3, tol_gnorm=2e-3, tol_param=0.1,"
            " tol_global=1e-2); print('ok')")
    env = dict(os.environ, SF_DW_FWD_V2="1", SF_DW_DGRAD_V2="1", SF_DW_WGRAD_V2="1")
    env.pop("SFAMD_LIBRARY", None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("name", ["mvit_nocls_sepqkv_tiny", "mvit_poolfirst_tiny", "mvit_relinterp_tiny"])
@pytest.mark.parametrize("fused_attn", ["1", "0"])
def test_mvit_attention_options_match_reference_gpu(gpu, name, fused_attn, monkeypatch):
    """CLS_EMBED_ON False + SEPARATE_QKV, and POOL_FIRST, vs the reference's outputs (tests/golden)."""
    from tests import model_checks as mc
    monkeypatch.setenv("SF_ATTN_FUSED", fused_attn)
    rep = {}
    try:
        mc.check_engine(name, gpu, tol_logits=4e-3, tol_loss=1e-3, tol_gnorm=2e-3, tol_param=0.2, tol_global=1e-2, report=rep)
    finally:
        print(name, fused_attn, rep)


def test_wide_pooling_kernels_gpu(gpu):
    """Depthwise pooling kernels with more than 9 taps per plane (MViTv1 stride+1 kernels 1x5x5 / 1x9x9,
    configs/Kinetics/MVIT_B_32x3_CONV.yaml): generic forward / data gradient, chunked weight gradient."""
    from tests import token_checks as tc
    tc.check_dwconv(gpu, 2, 2, 96, (4, 28, 28), (1, 5, 5), (1, 4, 4), cls=1)
    tc.check_dwconv(gpu, 2, 1, 96, (4, 56, 56), (1, 9, 9), (1, 8, 8), cls=1)


def test_reversible_mvit_gpu(gpu):
    """Reversible MViT vs the reference's outputs (tests/golden/mvit_rev_tiny.json), then with pinned stochastic depth."""
    from tests import model_checks as mc
    rep = {}
    try:
        mc.check_engine("mvit_rev_tiny", gpu, tol_logits=4e-3, tol_loss=1e-3, tol_gnorm=2e-3, tol_param=0.2, tol_global=1e-2,
                        report=rep)
    finally:
        print(rep)
    print(mc.check_rev_mvit_drop_path(gpu))


def test_basic_transform_gpu(gpu):
    """RESNET.TRANS_FUNC basic_transform (Tx3x3 -> 1x3x3 blocks) vs the reference: training step, then the eval path
    running-statistics and inference-fused."""
    from tests import model_checks as mc
    rep = {}
    try:
        mc.check_engine("i3d_basic_tiny", gpu, tol_logits=4e-3, tol_loss=1e-3, tol_gnorm=2e-3, tol_param=0.1, tol_global=1e-2,
                        report=rep)
        mc.check_eval("eval_i3d_basic_tiny", gpu, fused=False, report=rep)
        mc.check_eval("eval_i3d_basic_tiny", gpu, fused=True, report=rep)
    finally:
        print(rep)


def test_x3d_bn_lin5_gpu(gpu):
    from tests import model_checks as mc
    print(mc.check_engine("x3d_bnlin5_tiny", gpu, loss_scale=1.0, tol_logits=4e-3, tol_loss=1e-3, tol_gnorm=2e-3, tol_param=0.1,
                          tol_global=1e-2))


def test_mvit_detection_gpu(gpu):
    from tests import model_checks as mc
    print(mc.check_engine("mvit_ava_roi_tiny", gpu, tol_logits=4e-3, tol_loss=1e-3, tol_gnorm=2e-3, tol_param=0.2,
                          tol_global=1e-2))
