"""GPU (-m gpu): option families written after the round-1 GPU budget ended -- green on the host simulator against goldens
pinned to the reference, not yet run on hardware.  The file sorts after every hardware-validated test file and before the
shape sweeps (tests/test_zz_dwconv_shapes_gpu.py), so that with ``pytest -x`` a failure here hides nothing that was green before."""
import pytest

pytestmark = pytest.mark.gpu


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
