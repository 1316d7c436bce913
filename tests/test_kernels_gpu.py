"""GPU (-m gpu): parity of every libsfamd kernel on a real MI355X against the torch fp32 reference op."""
import os
import subprocess
import sys

import pytest

from tests import kernel_checks as kc
from tests.test_kernels_hostsim import CONV_CASES

pytestmark = pytest.mark.gpu

BIG_CASES = [
    ((4, 64, 4, 28, 28), 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
    ((2, 320, 4, 28, 28), 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
    ((2, 256, 4, 14, 14), 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    ((2, 128, 4, 28, 28), 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1)),
    ((2, 640, 8, 14, 14), 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),
    ((2, 32, 16, 28, 28), 64, (7, 1, 1), (4, 1, 1), (3, 0, 0), (1, 1, 1)),
    ((2, 8, 8, 56, 56), 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
    ((2, 320, 2, 28, 28), 512, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 1, 1)),
    ((1, 512, 4, 7, 7), 2048, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
    ((2, 64, 4, 14, 14), 64, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),
]


# deep contractions on many rows: the launcher takes the second-generation implicit GEMM (csrc/sf_igemm2.h: 256-row tiles,
# three-stage direct-to-LDS ring) -- many tiles per CU and tens of K steps, which is what exposes a pipelining race
IGEMM2_CASES = [
    ((8, 64, 4, 28, 28), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),       # 25088 rows, BN 64, 9 taps
    ((8, 256, 4, 14, 14), 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),     # 6272 rows, K 2304
    ((4, 1024, 4, 14, 14), 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),    # K 3072, temporal taps
    ((8, 128, 4, 28, 28), 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1)),     # stride 2 (forward)
    ((4, 2048, 4, 7, 7), 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),      # res5: 784 rows x K 6144 (below the row cut)
    ((16, 96, 2, 20, 20), 288, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),     # BK 32 (96 channels), ragged N
    ((8, 320, 2, 28, 28), 512, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 1, 1)),     # strided shortcut: dgrad = 4 residue classes
    ((2, 64, 16, 28, 28), 128, (7, 1, 1), (4, 1, 1), (3, 0, 0), (1, 1, 1)),     # lateral connection: temporal stride 4
]


# thin weight gradients (sf_wgrad2t_kernel: <= 32 output channels) at Fast-pathway sizes: hundreds of 128-row stages per
# workgroup, every tile / stage-count combination of the kernel
WGRAD2T_GPU_CASES = [
    ((4, 8, 16, 56, 56), 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),        # res2 b: 200k rows, BMW 16 x BKW 128 (K 72)
    ((4, 32, 16, 56, 56), 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),       # res2 a: K 96
    ((4, 8, 16, 56, 56), 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),       # res2 c: BMW 32 x BKW 32, three stages
    ((4, 16, 16, 28, 28), 16, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1)),      # res3 b first block: stride 2, K 144 (two k tiles)
    ((4, 128, 16, 14, 14), 32, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),     # res4 a: BMW 32 x BKW 128, K 384
    ((8, 16, 8, 28, 28), 16, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),       # BMW 16 x BKW 32
]


@pytest.mark.parametrize("case", WGRAD2T_GPU_CASES)
def test_wgrad2_thin(gpu, case):
    kc.check_conv_wgrad(gpu, *case)


# thin layers at Fast-pathway sizes (<= 32 output columns, K <= 128) through the general forward / data-gradient kernels
THIN_GPU_CASES = [
    ((4, 8, 16, 56, 56), 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),        # res2 b: both directions 128-wide, BN 16
    ((4, 32, 16, 56, 56), 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),       # res2 a: fwd K 96 BN 16 / dgrad K 24 BN 32 (32-wide)
    ((4, 8, 16, 56, 56), 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),       # res2 c: fwd K 8 BN 32 / dgrad K 32 BN 16
    ((4, 16, 16, 28, 28), 16, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),      # K 16
    ((3, 8, 7, 30, 30), 24, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),        # 18900 rows (ragged last stage), Co 24
    ((4, 8, 16, 57, 57), 16, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1)),       # stride 2: thin forward, general data gradient
]


@pytest.mark.parametrize("case", THIN_GPU_CASES)
def test_thin_layers(gpu, case):
    kc.check_conv_fwd(gpu, *case)
    kc.check_conv_dgrad(gpu, *case)


@pytest.mark.parametrize("case", IGEMM2_CASES)
def test_igemm2(gpu, case):
    kc.check_conv_fwd(gpu, *case)
    kc.check_conv_dgrad(gpu, *case)
    kc.check_conv_wgrad(gpu, *case)       # sf_wgrad2.h (row table + direct-to-LDS + transpose reads) at these sizes


def test_igemm2_small_shapes_forced(gpu):
    """The hostsim case list of the second-generation kernel on hardware, thresholds lowered through the environment
    (a subprocess: the dispatcher reads them once)."""
    code = ("import torch; from tests import kernel_checks as kc; from tests.test_igemm2_hostsim import CASES;"
            "d=torch.device('cuda:0');"
            "from tests.test_igemm2_hostsim import STRIDED;"
            "[ (kc.check_conv_fwd(d,*c), kc.check_conv_dgrad(d,*c)) for c in CASES ];"
            "[ kc.check_conv_dgrad(d,*c) for c in STRIDED ];"
            "kc.check_conv_dgrad(d,(1,32,9,4,4),64,(7,1,1),(4,1,1),(3,0,0),resid=True);"
            "from tests.test_igemm2_hostsim import WGRAD2_CASES;"
            "[ kc.check_conv_wgrad(d,*c) for c in WGRAD2_CASES ];"
            "from tests.test_igemm2_hostsim import WGRAD2T_CASES, THIN_CASES;"
            "[ kc.check_conv_wgrad(d,*c) for c in WGRAD2T_CASES ];"
            "[ (kc.check_conv_fwd(d,*c), kc.check_conv_dgrad(d,*c)) for c in THIN_CASES ];"
            "kc.check_conv_dgrad(d,(1,64,2,9,9),64,(1,3,3),(1,1,1),(0,1,1),resid=True);"
            "kc.check_conv_fwd_fused(d,(1,64,2,9,9),72,(1,3,3),(1,1,1),(0,1,1),resid=True,relu=True); print('ok')")
    env = dict(os.environ, SF_IGEMM2_MINK="32", SF_IGEMM2_MINROWS="1", SF_WGRAD2_MINK="32", SF_WGRAD2_MINROWS="1",
               SF_WGRAD2_BLOCKS="6", SF_WGRAD2T_MINROWS="1", SF_WGRAD2T_BLOCKS="5")
    env.pop("SFAMD_LIBRARY", None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("case", CONV_CASES + BIG_CASES)
def test_conv_fwd(gpu, case):
    kc.check_conv_fwd(gpu, *case)


@pytest.mark.parametrize("case", CONV_CASES + BIG_CASES)
def test_conv_dgrad(gpu, case):
    kc.check_conv_dgrad(gpu, *case)


@pytest.mark.parametrize("case", CONV_CASES + BIG_CASES)
def test_conv_wgrad(gpu, case):
    kc.check_conv_wgrad(gpu, *case)


def test_conv_fused_input_bn(gpu):
    kc.check_conv_fwd(gpu, (2, 64, 4, 14, 14), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), affine=True)
    kc.check_conv_wgrad(gpu, (2, 64, 4, 14, 14), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), affine=True, out_scale=0.25)
    kc.check_conv_fwd(gpu, (1, 16, 1, 6, 6), 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), ldx_extra=8)
    kc.check_conv_dgrad(gpu, (2, 16, 2, 9, 9), 16, (3, 1, 1), (1, 1, 1), (1, 0, 0), resid=True)


def test_conv_stem(gpu):
    kc.check_conv_fwd(gpu, (2, 8, 2, 32, 32), 64, (1, 7, 7), (1, 2, 2), (0, 3, 3), Cw=3)
    kc.check_conv_fwd(gpu, (1, 8, 8, 32, 32), 8, (5, 7, 7), (1, 2, 2), (2, 3, 3), Cw=3)
    kc.check_conv_wgrad(gpu, (2, 8, 2, 32, 32), 64, (1, 7, 7), (1, 2, 2), (0, 3, 3), Cw=3)
    kc.check_conv_wgrad(gpu, (1, 8, 8, 32, 32), 8, (5, 7, 7), (1, 2, 2), (2, 3, 3), Cw=3)


STEM_DIRECT_CASES = [
    ((2, 8, 6, 36, 22), 8, (5, 7, 4), (1, 2, 1), (2, 3, 2)),      # ragged tiles in t, h and w
    ((1, 8, 3, 20, 20), 16, (1, 7, 4), (1, 2, 1), (0, 3, 2)),     # kT = 1, 16 output channels
    ((1, 8, 9, 10, 9), 8, (3, 5, 4), (2, 1, 1), (1, 2, 2)),       # temporal stride 2, spatial stride 1
    ((2, 8, 16, 112, 56), 8, (5, 7, 4), (1, 2, 1), (2, 3, 2)),    # Fast-stem geometry, many workgroups per CU
]


@pytest.mark.parametrize("case", STEM_DIRECT_CASES)
def test_stem_direct(gpu, case):
    """W-pair-folded thin stems: the LDS-patch direct convolution of sf_stem.h (forward + weight gradient)."""
    kc.check_conv_fwd(gpu, *case, Cw=8)
    kc.check_conv_wgrad(gpu, *case, Cw=8)


def test_thin3_direct_convolution(gpu):
    """8-channel (kT, kH, 3) stride-1 layers on the LDS-patch direct convolution of sf_stem.h (forward, data gradient with the fused
    BatchNorm-backward sums, weight gradient) at Fast-pathway res2 size and two ragged shapes."""
    for shp, co, k, p in (((4, 8, 16, 56, 56), 8, (1, 3, 3), (0, 1, 1)), ((2, 8, 5, 30, 30), 16, (1, 3, 3), (0, 1, 1)),
                          ((2, 8, 6, 14, 14), 8, (3, 3, 3), (1, 1, 1))):
        kc.check_conv_fwd(gpu, shp, co, k, (1, 1, 1), p)
        kc.check_conv_wgrad(gpu, shp, co, k, (1, 1, 1), p)
    for shp, co, k, p in (((4, 8, 16, 56, 56), 8, (1, 3, 3), (0, 1, 1)), ((2, 16, 5, 30, 30), 8, (1, 3, 3), (0, 1, 1)),
                          ((2, 8, 6, 14, 14), 8, (3, 3, 3), (1, 1, 1))):
        kc.check_conv_dgrad(gpu, shp, co, k, (1, 1, 1), p)
        kc.check_conv_dgrad_bn(gpu, shp, co, k, p)


def test_wgrad_round1_kernel_shapes(gpu):
    """Two shapes that take the round-1 weight-gradient kernel (K < 192 / few rows): transpose-read fragments vs torch."""
    kc.check_conv_wgrad(gpu, (2, 64, 4, 14, 14), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    kc.check_conv_wgrad(gpu, (2, 16, 2, 9, 9), 24, (1, 1, 1), (1, 1, 1), (0, 0, 0))


@pytest.mark.parametrize("shape,relu,residual", [
    ((2, 16, 2, 5, 5), True, None),
    ((4, 256, 4, 14, 14), True, True),
    ((2, 2048, 2, 7, 7), True, True),
    ((4, 8, 8, 28, 28), True, None),
    ((2, 80, 4, 14, 14), False, True),
    ((2, 64, 4, 28, 28), False, None),
])
def test_bn_chain(gpu, shape, relu, residual):
    kc.check_bn_chain(gpu, shape, relu=relu, residual=residual)


def test_pool(gpu):
    kc.check_pool(gpu, (1, 8, 2, 9, 9))
    kc.check_pool(gpu, (2, 64, 2, 56, 56))


def test_head_mean(gpu):
    kc.check_head_mean(gpu, (4, 2048, 8, 7, 7))
    kc.check_head_mean(gpu, (4, 256, 32, 7, 7))
    kc.check_head_mean(gpu, (2, 16, 2, 3, 4))


def test_layout(gpu):
    kc.check_layout(gpu, (2, 3, 4, 32, 32))
    kc.check_layout(gpu, (1, 16, 1, 4, 4))
    kc.check_layout(gpu, (3, 3, 2, 6, 10))       # S % 4 == 0, fewer positions than one workgroup
    kc.check_layout(gpu, (2, 3, 3, 5, 5))        # S % 4 != 0: the one-position kernel


def test_bn_finalize_long_tables(gpu):
    kc.check_bn_finalize_long(gpu, 300, 24)
    kc.check_bn_finalize_long(gpu, 2500, 8)
    kc.check_bn_finalize_long(gpu, 25088, 64)    # the slow-pathway stem at batch 32
    kc.check_bn_finalize_long(gpu, 200, 40)


def test_wgrad_many_splits(gpu):
    kc.check_conv_wgrad(gpu, (4, 8, 16, 56, 56), 8, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    kc.check_conv_wgrad(gpu, (4, 32, 16, 56, 56), 8, (3, 1, 1), (1, 1, 1), (1, 0, 0))
    kc.check_conv_wgrad(gpu, (4, 8, 16, 56, 56), 32, (1, 1, 1), (1, 1, 1), (0, 0, 0))


@pytest.mark.gpu
def test_roi_align_published_vectors(gpu):
    """ROIAlign kernel pinned to detectron2's published unit-test vectors (tests/golden/roi_align_detectron2.json)."""
    kc.check_roi_known_answer(gpu)


@pytest.mark.gpu
@pytest.mark.parametrize("aligned", [True, False])
def test_roi_pool(gpu, aligned):
    """RoI head pooling (temporal mean -> ROIAlign -> max) vs the oracle's ROIAlign restatement."""
    kc.check_roi_pool(gpu, aligned=aligned)
    kc.check_roi_pool(gpu, shape=(4, 256, 8, 16, 16), aligned=aligned, seed=3)


@pytest.mark.parametrize("arch,reverse", [("slowfast", False), ("slowfast", True), ("c2d", False)])
def test_pack_clip_u8(gpu, arch, reverse):
    kc.check_pack_clip(gpu, arch, reverse)


def test_conv_fwd_fused(gpu):
    """sf_conv_fwd_fused at res-stage geometries: direct-to-LDS 1x1x1 with residual + ReLU, strided 1x3x3, temporal
    3x1x1, a strided projection shortcut without ReLU."""
    kc.check_conv_fwd_fused(gpu, (2, 64, 4, 14, 14), 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), resid=True)
    kc.check_conv_fwd_fused(gpu, (2, 64, 4, 14, 14), 64, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    kc.check_conv_fwd_fused(gpu, (2, 256, 4, 7, 7), 64, (3, 1, 1), (1, 1, 1), (1, 0, 0))
    kc.check_conv_fwd_fused(gpu, (2, 80, 4, 14, 14), 136, (1, 1, 1), (1, 2, 2), (0, 0, 0), relu=False)
    kc.check_conv_fwd_fused(gpu, (1, 32, 8, 7, 7), 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), resid=True, bias=False)
    # thin W-pair-folded stems: LDS-patch direct convolution with the bias / ReLU epilogue
    kc.check_conv_fwd_fused(gpu, (2, 8, 8, 64, 32), 8, (5, 7, 4), (1, 2, 1), (2, 3, 2))
    kc.check_conv_fwd_fused(gpu, (2, 8, 3, 36, 22), 16, (1, 7, 4), (1, 2, 1), (0, 3, 2), relu=False)


@pytest.mark.gpu
def test_prep_weights_batch_equals_per_layer_gpu(gpu):
    kc.check_prep_weights_batch(gpu, [
        (64, 64, 64, (1, 3, 3), True), (2048, 512, 512, (1, 1, 1), True), (8, 32, 32, (3, 1, 1), True),
        (54, 40, 40, (1, 1, 1), True), (64, 3, 8, (1, 7, 7), False), (16, 16, 16, (3, 3, 3), True),
        (512, 512, 512, (1, 3, 3), True), (8, 8, 8, (5, 11, 11), True)])


@pytest.mark.gpu
def test_conv_dgrad_fused_bn_reduce(gpu):
    """sf_conv_dgrad_bn on the SlowFast-R50 inner-BatchNorm geometries: c -> b (pointwise, K = C) and b -> a (1x3x3 / 3x1x1)."""
    assert kc.check_conv_dgrad_bn(gpu, (4, 64, 8, 56, 56), 256, (1, 1, 1), (0, 0, 0)) == 784       # res2 c: 128-row tiles
    assert kc.check_conv_dgrad_bn(gpu, (4, 64, 8, 56, 56), 64, (1, 3, 3), (0, 1, 1)) == 392        # res2 b: igemm2, 256 rows
    kc.check_conv_dgrad_bn(gpu, (4, 128, 8, 28, 28), 512, (1, 1, 1), (0, 0, 0))                     # res3 c: igemm2
    kc.check_conv_dgrad_bn(gpu, (4, 256, 8, 14, 14), 256, (1, 3, 3), (0, 1, 1))
    kc.check_conv_dgrad_bn(gpu, (4, 512, 8, 7, 7), 2048, (1, 1, 1), (0, 0, 0))
    kc.check_conv_dgrad_bn(gpu, (4, 8, 32, 56, 56), 32, (1, 1, 1), (0, 0, 0))                       # Fast pathway c
    kc.check_conv_dgrad_bn(gpu, (4, 8, 32, 56, 56), 8, (1, 3, 3), (0, 1, 1))
    kc.check_conv_dgrad_bn(gpu, (4, 16, 32, 28, 28), 64, (1, 1, 1), (0, 0, 0), resid=True)
