"""CPU: every kernel + launcher of libsfamd executed through the host functional simulator."""
import pytest

from tests import kernel_checks as kc

CONV_CASES = [
    # in_shape (N,Ci,T,H,W), Co, kernel, stride, pad, dil
    ((2, 16, 2, 5, 5), 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),      # pointwise, M=100 < one tile
    ((1, 32, 2, 6, 6), 64, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 1, 1)),      # strided shortcut conv
    ((1, 16, 4, 4, 4), 16, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),      # temporal conv
    ((1, 8, 2, 9, 9), 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),        # spatial conv, M=162 (2 tiles)
    ((1, 16, 1, 9, 9), 24, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1)),      # stride 2, Co=24 (masked cols)
    ((1, 8, 1, 8, 8), 16, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),       # dilation 2
    ((1, 8, 8, 3, 3), 16, (7, 1, 1), (4, 1, 1), (3, 0, 0), (1, 1, 1)),       # FuseFastToSlow lateral
    ((1, 80, 1, 4, 4), 136, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),     # Ci=80, two N tiles (136)
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(sim, case):
    kc.check_conv_fwd(sim, *case)


def test_conv_fwd_fused_input_bn(sim):
    kc.check_conv_fwd(sim, (1, 16, 1, 7, 7), 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), affine=True)


def test_conv_fwd_channel_slice_input(sim):
    kc.check_conv_fwd(sim, (1, 16, 1, 6, 6), 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), ldx_extra=8)


def test_conv_fwd_stem(sim):
    # 3-channel clip zero-padded to 8 channels, weight has Cw=3
    kc.check_conv_fwd(sim, (1, 8, 1, 12, 12), 8, (1, 7, 7), (1, 2, 2), (0, 3, 3), Cw=3)
    kc.check_conv_fwd(sim, (1, 8, 5, 6, 6), 8, (5, 7, 7), (1, 2, 2), (2, 3, 3), Cw=3)


# W-pair-folded thin stems (engine.StemConvUnit geometry): the LDS-patch direct convolution of sf_stem.h
STEM_CASES = [
    ((2, 8, 6, 36, 22), 8, (5, 7, 4), (1, 2, 1), (2, 3, 2)),     # Fast stem, ragged tiles in t, h and w
    ((1, 8, 3, 20, 20), 16, (1, 7, 4), (1, 2, 1), (0, 3, 2)),    # kT = 1, 16 output channels
    ((1, 8, 9, 10, 9), 8, (3, 5, 4), (2, 1, 1), (1, 2, 2)),      # temporal stride 2, spatial stride 1
]


@pytest.mark.parametrize("case", STEM_CASES)
def test_stem_direct_fwd(sim, case):
    kc.check_conv_fwd(sim, *case, Cw=8)


@pytest.mark.parametrize("groups", [8, 3, 2, 1, 0])
def test_stem_fwd_sliding_ring(sim, monkeypatch, capfd, groups):
    """Fast-stem forward on the sliding kernel (sf_stem_fwd_slide_kernel): 14 frames = 3.5 groups of 4, so the 8-slot frame ring wraps
    and the last group is ragged; a workgroup walks the whole clip (8), runs of 3 + 1 groups, single groups, or the tile kernel (0).
    Output and the BatchNorm statistics table against F.conv3d in every split."""
    monkeypatch.setenv("SF_STEM_SLIDE", str(groups))
    monkeypatch.setenv("SF_TRACE", "1")
    kc.check_conv_fwd(sim, (2, 8, 14, 36, 22), 8, (5, 7, 4), (1, 2, 1), (2, 3, 2), Cw=8)
    if groups == 2:        # bias + ReLU epilogue (the eval-mode stem)
        kc.check_conv_fwd_fused(sim, (1, 8, 9, 20, 20), 8, (5, 7, 4), (1, 2, 1), (2, 3, 2))
    err = capfd.readouterr().err
    assert ("stem_fwd_slide" in err) == (groups > 0), err


@pytest.mark.parametrize("case", STEM_CASES)
def test_stem_direct_wgrad(sim, case):
    kc.check_conv_wgrad(sim, *case, Cw=8)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_dgrad(sim, case):
    kc.check_conv_dgrad(sim, *case)


def test_conv_dgrad_residual(sim):
    kc.check_conv_dgrad(sim, (1, 16, 2, 5, 5), 16, (3, 1, 1), (1, 1, 1), (1, 0, 0), resid=True)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_wgrad(sim, case):
    kc.check_conv_wgrad(sim, *case)


def test_conv_wgrad_fused_input_bn_and_scale(sim):
    kc.check_conv_wgrad(sim, (1, 16, 1, 7, 7), 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), affine=True, out_scale=0.25)


def test_conv_wgrad_stem(sim):
    kc.check_conv_wgrad(sim, (1, 8, 1, 12, 12), 8, (1, 7, 7), (1, 2, 2), (0, 3, 3), Cw=3)



@pytest.mark.parametrize("shape,relu,residual", [
    ((2, 16, 2, 5, 5), True, None),
    ((2, 8, 3, 7, 7), False, None),
    ((1, 80, 2, 4, 4), True, True),
    ((2, 24, 1, 6, 6), False, True),
])
def test_bn_chain(sim, shape, relu, residual):
    kc.check_bn_chain(sim, shape, relu=relu, residual=residual)


def test_pool(sim):
    kc.check_pool(sim, (1, 8, 2, 9, 9))
    kc.check_pool(sim, (2, 16, 1, 8, 8))


def test_head_mean(sim):
    kc.check_head_mean(sim, (2, 16, 2, 3, 4))
    kc.check_head_mean(sim, (1, 8, 4, 2, 2))


def test_layout(sim):
    kc.check_layout(sim, (2, 3, 2, 5, 5))
    kc.check_layout(sim, (1, 16, 1, 4, 4))
    kc.check_layout(sim, (2, 3, 2, 4, 6))        # S % 4 == 0: four positions per lane (sf_ncthw_to_cl_quad_kernel)
    kc.check_layout(sim, (1, 4, 1, 2, 2))


def test_bn_finalize_long_tables(sim):
    kc.check_bn_finalize_long(sim, 300, 24)      # 1024-thread finalize (128 row segments per channel), no fold
    kc.check_bn_finalize_long(sim, 2500, 8)
    kc.check_bn_finalize_long(sim, 9000, 16)     # ragged segments
    kc.check_bn_finalize_long(sim, 200, 40)      # short table: 256-thread finalize
    kc.check_bn_finalize_long(sim, 17000, 8)     # > 16384 rows: in-place fold (group 64, ragged last group) first


def test_wgrad_many_splits(sim):
    """Tiny-channel layer with a long position axis: many split partials, multi-lane reduce kernel."""
    kc.check_conv_wgrad(sim, (2, 8, 8, 24, 24), 8, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    kc.check_conv_wgrad(sim, (1, 8, 4, 32, 32), 32, (3, 1, 1), (1, 1, 1), (1, 0, 0))


def test_roi_align_published_vectors(sim):
    kc.check_roi_known_answer(sim)


@pytest.mark.parametrize("aligned", [True, False])
def test_roi_pool(sim, aligned):
    kc.check_roi_pool(sim, aligned=aligned)


@pytest.mark.parametrize("arch,reverse", [("slowfast", False), ("slowfast", True), ("c2d", False)])
def test_pack_clip_u8(sim, arch, reverse):
    kc.check_pack_clip(sim, arch, reverse)


def test_prep_weights_batch_equals_per_layer(sim):
    kc.check_prep_weights_batch(sim, [
        (64, 64, 64, (1, 3, 3), True),        # 3x3 bottleneck conv
        (256, 64, 64, (1, 1, 1), True),       # pointwise, several brick rows / columns
        (8, 32, 32, (3, 1, 1), True),         # Fast pathway: fewer output channels than a brick holds
        (54, 40, 40, (1, 1, 1), True),        # ragged: Co 54 -> 56 padded rows, 40 channels = 1.25 bricks
        (64, 3, 8, (1, 7, 7), False),         # RGB stem: 3 of 8 channels real, 49 taps, no dgrad operand
        (16, 16, 16, (3, 3, 3), True),        # 27 taps: 16-channel bricks
        (8, 8, 8, (5, 11, 11), True),         # 605 taps: does not fit a brick -> element-wise blocks
    ])


def test_conv_dgrad_fused_bn_reduce(sim, monkeypatch):
    """Data gradient + the reduction pass of the consumer BatchNorm's backward in one launch: both implicit-GEMM generations
    (tile heights 128 / 256), every tile width of the first one, a 3x3 with padding, a residual, ragged last tiles."""
    assert kc.check_conv_dgrad_bn(sim, (2, 64, 2, 9, 9), 256, (1, 1, 1), (0, 0, 0)) == 3           # 324 rows / 128
    assert kc.check_conv_dgrad_bn(sim, (2, 16, 2, 8, 8), 24, (1, 3, 3), (0, 1, 1)) == 2            # BN = 16 tile
    kc.check_conv_dgrad_bn(sim, (2, 32, 2, 8, 8), 64, (3, 1, 1), (1, 0, 0), resid=True)
    kc.check_conv_dgrad_bn(sim, (1, 8, 4, 8, 8), 32, (1, 1, 1), (0, 0, 0))
    kc.check_conv_dgrad_bn(sim, (2, 128, 2, 8, 8), 128, (1, 1, 1), (0, 0, 0))                       # two column tiles
    monkeypatch.setenv("SF_IGEMM2_MINK", "64")
    monkeypatch.setenv("SF_IGEMM2_MINROWS", "64")
    assert kc.check_conv_dgrad_bn(sim, (2, 64, 2, 9, 9), 64, (1, 3, 3), (0, 1, 1)) == 2            # igemm2: 324 rows / 256
    kc.check_conv_dgrad_bn(sim, (2, 64, 2, 8, 8), 128, (1, 1, 1), (0, 0, 0), resid=True)
