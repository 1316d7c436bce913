"""CPU: shape sweeps of the depthwise and pointwise (direct-to-LDS) kernels through the host simulator."""


def test_depthwise_blocked_stencil_shapes(sim):
    """The W-blocked depthwise stencils (uniform plane loop, address-selected zero taps; the only version since round 3) on narrow
    and wide layers, strides 1 / 2, whole and ragged 4-column groups, cls rows, (5,1,1) temporal kernels."""
    from tests import token_checks as tc
    d = sim
    tc.check_dwconv(d, 2, 2, 16, (2, 6, 6), (3, 3, 3), (1, 2, 2), cls=1)
    tc.check_dwconv(d, 1, 1, 32, (4, 5, 5), (3, 3, 3), (1, 1, 1), cls=1)
    tc.check_dwconv(d, 2, 1, 24, (6, 4, 4), (5, 1, 1), (1, 1, 1), cls=0)
    tc.check_dwconv(d, 1, 2, 8, (2, 7, 7), (3, 3, 3), (1, 2, 2), cls=0)
    tc.check_dwconv(d, 1, 1, 16, (3, 5, 8), (3, 3, 3), (1, 1, 1), cls=0)
    tc.check_dwconv(d, 1, 1, 8, (2, 6, 16), (3, 3, 3), (1, 2, 2), cls=1)
    tc.check_dwconv(d, 1, 1, 120, (2, 4, 8), (3, 3, 3), (1, 1, 1), cls=0)
    tc.check_dwconv(d, 1, 1, 16, (5, 4, 4), (5, 1, 1), (1, 1, 1), cls=0)


def test_igemm_direct_to_lds_shapes(sim):
    """The direct-to-LDS (pointwise) path of sf_igemm_kernel on 1x1x1 convolutions (forward, fused eval form, data gradient), a
    Linear layer and a batched attention-style GEMM, 128- and 64-wide tiles, K of 3 to 9 steps, ragged M."""
    from tests import kernel_checks as kc, token_checks as tc
    d = sim
    kc.check_conv_fwd(d, (1, 96, 2, 7, 9), 128, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    kc.check_conv_fwd(d, (2, 288, 1, 5, 5), 64, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    kc.check_conv_fwd_fused(d, (1, 128, 2, 6, 6), 72, (1, 1, 1), (1, 1, 1), (0, 0, 0), resid=True)
    kc.check_conv_dgrad(d, (1, 64, 2, 6, 6), 160, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    tc.check_gemm(d, 150, 96, 192)
    tc.check_gemm(d, 77, 256, 64)
    tc.check_gemm_gelu(d, 130, 96, 128)
    tc.check_attention_core(d, 1, 1, 96, (2, 3, 3), (2, 3, 3))


# 8-channel (kT, kH, 3) stride-1 layers take the LDS-patch direct convolution of sf_stem.h in all three directions (round 3:
# the Fast pathway's res2 1x3x3 bottleneck): (in_shape, Co, kernel, pad)
THIN3_CASES = [
    ((2, 8, 3, 10, 10), 8, (1, 3, 3), (0, 1, 1)),       # one tile column, ragged rows
    ((1, 8, 5, 9, 21), 16, (1, 3, 3), (0, 1, 1)),       # two column tiles (second ragged), 16 output channels, 5 frames (ragged frame tile)
    ((1, 8, 6, 6, 6), 8, (3, 3, 3), (1, 1, 1)),         # temporal taps: 9 slices, two per wave in the weight gradient
]


def test_thin3_direct_convolution(sim, capfd, monkeypatch):
    import pytest
    from tests import kernel_checks as kc
    monkeypatch.setenv("SF_TRACE", "1")
    for shp, co, k, p in THIN3_CASES:
        kc.check_conv_fwd(sim, shp, co, k, (1, 1, 1), p)
        kc.check_conv_wgrad(sim, shp, co, k, (1, 1, 1), p)
    assert capfd.readouterr().err.count("stem_") >= 2 * len(THIN3_CASES), "the direct convolution must be taken"
    # data gradient: dy has 8 channels (Co = 8), dx 8 or 16; plain, and with the fused BatchNorm-backward sums (both mask forms)
    for shp, co, k, p in (((2, 8, 3, 10, 10), 8, (1, 3, 3), (0, 1, 1)), ((1, 16, 2, 12, 19), 8, (1, 3, 3), (0, 1, 1)),
                          ((1, 8, 6, 6, 6), 8, (3, 3, 3), (1, 1, 1))):
        kc.check_conv_dgrad(sim, shp, co, k, (1, 1, 1), p)
        kc.check_conv_dgrad_bn(sim, shp, co, k, p)
    # not taken: stride 2, 16 input channels in the forward, dilation
    kc.check_conv_fwd(sim, (1, 8, 2, 11, 11), 8, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    kc.check_conv_fwd(sim, (1, 16, 2, 9, 9), 16, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    kc.check_conv_fwd(sim, (1, 8, 2, 9, 9), 8, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2))
