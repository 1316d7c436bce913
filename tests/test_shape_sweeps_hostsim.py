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
