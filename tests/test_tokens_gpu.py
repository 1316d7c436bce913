"""GPU (-m gpu): token-space kernels (MViT path) on a real MI355X against the torch fp32 reference ops."""
import pytest

from tests import token_checks as tc

pytestmark = pytest.mark.gpu


def test_gemm_linear(gpu):
    tc.check_gemm(gpu, 6273, 192, 576)
    tc.check_gemm(gpu, 1000, 96, 288)
    tc.check_gemm(gpu, 394, 768, 3072, resid=False)
    tc.check_gemm(gpu, 77, 32, 24, bias=False, resid=False)
    tc.check_gemm(gpu, 100001, 512, 264)                 # sf_igemm2 with 128 x 128 tiles (plain matrix product, K >= 512, > 320 tiles)


def test_rows32_side_rows(gpu):
    """fp32 side rows of the residual sums at MViTv2-S shapes: the attention projection of stage 3 (round-1 GEMM kernel), fc2 of
    stage 3 (K = 1536: second-generation kernel), every row of the last stage."""
    tc.check_rows32(gpu, 4, 1569, 384, 384)
    tc.check_rows32(gpu, 4, 1569, 1536, 384)
    tc.check_rows32(gpu, 3, 1569, 192, 192, bias=False, src32=False)
    tc.check_rows32(gpu, 4, 393, 768, 768, period_full=True)
    tc.check_rows32(gpu, 16, 393, 3072, 768, period_full=True)


def test_layernorm(gpu):
    for M, C in ((5000, 96), (3001, 192), (777, 384), (400, 768), (37, 32)):
        tc.check_layernorm(gpu, M, C)


def test_gelu(gpu):
    tc.check_gelu(gpu, 1 << 16)


def test_dwconv_tokens(gpu):
    tc.check_dwconv(gpu, 2, 2, 96, (4, 14, 14), (3, 3, 3), (1, 2, 2), cls=1)
    tc.check_dwconv(gpu, 2, 1, 96, (4, 28, 28), (3, 3, 3), (1, 4, 4), cls=1)
    tc.check_dwconv(gpu, 1, 4, 96, (4, 7, 7), (3, 3, 3), (1, 1, 1), cls=1)
    tc.check_dwconv(gpu, 2, 1, 24, (8, 12, 12), (5, 1, 1), (1, 1, 1), cls=0)
    tc.check_dwconv(gpu, 2, 1, 216, (4, 14, 14), (3, 3, 3), (1, 2, 2), cls=0)
    # X3D widths: narrow (56) and wide (432) layers, whole 4-column groups
    tc.check_dwconv(gpu, 2, 1, 56, (4, 28, 28), (3, 3, 3), (1, 1, 1), cls=0)
    tc.check_dwconv(gpu, 2, 1, 56, (4, 56, 56), (3, 3, 3), (1, 2, 2), cls=0)
    tc.check_dwconv(gpu, 1, 1, 432, (4, 8, 8), (3, 3, 3), (1, 1, 1), cls=0)
    # MViTv2-S stage 3 / 4 production planes (VERDICT r4 test gap): 14-wide, stride 1 (q pooling: sf_dwtile_kernel forward + data
    # gradient, sf_dwtile_wgrad_kernel<1, ...>) and stride 2 at C = 384 (4 heads) and C = 768 (8 heads)
    tc.check_dwconv(gpu, 2, 4, 96, (4, 14, 14), (3, 3, 3), (1, 1, 1), cls=1)
    tc.check_dwconv(gpu, 1, 8, 96, (4, 14, 14), (3, 3, 3), (1, 1, 1), cls=1)
    tc.check_dwconv(gpu, 2, 4, 96, (8, 14, 14), (3, 3, 3), (1, 2, 2), cls=1)
    tc.check_dwconv(gpu, 1, 8, 96, (2, 7, 7), (3, 3, 3), (1, 1, 1), cls=1)


def test_dwconv_temporal(gpu, monkeypatch):
    """sf_dwtemporal.h at the X3D-M stem's production plane (24 channels, 16 frames, 112 x 112; batch 3 here: several passes per
    workgroup), other window lengths, and the stencils it replaces."""
    tc.check_dwconv(gpu, 3, 1, 24, (16, 112, 112), (5, 1, 1), (1, 1, 1), cls=0)
    tc.check_dwconv(gpu, 2, 2, 8, (5, 20, 20), (3, 1, 1), (1, 1, 1), cls=0)
    tc.check_dwconv(gpu, 1, 1, 40, (2, 9, 9), (7, 1, 1), (1, 1, 1), cls=0)
    monkeypatch.setenv("SF_DWTP_BLOCKS", "7")
    tc.check_dwconv(gpu, 2, 1, 24, (8, 12, 12), (5, 1, 1), (1, 1, 1), cls=0)
    monkeypatch.delenv("SF_DWTP_BLOCKS")
    monkeypatch.setenv("SF_DW_TEMPORAL", "0")
    tc.check_dwconv(gpu, 2, 1, 24, (8, 12, 12), (5, 1, 1), (1, 1, 1), cls=0)


def test_dwconv_pair(gpu):
    """pool_k / pool_v in ONE launch per direction at the MViTv2-S production planes (stage 3: 14 -> 7 stride 2, C = 384; stage 4:
    7 x 7 stride 1, C = 768; block 3: 28 -> 14 stride 2): bit-equal to two single launches."""
    tc.check_dwconv_pair(gpu, 2, 4, 96, (8, 14, 14), (1, 2, 2), cls=1)
    tc.check_dwconv_pair(gpu, 2, 8, 96, (8, 7, 7), (1, 1, 1), cls=1)
    tc.check_dwconv_pair(gpu, 2, 4, 96, (4, 28, 28), (1, 2, 2), cls=1)
    tc.check_dwconv_pair(gpu, 1, 2, 96, (4, 56, 56), (1, 1, 1), cls=1)


def test_dwconv_ring_sweep(gpu, monkeypatch):
    """sf_dwsweep.h at the production planes of MViTv2-S (56- / 28- / 14- / 7-wide, every stride, heads sharing the weight, cls
    rows, slices of a wider tensor) and X3D-M (54 -> 56, 108 -> 112, 216, 432 channels: tail chunks; BatchNorm partial sums), plus
    forced ragged tilings."""
    tc.check_dwconv(gpu, 2, 1, 96, (4, 56, 56), (3, 3, 3), (1, 1, 1), cls=1)     # block 0 q
    tc.check_dwconv(gpu, 2, 2, 96, (4, 56, 56), (3, 3, 3), (1, 2, 2), cls=1)     # block 1 q
    tc.check_dwconv(gpu, 2, 2, 96, (4, 28, 28), (3, 3, 3), (1, 1, 1), cls=1)     # block 2 q
    tc.check_dwconv(gpu, 2, 4, 96, (4, 28, 28), (3, 3, 3), (1, 2, 2), cls=1)     # block 3 q / k / v
    tc.check_dwconv(gpu, 1, 8, 96, (4, 14, 14), (3, 3, 3), (1, 2, 2), cls=1)     # block 14 q
    tc.check_dwconv(gpu, 2, 1, 56, (4, 112, 112), (3, 3, 3), (1, 2, 2), cls=0)   # X3D s2 first block
    tc.check_dwconv(gpu, 2, 1, 56, (4, 56, 56), (3, 3, 3), (1, 1, 1), cls=0)
    tc.check_dwconv(gpu, 2, 1, 112, (4, 56, 56), (3, 3, 3), (1, 2, 2), cls=0)
    tc.check_dwconv(gpu, 2, 1, 112, (4, 28, 28), (3, 3, 3), (1, 1, 1), cls=0)
    tc.check_dwconv(gpu, 2, 1, 216, (4, 28, 28), (3, 3, 3), (1, 2, 2), cls=0)
    tc.check_dwconv(gpu, 2, 1, 216, (4, 14, 14), (3, 3, 3), (1, 1, 1), cls=0)
    tc.check_dwconv(gpu, 2, 1, 432, (4, 14, 14), (3, 3, 3), (1, 2, 2), cls=0)
    tc.check_dwconv(gpu, 2, 1, 432, (4, 7, 7), (3, 3, 3), (1, 1, 1), cls=0)
    # strides 4 and 8 (k / v pooling of blocks 0 - 2): packed staging, one-tap data gradient
    tc.check_dwconv(gpu, 2, 1, 96, (4, 56, 56), (3, 3, 3), (1, 8, 8), cls=1)
    tc.check_dwconv(gpu, 2, 2, 96, (4, 56, 56), (3, 3, 3), (1, 4, 4), cls=1)
    tc.check_dwconv(gpu, 2, 2, 96, (4, 28, 28), (3, 3, 3), (1, 4, 4), cls=1)
    for sl, grp, seg in ((7, 2, 2), (7, 4, 1), (4, 2, 2), (4, 1, 3)):
        monkeypatch.setenv("SF_DWR_SL", str(sl))
        monkeypatch.setenv("SF_DWR_NGRP", str(grp))
        monkeypatch.setenv("SF_DWR_NSEG", str(seg))
        tc.check_dwconv(gpu, 2, 4, 96, (5, 14, 14), (3, 3, 3), (1, 1, 1), cls=1)
        tc.check_dwconv(gpu, 1, 1, 40, (3, 13, 21), (3, 3, 3), (1, 2, 2), cls=0)
    for k in ("SF_DWR_SL", "SF_DWR_NGRP", "SF_DWR_NSEG"):
        monkeypatch.delenv(k)
    monkeypatch.setenv("SF_DW_ROT", "0")                                         # the four-channel v_fma_mix body, forced tilings
    monkeypatch.setenv("SF_DWS_TH", "5")
    monkeypatch.setenv("SF_DWS_TW", "6")
    tc.check_dwconv(gpu, 2, 4, 96, (3, 14, 14), (3, 3, 3), (1, 1, 1), cls=1)
    tc.check_dwconv(gpu, 2, 4, 96, (3, 14, 14), (3, 3, 3), (1, 2, 2), cls=1)
    monkeypatch.setenv("SF_DWS_TH", "11")
    monkeypatch.setenv("SF_DWS_TW", "20")
    monkeypatch.setenv("SF_DWS_NSEG", "3")
    tc.check_dwconv(gpu, 1, 1, 40, (2, 12, 20), (3, 3, 3), (1, 1, 1), cls=1)
    tc.check_dwconv(gpu, 1, 1, 40, (1, 13, 21), (3, 3, 3), (1, 2, 2), cls=0)


def test_dwconv_tokens_stencils(gpu, monkeypatch):
    """The stencils and the round-4 LDS plane sweep the ring sweep replaced (still the fallback for other geometries)."""
    monkeypatch.setenv("SF_DW_SWEEP", "0")
    tc.check_dwconv(gpu, 2, 2, 96, (4, 14, 14), (3, 3, 3), (1, 2, 2), cls=1)
    tc.check_dwconv(gpu, 2, 4, 96, (4, 14, 14), (3, 3, 3), (1, 1, 1), cls=1)
    tc.check_dwconv(gpu, 2, 1, 56, (4, 28, 28), (3, 3, 3), (1, 1, 1), cls=0)
    tc.check_dwconv(gpu, 2, 1, 216, (4, 14, 14), (3, 3, 3), (1, 2, 2), cls=0)


def test_token_pool(gpu):
    tc.check_token_pool(gpu, 2, 192, (4, 28, 28), (1, 2, 2))
    tc.check_token_pool(gpu, 1, 96, (3, 13, 15), (1, 2, 2))


def test_attention_core(gpu):
    tc.check_attention_core(gpu, 2, 2, 96, (4, 14, 14), (4, 7, 7))
    tc.check_attention_core(gpu, 1, 1, 96, (8, 28, 28), (8, 7, 7))     # Nk = 393: stage-1 key count
    tc.check_attention_core(gpu, 1, 4, 96, (4, 7, 7), (4, 7, 7))
    tc.check_attention_core(gpu, 1, 1, 32, (4, 16, 16), (4, 16, 16))   # 1025 keys: 4-slot softmax rows


@pytest.mark.parametrize("case", [
    (2, 1, 96, (8, 56, 56), (8, 7, 7), True, True, True),      # MViTv2-S block 0: 25089 queries x 393 keys
    (1, 2, 96, (8, 28, 28), (8, 14, 14), True, True, True),    # block 1: 6273 x 1569, 14+14+8 bias buckets
    (2, 8, 96, (8, 7, 7), (8, 7, 7), True, True, True),        # stage 4: 8 heads, 393 x 393
    (1, 2, 32, (2, 6, 6), (2, 3, 3), True, True, True),
    (1, 1, 64, (1, 5, 9), (1, 5, 9), False, False, False),
    (1, 1, 128, (2, 9, 9), (2, 5, 5), True, False, True),
])
def test_attention_fused(gpu, case):
    """sf_attn_fwd / sf_attn_bwd (no score tensor) vs the reference attention math incl. rel-pos table gradients."""
    tc.check_attention_fused(gpu, *case)


def test_gemm_gelu_epilogues(gpu):
    tc.check_gemm_gelu(gpu, 6273, 192, 768)
    tc.check_gemm_gelu(gpu, 1000, 96, 384, seed=1)
