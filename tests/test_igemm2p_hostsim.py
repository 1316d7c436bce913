"""CPU: the persistent implicit GEMM with drain waves (csrc/sf_igemm2p.h) through the host functional simulator: tile lists
(several tiles per workgroup, XCD ranges, idle workgroups), the barrier protocol between the compute and the drain waves (every
wave must meet every barrier: a mismatch hangs the simulator), every epilogue the drain waves cover."""
import pytest

from tests import kernel_checks as kc
from tests import token_checks as tc


@pytest.fixture(params=[3, 256])
def force_p(request, monkeypatch):
    monkeypatch.setenv("SF_IGEMM2", "1")
    monkeypatch.setenv("SF_IGEMM2_MINK", "32")
    monkeypatch.setenv("SF_IGEMM2_MINROWS", "1")
    monkeypatch.setenv("SF_IGEMM2P", "1")
    monkeypatch.setenv("SF_IGEMM2P_MINK", "32")
    monkeypatch.setenv("SF_IGEMM2P_GRID", str(request.param))      # 3 workgroups: up to several tiles each, XCD ranges of 1-2 tiles


CASES = [
    # in_shape (N,Ci,T,H,W), Co, kernel, stride, pad, dil
    ((1, 64, 2, 9, 9), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),       # one ragged tile
    ((2, 64, 3, 12, 12), 136, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),    # 4 M tiles x 2 N tiles, ragged N
    ((1, 128, 4, 6, 6), 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),      # temporal taps
    ((2, 192, 1, 24, 24), 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),   # plain GEMM, 5 x 2 tiles
    ((2, 32, 1, 40, 40), 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),     # ONE K step, 13 tiles
    ((1, 96, 1, 8, 8), 72, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),       # dilation
]


@pytest.mark.parametrize("case", CASES)
def test_igemm2p_fwd(sim, force_p, case):
    kc.check_conv_fwd(sim, *case)


@pytest.mark.parametrize("case", CASES[:4])
def test_igemm2p_dgrad(sim, force_p, case):
    kc.check_conv_dgrad(sim, *case)


def test_igemm2p_epilogues(sim, force_p):
    kc.check_conv_dgrad(sim, (2, 64, 2, 9, 9), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), resid=True)          # residual
    kc.check_conv_fwd_fused(sim, (2, 64, 2, 9, 9), 72, (1, 3, 3), (1, 1, 1), (0, 1, 1), resid=True, relu=True)
    kc.check_bn_chain(sim, (2, 64, 2, 9, 9))                                                            # BatchNorm statistics of the tile


def test_igemm2p_token_gemms(sim, force_p):
    tc.check_gemm(sim, 1000, 96, 288)
    tc.check_gemm(sim, 700, 192, 576)
    tc.check_gemm_gelu(sim, 1000, 96, 384, seed=1)          # fc1 + GELU (two outputs), fc2 data gradient x gelu' + bias column sums
    tc.check_rows32(sim, 4, 157, 384, 384)                  # fp32 side rows of the residual sum
