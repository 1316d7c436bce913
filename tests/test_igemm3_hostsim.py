"""CPU: the third-generation implicit GEMM (csrc/sf_igemm3.h: 256 x 256 x 64 tiles, eight-phase ping-pong of the two wave rows)
through the host functional simulator, against F.conv3d on identical fp16-rounded operands.  The simulator checks index math,
the copy streams (A one K tile ahead, B two) and the barrier structure (the staggered wave rows must meet at every barrier); it
executes copies immediately, so landing-order hazards are the GPU tests' business."""
import pytest

from tests import kernel_checks as kc


@pytest.fixture()
def force_v3(monkeypatch):
    monkeypatch.setenv("SF_IGEMM2", "1")
    monkeypatch.setenv("SF_IGEMM2_MINK", "32")
    monkeypatch.setenv("SF_IGEMM2_MINROWS", "1")
    monkeypatch.setenv("SF_IGEMM3", "1")
    monkeypatch.setenv("SF_IGEMM3_MINN", "8")


CASES = [
    # in_shape (N,Ci,T,H,W), Co, kernel, stride, pad, dil
    ((1, 64, 2, 9, 9), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),       # 9 taps, one K chunk, one ragged tile, 64 of 256 columns
    ((2, 64, 3, 12, 12), 264, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),    # 4 M tiles, two N tiles, ragged N
    ((1, 128, 4, 6, 6), 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),      # temporal taps, two channel chunks per tap (6 K tiles)
    ((2, 192, 1, 20, 20), 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),   # plain GEMM, K = 192 (3 K tiles), 800 rows
    ((2, 64, 1, 20, 20), 320, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),    # ONE K tile (prologue only)
    ((2, 128, 1, 12, 12), 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),   # TWO K tiles
    ((1, 64, 3, 5, 5), 96, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),       # 27 taps
]


@pytest.mark.parametrize("case", CASES)
def test_igemm3_fwd(sim, force_v3, case):
    kc.check_conv_fwd(sim, *case)


@pytest.mark.parametrize("case", CASES)
def test_igemm3_dgrad(sim, force_v3, case):
    kc.check_conv_dgrad(sim, *case)


def test_igemm3_epilogues(sim, force_v3):
    kc.check_conv_dgrad(sim, (1, 64, 2, 9, 9), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), resid=True)
    kc.check_conv_fwd_fused(sim, (1, 64, 2, 9, 9), 72, (1, 3, 3), (1, 1, 1), (0, 1, 1), resid=True, relu=True)
    kc.check_conv_dgrad_bn(sim, (2, 64, 2, 9, 9), 64, (1, 3, 3), (0, 1, 1))
    kc.check_conv_dgrad_bn(sim, (2, 256, 2, 8, 8), 128, (1, 1, 1), (0, 0, 0), resid=True)
