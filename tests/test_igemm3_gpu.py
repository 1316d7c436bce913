"""GPU (-m gpu): the third-generation implicit GEMM (csrc/sf_igemm3.h) on MI355X at production geometries, against F.conv3d on
identical fp16 operands.  Every case runs several times: the eight-phase schedule keeps copies in flight across barriers, a
landing-order hazard shows as a rare wrong tile."""
import pytest

from tests import kernel_checks as kc

pytestmark = pytest.mark.gpu


@pytest.fixture()
def force_v3(monkeypatch):
    monkeypatch.setenv("SF_IGEMM3", "1")


CASES = [
    ((4, 256, 8, 14, 14), 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),      # res4 b
    ((4, 1024, 8, 14, 14), 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),     # res4 a (48 K tiles)
    ((4, 512, 8, 7, 7), 512, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),        # res5 b
    ((2, 128, 8, 28, 28), 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),      # res3 c: two K tiles
    ((1, 384, 1, 40, 40), 1536, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),     # MViT fc1-like: 6 K tiles, 6 N tiles, ragged M
]


@pytest.mark.parametrize("case", CASES)
def test_igemm3_fwd_dgrad(gpu, force_v3, case):
    for seed in range(3):
        kc.check_conv_fwd(gpu, *case, seed=seed)
        kc.check_conv_dgrad(gpu, *case, seed=seed)


def test_igemm3_epilogues(gpu, force_v3):
    kc.check_conv_dgrad_bn(gpu, (4, 256, 8, 14, 14), 256, (1, 3, 3), (0, 1, 1))
    kc.check_conv_dgrad_bn(gpu, (4, 256, 8, 14, 14), 1024, (1, 1, 1), (0, 0, 0), resid=True)
    kc.check_conv_fwd_fused(gpu, (4, 256, 8, 14, 14), 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), resid=True, relu=True)
