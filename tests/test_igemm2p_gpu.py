"""GPU (-m gpu): the persistent implicit GEMM with drain waves (csrc/sf_igemm2p.h) on MI355X at production shapes.  Cases run
several times: the staging image is handed from the compute waves to the drain waves through workgroup barriers while the next
tile's copies are in flight -- a protocol error shows as a rare wrong tile."""
import pytest

from tests import kernel_checks as kc
from tests import token_checks as tc

pytestmark = pytest.mark.gpu


@pytest.fixture()
def force_p(monkeypatch):
    monkeypatch.setenv("SF_IGEMM2P", "1")


def test_igemm2p_token_gemms(gpu, force_p):
    for seed in range(3):
        tc.check_gemm(gpu, 50208, 384, 1152, seed=seed)             # MViTv2-S stage 3 qkv: 1773 tiles on 256 workgroups
        tc.check_gemm(gpu, 6273, 192, 576, seed=seed)
        tc.check_gemm_gelu(gpu, 12552, 384, 1536, seed=seed)        # fc1 + GELU, fc2 data gradient x gelu' + column sums
    tc.check_gemm(gpu, 394, 768, 3072, resid=False)
    tc.check_rows32(gpu, 4, 1569, 384, 384)
    tc.check_rows32(gpu, 4, 1569, 1536, 384)


CASES = [
    ((4, 64, 8, 56, 56), 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),       # res2 c (K = 64: two K steps)
    ((4, 256, 8, 14, 14), 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),      # res4 b
    ((4, 128, 8, 28, 28), 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),      # res3 c
    ((4, 1024, 8, 14, 14), 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),     # res4 a
]


@pytest.mark.parametrize("case", CASES)
def test_igemm2p_conv(gpu, force_p, case):
    for seed in range(2):
        kc.check_conv_fwd(gpu, *case, seed=seed)
        kc.check_conv_dgrad(gpu, *case, seed=seed)


def test_igemm2p_bn_chain(gpu, force_p):
    kc.check_bn_chain(gpu, (4, 64, 8, 28, 28))
