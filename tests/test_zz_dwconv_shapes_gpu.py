"""GPU (-m gpu): larger shape sweeps of the depthwise stencils than the default kernel tests carry (a file that sorts last: with
``pytest -x`` the model-level results come first)."""
import pytest

pytestmark = pytest.mark.gpu


def test_depthwise_blocked_stencils_gpu(gpu):
    """The W-blocked depthwise stencils on the real kernels: narrow (fp32 LDS weights) and wide (fp16) layers, stride 1 and 2, whole
    and ragged 4-column groups, cls rows."""
    from tests import token_checks as tc
    d = gpu
    tc.check_dwconv(d, 2, 2, 96, (4, 14, 14), (3, 3, 3), (1, 2, 2), cls=1)
    tc.check_dwconv(d, 1, 4, 96, (4, 7, 7), (3, 3, 3), (1, 1, 1), cls=1)
    tc.check_dwconv(d, 2, 1, 24, (8, 12, 12), (5, 1, 1), (1, 1, 1), cls=0)
    tc.check_dwconv(d, 2, 1, 216, (4, 14, 14), (3, 3, 3), (1, 2, 2), cls=0)
    tc.check_dwconv(d, 2, 1, 56, (4, 28, 28), (3, 3, 3), (1, 1, 1), cls=0)
    tc.check_dwconv(d, 2, 1, 56, (4, 56, 56), (3, 3, 3), (1, 2, 2), cls=0)
    tc.check_dwconv(d, 1, 1, 432, (4, 8, 8), (3, 3, 3), (1, 1, 1), cls=0)
