"""CPU: fused block schedules (forward AND hand-written backward) through the host simulator vs the oracle."""
from tests import block_checks as bc


def test_resblock_projection_stride2_temporal(sim):
    bc.check_resblock(sim, 16, 32, 3, 2, 8, (2, 16, 2, 8, 8))


def test_resblock_identity(sim):
    bc.check_resblock(sim, 32, 32, 1, 1, 8, (2, 32, 2, 8, 8))


def test_resblock_dilated(sim):
    bc.check_resblock(sim, 16, 32, 1, 1, 8, (2, 16, 1, 8, 8), dilation=2)


def test_stem_slow_and_fast(sim):
    bc.check_stem(sim, 16, [1, 7, 7], (1, 3, 2, 20, 20))
    bc.check_stem(sim, 8, [5, 7, 7], (1, 3, 4, 16, 16))


def test_fuse_fast_to_slow(sim):
    bc.check_fuse(sim, 8, 2, 5, 4, (1, 8, 8, 6, 6))


def test_bottleneck_transform_standalone(sim):
    bc.check_bottleneck_alone(sim, (2, 16, 2, 8, 8))
