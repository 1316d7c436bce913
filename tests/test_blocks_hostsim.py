"""CPU: fused block schedules (forward AND hand-written backward) through the host simulator vs the oracle."""
from tests import block_checks as bc


def test_resblock_projection_stride2_temporal(sim):
    bc.check_resblock(sim, 16, 32, 3, 2, 8, (2, 16, 2, 8, 8))


def test_resblock_identity(sim):
    bc.check_resblock(sim, 32, 32, 1, 1, 8, (2, 32, 2, 8, 8))


def test_resblock_dilated(sim):
    bc.check_resblock(sim, 16, 32, 1, 1, 8, (2, 16, 1, 8, 8), dilation=2)


def test_bn_part_tag_survives_only_unmodified_gradients(sim):
    bc.check_bn_part_tag_second_consumer(sim)


def test_stem_slow_and_fast(sim):
    bc.check_stem(sim, 16, [1, 7, 7], (1, 3, 2, 20, 20))
    bc.check_stem(sim, 8, [5, 7, 7], (1, 3, 4, 16, 16))


def test_fuse_fast_to_slow(sim):
    bc.check_fuse(sim, 8, 2, 5, 4, (1, 8, 8, 6, 6))


def test_bottleneck_transform_standalone(sim):
    bc.check_bottleneck_alone(sim, (2, 16, 2, 8, 8))


def test_x3d_se_squeeze_from_statistics_table(sim, monkeypatch):
    """The SE squeeze of an X3D block is read from the depthwise convolution's per-(sample, tile) statistics table (round 6:
    x3d.SE_FROM_STATS, sf_dwconv_fwd_sample_rows): no sf_sample_mean pass over the activation, and the block still matches the
    oracle; with the switch off the pass is back."""
    from slowfast_amd import x3d
    calls = []
    real = x3d.sample_mean
    monkeypatch.setattr(x3d, "sample_mean", lambda y, sc, sh, relu: (calls.append(relu), real(y, sc, sh, relu))[1])
    bc.check_x3d_block(sim, 24, 48, 2, 108, (4, 24, 4, 16, 16))              # SE block, stride 2, 108 -> 112 channels
    assert calls == []
    monkeypatch.setattr(x3d, "SE_FROM_STATS", False)
    bc.check_x3d_block(sim, 24, 48, 2, 108, (4, 24, 4, 16, 16))
    assert calls == [False]


def test_x3d_se_backward_in_one_pass(sim, monkeypatch):
    """SE blocks: sf_gate_bwd_sums (one pass over y and dz: du0 + per-sample sums) + sf_bn_bwd_apply_sample (the squeeze term added
    back as a per-sample constant) against the oracle, and the three-pass form (sf_gate_grad, sf_gate_act_bwd[_bn]) behind
    x3d.GATE_ONE_PASS = False; which entry points run is checked through the call observer."""
    from slowfast_amd import lib, x3d
    seen = []
    lib.set_call_observer(lambda name, thunk, work: (seen.append(name), thunk())[1])
    try:
        bc.check_x3d_block(sim, 24, 48, 2, 108, (4, 24, 4, 16, 16))
        assert "sf_gate_bwd_sums" in seen and "sf_bn_bwd_apply_sample" in seen and "sf_gate_grad" not in seen
        del seen[:]
        monkeypatch.setattr(x3d, "GATE_ONE_PASS", False)
        bc.check_x3d_block(sim, 24, 48, 2, 108, (4, 24, 4, 16, 16))
        assert "sf_gate_grad" in seen and "sf_gate_act_bwd_bn" in seen and "sf_gate_bwd_sums" not in seen
    finally:
        lib.set_call_observer(None)


def test_x3d_block_masks_handed(sim):
    bc.check_x3d_block(sim, 24, 48, 2, 108, (4, 24, 4, 16, 16))              # projection shortcut, stride 2, SE block
    bc.check_x3d_block(sim, 48, 48, 1, 108, (4, 48, 4, 8, 8), block_idx=1)   # identity shortcut, no SE


def test_nonlocal_block_routes_handed(sim):
    bc.check_nonlocal(sim, 64, 32, [1, 2, 2], (4, 64, 4, 8, 8), "softmax")
    bc.check_nonlocal(sim, 64, 32, [1, 2, 2], (4, 64, 4, 8, 8), "dot_product")


def test_multiscale_block(sim):
    bc.check_multiscale_block(sim, 96, 192, 2, (2, 8, 8), (1, 2, 2), (1, 2, 2))      # dim change + q pooling + max-pooled skip
    bc.check_multiscale_block(sim, 96, 96, 1, (2, 8, 8), (1, 1, 1), (1, 4, 4))
