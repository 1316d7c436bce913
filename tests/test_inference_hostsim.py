"""CPU: the eval / multi-view test path on the host simulator -- eval-mode parity of every model
family against the probabilities the unmodified reference produced (tests/golden/eval_*.json), the inference-fused
schedule (BatchNorm folded into the convolutions), the fused-convolution kernel, and TestStep's device-side view
ensemble against a restatement of TestMeter.update_stats."""
import pytest
import torch

from tests import kernel_checks as kc
from tests import model_checks as mc


def test_conv_fwd_fused_kernel(sim):
    kc.check_conv_fwd_fused(sim, (1, 16, 2, 6, 6), 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), resid=True)       # direct-to-LDS
    kc.check_conv_fwd_fused(sim, (1, 16, 2, 7, 7), 16, (1, 3, 3), (1, 2, 2), (0, 1, 1), resid=False)
    kc.check_conv_fwd_fused(sim, (1, 32, 3, 5, 5), 72, (3, 1, 1), (1, 1, 1), (1, 0, 0), resid=True, relu=False)
    kc.check_conv_fwd_fused(sim, (1, 24, 1, 6, 6), 136, (1, 1, 1), (1, 2, 2), (0, 0, 0), resid=True, bias=False)


def test_conv_fwd_fused_thin_stem(sim):
    """W-pair-folded thin stems take the LDS-patch direct convolution with the bias / ReLU epilogue (sf_stem.h)."""
    kc.check_conv_fwd_fused(sim, (2, 8, 6, 36, 22), 8, (5, 7, 4), (1, 2, 1), (2, 3, 2))
    kc.check_conv_fwd_fused(sim, (1, 8, 3, 20, 20), 16, (1, 7, 4), (1, 2, 1), (0, 3, 2), relu=False)
    kc.check_conv_fwd_fused(sim, (1, 8, 9, 10, 9), 8, (3, 5, 4), (2, 1, 1), (1, 2, 2), bias=False)


@pytest.mark.parametrize("name", ["eval_slowfast_tiny", "eval_c2d_tiny", "eval_slowfast_nln_tiny", "eval_i3d_basic_tiny"])
@pytest.mark.parametrize("fused", [False, True])
def test_eval_resnet_family_matches_reference(sim, name, fused):
    """Running-statistics BatchNorm, fully-convolutional head (test crop > train crop), softmax + spatial mean; with
    ``fused`` every conv -> BN -> ReLU (+ residual) unit is one launch."""
    rep = {}
    try:
        mc.check_eval(name, sim, fused=fused, report=rep)
    finally:
        print(name, fused, rep.get(name))


@pytest.mark.parametrize("name", ["eval_x3d_tiny", "eval_mvit_tiny"])
def test_eval_x3d_mvit_match_reference(sim, name):
    """X3D: sliding-window head at the larger test crop; MViTv2: softmax head."""
    rep = {}
    try:
        mc.check_eval(name, sim, fused=True, report=rep)      # fuse_for_inference is a no-op for these families
    finally:
        print(name, rep.get(name))


def test_fused_path_is_left_in_training_mode(sim):
    """model.train() after fuse_for_inference runs the training schedule again (batch statistics, autograd)."""
    import slowfast_amd as sa
    from slowfast_amd import inference
    gold = mc.load_golden("slowfast_tiny")
    cfg = mc.cfg_for(gold)
    torch.manual_seed(0)
    model = sa.MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)
    from oracle import video_ref
    inputs, labels = video_ref.synthetic_batch(cfg, 2, 5)
    inference.fuse_for_inference(model)
    assert model.__dict__["_sf_fused_modules"] > 10
    with torch.no_grad():
        p = model(inputs)
    assert float((p.sum(1) - 1).abs().max()) < 1e-3
    model.train()
    logits = model(inputs)
    torch.nn.functional.cross_entropy(logits.float(), labels).backward()
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in model.parameters())
    inference.unfuse(model)
    assert "_sf_fused_modules" not in model.__dict__


def _meter_restatement(num_videos, num_clips, num_cls, batches, method):
    """TestMeter.update_stats (slowfast/utils/meters.py:305-336) restated with its Python loop."""
    vp = torch.zeros((num_videos, num_cls))
    vl = torch.zeros((num_videos,), dtype=torch.long)
    cnt = torch.zeros((num_videos,), dtype=torch.long)
    for preds, labels, clip_ids in batches:
        for i in range(preds.shape[0]):
            v = int(clip_ids[i]) // num_clips
            vl[v] = labels[i]
            if method == "sum":
                vp[v] += preds[i]
            else:
                vp[v] = torch.max(vp[v], preds[i])
            cnt[v] += 1
    return vp, vl, cnt


@pytest.mark.parametrize("method", ["sum", "max"])
def test_view_ensemble_matches_testmeter(method):
    from slowfast_amd.inference import TestStep

    class _Dummy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))

        def forward(self, x):
            return x[0]

    g = torch.Generator().manual_seed(0)
    V, K, C = 5, 6, 7
    ids = torch.randperm(V * K, generator=g)
    labels_of = torch.randint(0, C, (V,), generator=g)
    step = TestStep(_Dummy(), V, K, C, ensemble_method=method, use_graph=False)
    batches = []
    for chunk in ids.split(4):
        preds = torch.softmax(torch.randn((chunk.numel(), C), generator=g), 1)
        labels = labels_of[chunk // K]
        batches.append((preds, labels, chunk))
        step.step([preds], labels, chunk)
    vp, vl, cnt = _meter_restatement(V, K, C, batches, method)
    assert torch.allclose(step.video_preds, vp, atol=1e-6) and torch.equal(step.video_labels, vl)
    assert torch.equal(step.clip_count, cnt)
    stats = step.finalize()
    top1 = float((vp.argmax(1) == vl).float().mean() * 100)
    assert abs(stats["top1_acc"] - top1) < 1e-4 and stats["all_clips_seen"]


def test_fused_eval_with_roi_head_and_nonlocal(sim):
    """Detection models (SlowFast + Nonlocal + ResNetRoIHead on boxes): the inference-fused backbone feeds the RoI head the
    same features as the running-statistics schedule, and both agree with the oracle's eval forward."""
    import slowfast_amd as sa
    from oracle import video_ref
    from slowfast_amd import inference
    gold = mc.load_golden("slowfast_ava_roi_tiny")
    cfg = mc.cfg_for(gold)
    model = sa.MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)
    sd = video_ref.randomize_state({k: tuple(v.shape) for k, v in model.state_dict().items()}, gold["param_seed"])
    video_ref.scale_final_bn(sd, gold["state_tweaks"]["final_bn_gamma_scale"])
    inputs, _ = video_ref.synthetic_batch(cfg, 2, gold["data_seed"])
    boxes = video_ref.synthetic_boxes(cfg, 2, seed=77, per_clip=2)
    sd = video_ref.calibrate_running_stats(sd, cfg, inputs, bboxes=boxes)
    with torch.no_grad():
        ref = video_ref.video_forward(sd, cfg, inputs, training=False, bboxes=boxes)
    model.load_state_dict(sd)
    model.eval()
    with torch.no_grad():
        plain = model(inputs, boxes).float()
        inference.fuse_for_inference(model)
        fused = model(inputs, boxes).float()
    assert plain.shape == ref.shape == fused.shape == (4, cfg.MODEL.NUM_CLASSES)
    scale = float(ref.abs().max())
    assert float((plain - ref).abs().max()) < 2e-2 * scale and float((fused - ref).abs().max()) < 2e-2 * scale
    assert float((fused - plain).abs().max()) < 2e-2 * scale
