"""CPU (host simulator + the reference tree): the drop-in models driven by the REFERENCE's own code, unchanged.

SURVEY.md 8b / VERDICT r1 item 9: register the drop-ins in the reference's MODEL_REGISTRY (slowfast/models/build.py:13-19) and run
the reference's ``build_model(cfg)``, ``init_weights``, ``construct_optimizer``, ``get_loss_func``, ``get_epoch_lr`` /
``set_lr`` and one iteration shaped like ``train_epoch`` (tools/train_net.py:104-172) on them.  Needs /root/reference
(oracle/refshim.py stubs only the un-vendored third-party imports); skipped where the tree is absent (the GPU box)."""
import pytest
import torch

from oracle import refshim

pytestmark = pytest.mark.skipif(not refshim.available(), reason="reference tree not present")

OPTS = ["NUM_GPUS", 0, "MODEL.DROPOUT_RATE", 0.0, "MODEL.NUM_CLASSES", 10, "DATA.TRAIN_CROP_SIZE", 32,
        "RESNET.WIDTH_PER_GROUP", 16, "RESNET.DEPTH", 18, "DATA.NUM_FRAMES", 8, "SLOWFAST.BETA_INV", 2,
        "RESNET.NUM_BLOCK_TEMP_KERNEL", "[[2, 2], [2, 2], [2, 2], [2, 2]]", "TRAIN.BATCH_SIZE", 2,
        "SOLVER.BASE_LR", 0.01, "SOLVER.WARMUP_EPOCHS", 0.0, "BN.WEIGHT_DECAY", 0.0]


@pytest.fixture()
def reference_with_drop_ins(sim):
    """The reference's registry with its SlowFast / ResNet entries replaced by the drop-ins."""
    refshim.install()
    from slowfast.models.build import MODEL_REGISTRY
    import slowfast_amd.video_models as amd
    table = MODEL_REGISTRY._obj_map if hasattr(MODEL_REGISTRY, "_obj_map") else MODEL_REGISTRY._obj   # fvcore / the shim
    saved = dict(table)
    table["SlowFast"] = amd.SlowFast
    table["ResNet"] = amd.ResNet
    yield
    table.clear()
    table.update(saved)


def test_reference_build_model_init_optimizer_and_iteration(reference_with_drop_ins):
    import slowfast.models.losses as losses
    import slowfast.models.optimizer as optim
    import slowfast.utils.weight_init_helper as init_helper
    from slowfast.models.build import build_model
    import slowfast_amd.video_models as amd
    from oracle import video_ref

    cfg = refshim.reference_cfg("configs/Kinetics/SLOWFAST_8x8_R50.yaml", OPTS)
    torch.manual_seed(0)
    model = build_model(cfg)                                    # slowfast/models/build.py:22-81, unchanged
    assert isinstance(model, amd.SlowFast), type(model)

    # weight init exactly as the reference's constructors apply it (video_model_builder.py:214-219)
    init_helper.init_weights(model, cfg.MODEL.FC_INIT_STD, cfg.RESNET.ZERO_INIT_FINAL_BN, cfg.RESNET.ZERO_INIT_FINAL_CONV)
    finals = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm3d) and getattr(m, "transform_final_bn", False)]
    assert finals and all(float(m.weight.abs().max()) == 0.0 for m in finals)      # ZERO_INIT_FINAL_BN reached the markers
    assert abs(float(model.head.projection.weight.std()) - cfg.MODEL.FC_INIT_STD) < 0.3 * cfg.MODEL.FC_INIT_STD

    # state_dict contract against the reference's own model of the same cfg
    from slowfast.models.video_model_builder import SlowFast as RefSlowFast
    ref_model = RefSlowFast(cfg)
    ref_sd = ref_model.state_dict()
    sd = model.state_dict()
    assert list(sd) == list(ref_sd) and all(sd[k].shape == ref_sd[k].shape for k in sd)
    model.load_state_dict(ref_sd)                               # strict

    # the reference's optimizer construction: parameter grouping by isinstance(_NormBase) etc. (optimizer.py:15-140)
    optimizer = optim.construct_optimizer(model, cfg)
    assert sum(len(g["params"]) for g in optimizer.param_groups) == len(list(model.parameters()))
    loss_fun = losses.get_loss_func(cfg.MODEL.LOSS_FUNC)(reduction="mean")
    ref_opt = optim.construct_optimizer(ref_model, cfg)

    # one iteration shaped like train_epoch (tools/train_net.py:104-172), on both models
    inputs, labels = video_ref.synthetic_batch(cfg, 2, seed=3)
    model.train()
    ref_model.train()
    for gamma_fix in (model, ref_model):                        # non-zero final gammas so that every layer trains
        for m in gamma_fix.modules():
            if isinstance(m, torch.nn.BatchNorm3d) and getattr(m, "transform_final_bn", False):
                torch.nn.init.constant_(m.weight, 0.5)
    stats = []
    for net, opt in ((model, optimizer), (ref_model, ref_opt)):
        lr = optim.get_epoch_lr(0.0, cfg)
        optim.set_lr(opt, lr)
        opt.zero_grad()
        preds = net([x.clone() for x in inputs])
        loss = loss_fun(preds.float(), labels)
        loss.backward()
        grad_norm = optim.get_grad_norm_(net.parameters())
        opt.step()
        stats.append((float(loss), float(grad_norm), preds.detach().float()))
    (l0, g0, p0), (l1, g1, p1) = stats
    # this 2-clip 32x32 miniature is ill-conditioned (BatchNorm over a handful of samples): wiring-level agreement
    assert abs(l0 - l1) < 0.05 * abs(l1) and abs(g0 - g1) < 0.5 * g1, stats
    assert float((p0 - p1).abs().max()) < 0.2 * float(p1.abs().max()) + 1e-3
    # both optimizers moved the same parameters
    moved = sum(int(not torch.equal(a, b)) for a, b in zip(model.state_dict().values(), ref_sd.values()))
    assert moved > 50


def test_reference_build_model_resnet(reference_with_drop_ins):
    from slowfast.models.build import build_model
    import slowfast_amd.video_models as amd
    cfg = refshim.reference_cfg("configs/Kinetics/C2D_8x8_R50.yaml",
                                ["NUM_GPUS", 0, "MODEL.DROPOUT_RATE", 0.0, "MODEL.NUM_CLASSES", 10, "DATA.TRAIN_CROP_SIZE", 32,
                                 "RESNET.WIDTH_PER_GROUP", 16, "RESNET.DEPTH", 18, "DATA.NUM_FRAMES", 4,
                                 "RESNET.NUM_BLOCK_TEMP_KERNEL", "[[2], [2], [2], [2]]", "TRAIN.BATCH_SIZE", 2])
    model = build_model(cfg)
    assert isinstance(model, amd.ResNet)
    from oracle import video_ref
    inputs, labels = video_ref.synthetic_batch(cfg, 2, seed=4)
    out = model.eval()(inputs)
    assert out.shape == (2, 10) and torch.isfinite(out).all()
