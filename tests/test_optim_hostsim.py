"""CPU (host simulator): the flat-buffer training-step glue (csrc/sf_optim.h, slowfast_amd/optim.py) against torch.optim and
torch.amp.GradScaler semantics -- tools/train_net.py:150-172 + slowfast/models/optimizer.py:100-140 of the reference."""
import math

import pytest
import torch

from slowfast_amd.data_parallel import GradReducer
from slowfast_amd.optim import CTL_SKIPPED, CTL_STEPS, FlatOptimizer


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.a = torch.nn.Parameter(torch.randn(37, 19, generator=g))            # 703 elements: one ragged block
        self.b = torch.nn.Parameter(torch.randn(2100, generator=g))              # three blocks, ragged tail
        self.c = torch.nn.Parameter(torch.randn(5, 4, 3, generator=g))
        self.bn = torch.nn.BatchNorm1d(16)


def _pair():
    net, ref = _Net(), _Net()
    ref.load_state_dict(net.state_dict())
    return net, ref


def _groups(m, lr):
    return [{"params": [m.a, m.c], "weight_decay": 1e-2, "lr": lr},
            {"params": [m.b], "weight_decay": 0.0, "lr": lr},
            {"params": list(m.bn.parameters()), "weight_decay": 5e-3, "lr": lr}]


def _set_grads(net, ref, red, seed, scale, world=1.0, poison=False):
    g = torch.Generator().manual_seed(seed)
    red.zero_grad()
    for (k, p), q in zip(net.named_parameters(), ref.parameters()):
        gr = torch.randn(p.shape, generator=g)
        q.grad = gr.clone()
        p.grad.copy_(gr * scale * world)                 # what backward + a SUM all-reduce over `world` ranks would leave
    if poison:
        net.b.grad[17] = float("inf")


@pytest.mark.parametrize("nesterov,dampening", [(True, 0.0), (False, 0.1)])
def test_flat_sgd_matches_torch(sim, nesterov, dampening):
    net, ref = _pair()
    red = GradReducer(net)
    opt = FlatOptimizer(_groups(net, 0.1), red, method="sgd", momentum=0.9, dampening=dampening, nesterov=nesterov,
                        loss_scale=128.0)
    topt = torch.optim.SGD(_groups(ref, 0.1), momentum=0.9, dampening=dampening, nesterov=nesterov)
    for it in range(4):
        lr = 0.1 * (0.5 ** it)                           # the reference sets the lr per iteration (optimizer.set_lr)
        for g1, g2 in zip(opt.param_groups, topt.param_groups):
            g1["lr"] = g2["lr"] = lr
        _set_grads(net, ref, red, 10 + it, 128.0)
        red.finish(loss_scale=None)
        opt.step()
        topt.step()
        for (k, p), q in zip(net.named_parameters(), ref.parameters()):
            assert torch.allclose(p.data, q.data, rtol=2e-6, atol=2e-7), (it, k)
        gn = math.sqrt(sum(float(q.grad.double().pow(2).sum()) for q in ref.parameters()))
        assert abs(float(opt.grad_norm) - gn) < 1e-5 * gn
    assert float(opt.ctl[CTL_STEPS]) == 4 and float(opt.ctl[CTL_SKIPPED]) == 0
    red.close()


def test_flat_adamw_clipping_and_world(sim):
    """AdamW + clip_grad_norm_ on gradients that arrive as a loss-scaled SUM over 4 ranks."""
    net, ref = _pair()
    red = GradReducer(net)
    red.world = 4                                        # pretend: the buffer holds a sum over 4 ranks
    opt = FlatOptimizer(_groups(net, 3e-3), red, method="adamw", loss_scale=64.0, clip_grad_l2norm=1.0)
    topt = torch.optim.AdamW(_groups(ref, 3e-3), betas=(0.9, 0.999), eps=1e-8)
    for it in range(3):
        _set_grads(net, ref, red, 20 + it, 64.0, world=4.0)
        red.finish(loss_scale=None)
        opt.step()
        torch.nn.utils.clip_grad_norm_(list(ref.parameters()), 1.0)
        topt.step()
        for (k, p), q in zip(net.named_parameters(), ref.parameters()):
            assert torch.allclose(p.data, q.data, rtol=1e-5, atol=1e-6), (it, k)
    red.world = 1
    red.close()


def test_bucketwise_norm_pass_matches_whole_buffer(sim):
    """finish_and_step(): the norm / overflow pass taken bucket by bucket (as each bucket's all-reduce completes) leaves the
    same control block and parameters as the one-pass step(); tiny buckets make every bucket start unaligned."""
    for poison in (False, True):
        outs = []
        for bucketed in (False, True):
            net, ref = _pair()
            for m in (net, ref):                         # 7 elements ahead of everything else in the flat buffer
                m.d = torch.nn.Parameter(torch.linspace(-1, 1, 7))
            red = GradReducer(net, bucket_mb=4e-5 if bucketed else 48)            # 10 elements: a bucket per parameter
            assert (len(red.buckets) > 3) == bucketed
            if bucketed:
                assert any(s0 % 4 for s0, _, _ in red.buckets)
            groups = _groups(net, 0.1)
            groups[1]["params"].append(net.d)
            opt = FlatOptimizer(groups, red, method="sgd", momentum=0.9, nesterov=True, loss_scale=32.0,
                                clip_grad_l2norm=0.5, dynamic_loss_scale=True)
            seen = []
            for it in range(2):
                _set_grads(net, ref, red, 50 + it, 32.0, poison=poison and it == 1)
                if bucketed:
                    sumsq = opt._sumsq_bucket
                    opt._sumsq_bucket = lambda bi, f=sumsq: (seen.append(bi), f(bi))[1]
                    opt.finish_and_step()
                    opt._sumsq_bucket = sumsq
                else:
                    red.finish(loss_scale=None)
                    opt.step()
            if bucketed:
                assert sorted(seen) == sorted(2 * list(range(len(red.buckets))))
            outs.append(([p.data.clone() for p in net.parameters()], opt.ctl.clone()))
            red.close()
        for a, b in zip(outs[0][0], outs[1][0]):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
        ca, cb = outs[0][1], outs[1][1]
        fin = torch.isfinite(ca)
        assert torch.equal(fin, torch.isfinite(cb)) and torch.allclose(ca[fin], cb[fin], rtol=1e-6)
        assert float(cb[2]) == float(poison)


def test_flat_sgd_clip_value(sim):
    net, ref = _pair()
    red = GradReducer(net)
    opt = FlatOptimizer(_groups(net, 0.05), red, method="sgd", momentum=0.0, clip_grad_val=0.3)
    topt = torch.optim.SGD(_groups(ref, 0.05), momentum=0.0)
    _set_grads(net, ref, red, 5, 1.0)
    red.finish(loss_scale=None)
    opt.step()
    torch.nn.utils.clip_grad_value_(list(ref.parameters()), 0.3)
    topt.step()
    for (k, p), q in zip(net.named_parameters(), ref.parameters()):
        assert torch.allclose(p.data, q.data, rtol=2e-6, atol=2e-7), k
    red.close()


def test_dynamic_loss_scale_skips_and_recovers(sim):
    """GradScaler semantics on the device: an inf / NaN gradient skips the update (parameters and momentum untouched),
    halves the scale and resets the growth tracker; `growth_interval` clean steps double it (train_net.py:152-172;
    misc.check_nan_losses' job is done by the same flag)."""
    net, ref = _pair()
    red = GradReducer(net)
    opt = FlatOptimizer(_groups(net, 0.1), red, method="sgd", momentum=0.9, nesterov=True, loss_scale=1024.0,
                        dynamic_loss_scale=True, growth_interval=2)
    topt = torch.optim.SGD(_groups(ref, 0.1), momentum=0.9, nesterov=True)
    scales = []
    for it, poison in enumerate([False, True, False, False, False]):
        scale = float(opt.loss_scale)                    # (the training loop never reads it: it multiplies on the device)
        scales.append(scale)
        before = [p.data.clone() for p in net.parameters()]
        _set_grads(net, ref, red, 30 + it, scale, poison=poison)
        red.finish(loss_scale=None)
        opt.step()
        if poison:
            assert float(opt.found_inf) == 1.0 and math.isinf(float(opt.grad_norm))
            for p, b in zip(net.parameters(), before):
                assert torch.equal(p.data, b)
        else:
            topt.step()
            for (k, p), q in zip(net.named_parameters(), ref.parameters()):
                assert torch.allclose(p.data, q.data, rtol=2e-6, atol=2e-7), (it, k)
    # 1024 -> clean (tracker 1) -> overflow: 512 -> clean, clean: 1024 -> clean (tracker 1)
    assert scales == [1024.0, 1024.0, 512.0, 512.0, 1024.0], scales
    assert float(opt.ctl[CTL_STEPS]) == 4 and float(opt.ctl[CTL_SKIPPED]) == 1
    sd = opt.state_dict()
    opt.load_state_dict(sd)
    red.close()


def test_train_step_with_flat_optimizer(sim):
    """TrainStep drives FlatOptimizer: the loss is scaled by the device-side scale, statistics are queued without a sync."""
    import slowfast_amd as sa
    from slowfast_amd.optim import construct_optimizer
    from slowfast_amd.step import TrainStep
    from tests import model_checks as mc
    gold = mc.load_golden("c2d_tiny")
    cfg = mc.cfg_for(gold)
    model, sd, inputs, labels, *_ = mc.oracle_run(gold, cfg)
    model.load_state_dict(sd)
    model.train()
    red = GradReducer(model)
    red.attach_torch_param_hooks(model.head.parameters())
    opt = construct_optimizer(model, cfg, red, loss_scale=256.0, dynamic_loss_scale=True)
    for g in opt.param_groups:                           # optim.set_lr(): a learning rate this 2-clip miniature tolerates
        g["lr"] = 0.01
    step = TrainStep(model, red, opt, torch.nn.functional.cross_entropy, use_graph=False, track_stats=True)
    w0 = model.s1.pathway0_stem.conv.weight.detach().clone()
    losses = [float(step(inputs, labels)) for _ in range(3)]
    assert all(math.isfinite(v) for v in losses) and losses[2] < losses[0]
    assert not torch.equal(model.s1.pathway0_stem.conv.weight.detach(), w0)
    assert model.s1.pathway0_stem.conv.weight.data_ptr() >= opt.flat_param.data_ptr()      # parameters live in the flat buffer
    st = step.pop_stats()
    assert st is not None and len(st) == 4 and abs(st[0] - losses[0]) < 1e-5 and st[1] > 0 and 0.0 <= st[2] <= 100.0
    red.close()
