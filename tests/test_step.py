"""TrainStep (slowfast_amd/step.py): the captured-graph iteration must be the same computation as the eager
sequence zero_grad -> forward -> loss -> backward -> finish -> optimizer.step (tools/train_net.py:104-172)."""
import copy

import contextlib

import pytest
import torch
import torch.nn.functional as F

from tests.kernel_checks import host_to_cl
from tests.test_data_parallel import _build


def _run(device, use_graph, steps, seed=0, force_collectives=False, bucket_mb=48):
    from slowfast_amd.data_parallel import GradReducer
    from slowfast_amd.step import TrainStep
    torch.manual_seed(seed)
    net = _build().to(device).train()
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
    red = GradReducer(net, bucket_mb=bucket_mb, force_collectives=force_collectives)
    red.attach_torch_param_hooks(net.fc.parameters())
    step = TrainStep(net, red, opt, F.cross_entropy, loss_scale=8.0, use_graph=use_graph, warmup=1)
    g = torch.Generator().manual_seed(5)
    losses = []
    for i in range(steps):
        x = host_to_cl(torch.randn((4, 16, 2, 8, 8), generator=g), device)
        y = torch.randint(0, 5, (4,), generator=g).to(device)
        losses.append(float(step(x, y)))
    red.close()
    return losses, [p.detach().float().cpu().clone() for p in net.parameters()], \
        [b.detach().float().cpu().clone() for b in net.buffers()]


def test_train_step_eager_matches_manual_loop(sim):
    from slowfast_amd.data_parallel import GradReducer
    losses, params, bufs = _run(sim, use_graph=False, steps=3)
    torch.manual_seed(0)
    net = _build().train()
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
    red = GradReducer(net)
    red.attach_torch_param_hooks(net.fc.parameters())
    g = torch.Generator().manual_seed(5)
    for i in range(3):
        x = host_to_cl(torch.randn((4, 16, 2, 8, 8), generator=g), sim)
        y = torch.randint(0, 5, (4,), generator=g)
        red.zero_grad()
        loss = F.cross_entropy(net(x).float(), y)
        (loss * 8.0).backward()
        red.finish(loss_scale=8.0)
        opt.step()
        assert abs(float(loss) - losses[i]) < 1e-6
    for p, q in zip(net.parameters(), params):
        assert torch.equal(p.detach(), q)
    red.close()
    assert losses[0] != losses[1]


@pytest.mark.gpu
def test_train_step_graph_replay_matches_eager(gpu):
    """HIP-graph capture + replay (with fresh inputs copied into the static buffers, weights re-packed inside
    the graph, BN running statistics updated by replayed kernels) == the eager iteration, bit for bit."""
    le, pe, be = _run(gpu, use_graph=False, steps=5)
    lg, pg, bg = _run(gpu, use_graph=True, steps=5)
    assert le == lg, (le, lg)
    for a, b in zip(pe, pg):
        assert torch.equal(a, b)
    for a, b in zip(be, bg):
        assert torch.equal(a, b)


_RCCL_SCRIPT = """
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from tests.test_step import _run
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ref = {g: _run(dev, use_graph=g, steps=4) for g in (False, True)}
dist.init_process_group(backend="nccl", device_id=dev)
for g in (False, True):
    got = _run(dev, use_graph=g, steps=4, force_collectives=True, bucket_mb=0.002)
    assert got[0] == ref[g][0], (got[0], ref[g][0])
    for a, b in zip(got[1] + got[2], ref[g][1] + ref[g][2]):
        assert torch.equal(a, b)
dist.destroy_process_group()
print("rccl-ok")
"""


@pytest.mark.gpu
def test_train_step_with_rccl_collectives(gpu):
    """The bucketed all-reduce path on RCCL (backend "nccl", one rank: SUM over one rank is the identity) interleaved
    with the eager backward and after a graph replay gives the same parameters, bit for bit, as the run without
    collectives -- i.e. stream ordering between the compute stream, the captured graph and RCCL's stream is right."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_train_step_gradient_clipping(sim):
    """SOLVER.CLIP_GRAD_L2NORM / CLIP_GRAD_VAL as in tools/train_net.py:156-166, computed on the flat gradient buffer:
    same parameters as torch.nn.utils.clip_grad_norm_ / clip_grad_value_ on the per-parameter gradients."""
    from slowfast_amd.data_parallel import GradReducer
    from slowfast_amd.step import TrainStep
    for kind in ("norm", "value"):
        results = []
        for native in (True, False):
            torch.manual_seed(0)
            net = _build().train()
            opt = torch.optim.SGD(net.parameters(), lr=0.05)
            red = GradReducer(net)
            red.attach_torch_param_hooks(net.fc.parameters())
            g = torch.Generator().manual_seed(5)
            x = host_to_cl(torch.randn((4, 16, 2, 8, 8), generator=g), sim)
            y = torch.randint(0, 5, (4,), generator=g)
            if native:
                step = TrainStep(net, red, opt, F.cross_entropy, loss_scale=4.0, use_graph=False,
                                 clip_grad_l2norm=0.05 if kind == "norm" else None,
                                 clip_grad_val=0.002 if kind == "value" else None)
                step(x, y)
                assert step.grad_norm is not None and float(step.grad_norm) > 0
            else:
                red.zero_grad()
                (F.cross_entropy(net(x).float(), y) * 4.0).backward()
                red.finish(loss_scale=4.0)
                if kind == "norm":
                    total = torch.nn.utils.clip_grad_norm_(net.parameters(), 0.05)
                    assert float(total) > 0.05, "the test must actually clip"
                else:
                    torch.nn.utils.clip_grad_value_(net.parameters(), 0.002)
                opt.step()
            results.append([p.detach().clone() for p in net.parameters()])
            red.close()
        for a, b in zip(*results):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)


def _run_flat(device, use_graph, steps, poison_step=None):
    """TrainStep + FlatOptimizer (fused unscale / norm / clip / SGD-Nesterov with a dynamic loss scale on the device)."""
    from slowfast_amd.data_parallel import GradReducer
    from slowfast_amd.optim import FlatOptimizer
    from slowfast_amd.step import TrainStep
    torch.manual_seed(0)
    net = _build().to(device).train()
    red = GradReducer(net)
    red.attach_torch_param_hooks(net.fc.parameters())
    bn = [p for m in net.modules() if isinstance(m, torch.nn.modules.batchnorm._NormBase) for p in m.parameters(recurse=False)]
    rest = [p for p in net.parameters() if all(p is not q for q in bn)]
    opt = FlatOptimizer([{"params": bn, "weight_decay": 0.0, "lr": 0.05}, {"params": rest, "weight_decay": 1e-4, "lr": 0.05}],
                        red, method="sgd", momentum=0.9, nesterov=True, loss_scale=64.0, dynamic_loss_scale=True,
                        growth_interval=3, clip_grad_l2norm=5.0)
    step = TrainStep(net, red, opt, F.cross_entropy, use_graph=use_graph, warmup=1, track_stats=True)
    g = torch.Generator().manual_seed(5)
    losses = []
    for i in range(steps):
        x = host_to_cl(torch.randn((4, 16, 2, 8, 8), generator=g), device)
        if i == poison_step:
            x = x * float("inf")                    # forces non-finite gradients: the step must be skipped
        y = torch.randint(0, 5, (4,), generator=g).to(device)
        losses.append(float(step(x, y)))
    out = ([p.detach().float().cpu().clone() for p in net.parameters()], opt.ctl.detach().cpu().clone(), losses)
    red.close()
    return out


def test_flat_optimizer_train_step_skips_overflow(sim):
    params, ctl, losses = _run_flat(sim, use_graph=False, steps=4, poison_step=2)
    from slowfast_amd.optim import CTL_SCALE, CTL_SKIPPED, CTL_STEPS
    assert float(ctl[CTL_SKIPPED]) == 1 and float(ctl[CTL_STEPS]) == 3
    assert float(ctl[CTL_SCALE]) == 32.0            # 64 -> overflow at step 2: 32 (growth_interval 3 not reached again)
    assert all(torch.isfinite(p).all() for p in params)


@pytest.mark.gpu
def test_flat_optimizer_graph_replay_matches_eager(gpu):
    """The dynamic loss scale is a device scalar read inside the captured graph: graph replay == eager bit for bit, through an
    overflow (skipped step, halved scale) and a scale growth."""
    pe, ce, le = _run_flat(gpu, use_graph=False, steps=6, poison_step=3)
    pg, cg, lg = _run_flat(gpu, use_graph=True, steps=6, poison_step=3)
    assert torch.equal(ce, cg), (ce, cg)
    assert [a for a in le if a == a] == [a for a in lg if a == a]
    for a, b in zip(pe, pg):
        assert torch.equal(a, b)
    from slowfast_amd.optim import CTL_SKIPPED
    assert float(ce[CTL_SKIPPED]) == 1


def _run_model(device, name, segmented, use_graph, steps=2):
    """TrainStep on a golden-case model (its forward marks stage boundaries with engine.cut)."""
    from slowfast_amd.data_parallel import GradReducer
    from slowfast_amd.optim import construct_optimizer
    from slowfast_amd.step import TrainStep
    from tests import model_checks as mc
    gold = mc.load_golden(name)
    cfg = mc.cfg_for(gold)
    model, sd, inputs, labels, *_ = mc.oracle_run(gold, cfg)
    model.load_state_dict(sd)
    model = model.to(device).train()
    red = GradReducer(model, bucket_mb=0.05)
    red.attach_torch_param_hooks(model.head.parameters())
    opt = construct_optimizer(model, cfg, red, loss_scale=64.0, dynamic_loss_scale=False)
    for g in opt.param_groups:
        g["lr"] = 0.01
    step = TrainStep(model, red, opt, F.cross_entropy, use_graph=use_graph, warmup=1, segmented=segmented)
    xs, ys = [x.to(device) for x in inputs], labels.to(device)
    losses = [float(step(xs, ys)) for _ in range(steps)]
    out = (losses, [p.detach().float().cpu().clone() for p in model.parameters()], len(step._seg_params), list(step.overlap_log))
    red.close()
    return out


@pytest.mark.parametrize("name", [pytest.param("slowfast_tiny", marks=pytest.mark.slow), "mvit_tiny", "x3d_tiny", "c2d_tiny"])
def test_segmented_backward_equals_unsegmented(sim, name):
    """Backward run stage by stage across engine.cut() boundaries == one backward pass: same losses, same parameters.
    (x3d / c2d: their stages end in a residual block, whose fused BatchNorm-backward partial sums ride on the gradient tensor
    object -- engine.retag_cut_grad carries them across the cut.)"""
    l0, p0, n0, _ = _run_model(sim, name, segmented=False, use_graph=False)
    l1, p1, n1, _ = _run_model(sim, name, segmented=True, use_graph=False)
    assert n1 >= 3, "the model must expose at least two stage boundaries"
    assert l0 == l1, (l0, l1)
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["slowfast_tiny", "mvit_tiny", "x3d_tiny"])
def test_segmented_graph_replay_matches_eager(gpu, name):
    """Forward graph + one graph per backward segment (shared memory pool) replayed in order == the eager iteration."""
    l0, p0, _, _ = _run_model(gpu, name, segmented=False, use_graph=False, steps=4)
    l1, p1, n1, _ = _run_model(gpu, name, segmented=True, use_graph=True, steps=4)
    assert n1 >= 3
    assert l0 == l1, (l0, l1)
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)


@contextlib.contextmanager
def _poisoned_allocations():
    """Every torch.empty / empty_like / new_empty comes back filled with NaN (floating point) or 0xA5 (uint8 masks): a kernel
    that reads memory nobody wrote -- or relies on a fill that does not happen, as the memset node of round 4 did under graph
    replay (profiles/r4/r4_v13_graph_memset.md) -- shows up as a different result, in eager launches and in captured graphs alike
    (the fills are launches like any other and are captured with the step)."""
    orig = (torch.empty, torch.empty_like, torch.Tensor.new_empty)

    def poison(t):
        if t.numel():
            if t.is_floating_point():
                t.fill_(float("nan"))
            elif t.dtype == torch.uint8:
                t.fill_(0xA5)
        return t
    torch.empty = lambda *a, **k: poison(orig[0](*a, **k))
    torch.empty_like = lambda *a, **k: poison(orig[1](*a, **k))
    torch.Tensor.new_empty = lambda self, *a, **k: poison(orig[2](self, *a, **k))
    try:
        yield
    finally:
        torch.empty, torch.empty_like, torch.Tensor.new_empty = orig


@pytest.mark.parametrize("name", [pytest.param("slowfast_tiny", marks=pytest.mark.slow), "mvit_tiny", "x3d_tiny"])
def test_poisoned_allocations_do_not_change_the_step(sim, name):
    """Host simulator, eager: the training step reads nothing it (or a kernel before it) has not written."""
    l0, p0, _, _ = _run_model(sim, name, segmented=False, use_graph=False)
    with _poisoned_allocations():
        l1, p1, _, _ = _run_model(sim, name, segmented=False, use_graph=False)
    assert l0 == l1, (l0, l1)
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["slowfast_tiny", "mvit_tiny", "x3d_tiny"])
def test_graph_replay_with_poisoned_allocations(gpu, name):
    """Captured step (forward graph + backward segment graphs), four iterations, every allocation poisoned before use == the
    clean eager iterations bit for bit: nothing inside the graph depends on what a buffer held before the replay."""
    l0, p0, _, _ = _run_model(gpu, name, segmented=False, use_graph=False, steps=4)
    with _poisoned_allocations():
        l1, p1, _, _ = _run_model(gpu, name, segmented=True, use_graph=True, steps=4)
        l2, p2, _, _ = _run_model(gpu, name, segmented=False, use_graph=True, steps=4)
    assert l0 == l1 == l2, (l0, l1, l2)
    for a, b, c in zip(p0, p1, p2):
        assert torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.gpu
def test_pathway_streams_do_not_change_the_step(gpu, monkeypatch):
    """engine.run_pathways (Slow / Fast pathway of every stage on two HIP streams, graph branches under capture): four training
    steps give the same losses and the same parameters BIT FOR BIT as the single-stream run -- eagerly, as one captured graph and
    as segmented backward graphs, repeatedly (a race between the streams -- shared scratch memory, a join that is missing --
    shows as a run that differs)."""
    from slowfast_amd import engine
    monkeypatch.setattr(engine, "PATHWAY_STREAMS", False)
    l0, p0, _, _ = _run_model(gpu, "slowfast_tiny", segmented=False, use_graph=False, steps=4)
    monkeypatch.setattr(engine, "PATHWAY_STREAMS", True)
    for rep in range(3):
        for segmented, use_graph in ((False, False), (True, True), (False, True)):
            l1, p1, _, _ = _run_model(gpu, "slowfast_tiny", segmented=segmented, use_graph=use_graph, steps=4)
            assert l1 == l0, (rep, segmented, use_graph, l0, l1)
            for a, b in zip(p0, p1):
                assert torch.equal(a, b), (rep, segmented, use_graph)
    assert engine._pathway_streams, "the Fast pathway must have run on its own stream"


@pytest.mark.gpu
def test_branch_streams_do_not_change_the_step(gpu, monkeypatch):
    """engine.run_branches (the q and the k / v pooling chains of MultiScaleAttention on two HIP streams, forward and backward):
    four MViT training steps bit for bit what the single-stream run gives -- eager, one captured graph, segmented graphs."""
    from slowfast_amd import engine
    monkeypatch.setattr(engine, "BRANCH_STREAMS", False)
    l0, p0, _, _ = _run_model(gpu, "mvit_tiny", segmented=False, use_graph=False, steps=4)
    monkeypatch.setattr(engine, "BRANCH_STREAMS", True)
    for rep in range(3):
        for segmented, use_graph in ((False, False), (True, True), (False, True)):
            l1, p1, _, _ = _run_model(gpu, "mvit_tiny", segmented=segmented, use_graph=use_graph, steps=4)
            assert l1 == l0, (rep, segmented, use_graph, l0, l1)
            for a, b in zip(p0, p1):
                assert torch.equal(a, b), (rep, segmented, use_graph)


def _run_plain_loop(device, name, steps=3):
    """``loss.backward(); optimizer.step()`` with torch.optim.SGD and NO TrainStep / GradReducer: the caller never joins a side
    stream itself."""
    from tests import model_checks as mc
    gold = mc.load_golden(name)
    cfg = mc.cfg_for(gold)
    model, sd, inputs, labels, *_ = mc.oracle_run(gold, cfg)
    model.load_state_dict(sd)
    model = model.to(device).train()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    xs, ys = [x.to(device) for x in inputs], labels.to(device)
    losses = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=False)
        loss = F.cross_entropy(model(xs).float(), ys)
        (loss * 64.0).backward()
        opt.step()                  # reads every param.grad on the current stream right behind backward()
        losses.append(float(loss))
    return losses, [p.detach().float().cpu().clone() for p in model.parameters()]


@pytest.mark.gpu
@pytest.mark.parametrize("name,flag", [("slowfast_tiny", "PATHWAY_STREAMS"), ("mvit_tiny", "BRANCH_STREAMS")])
def test_plain_backward_joins_side_streams(gpu, monkeypatch, name, flag):
    """a user loop without TrainStep.  The Fast pathway's / the k-v branch's backward nodes write param.grad
    on side streams; engine._notify queues join_side_streams as an end-of-backward callback, so optimizer.step() behind a plain
    backward() reads finished gradients: three steps bit for bit what the single-stream run gives, repeatedly."""
    from slowfast_amd import engine
    monkeypatch.setattr(engine, flag, False)
    l0, p0 = _run_plain_loop(gpu, name)
    monkeypatch.setattr(engine, flag, True)
    for rep in range(3):
        l1, p1 = _run_plain_loop(gpu, name)
        assert l1 == l0, (rep, l0, l1)
        for a, b in zip(p0, p1):
            assert torch.equal(a, b), rep
    assert engine._pathway_streams and not engine._join_queued


def test_static_clone_keeps_wpair_tag_and_strides():
    """TrainStep._static_clone: the captured graph's copy of a packed clip keeps the W-pair tag and the channels-last strides
    (without the tag StemConvUnit would try to convert an 8-channel tensor and fail at capture)."""
    from slowfast_amd.step import TrainStep
    base = torch.zeros((2, 4, 6, 3, 8), dtype=torch.float16)
    x = base.permute(0, 4, 1, 2, 3)
    x._sf_wpairs = True
    c = TrainStep._static_clone(x)
    assert getattr(c, "_sf_wpairs", False) and c.stride() == x.stride() and c.data_ptr() != x.data_ptr()
    assert c.permute(0, 2, 3, 4, 1).is_contiguous()
    assert not hasattr(TrainStep._static_clone(torch.zeros(3)), "_sf_wpairs")


@pytest.mark.gpu
def test_packed_loader_writes_static_inputs(gpu):
    """A TrainStep captured on clips packed by pack_pathways_u8; the next batches are packed STRAIGHT INTO the graph's static
    input buffers (no device-to-device copy) and replayed: same losses and parameters as the eager step fed the same clips."""
    import slowfast_amd as sa
    from slowfast_amd.data_parallel import GradReducer
    from slowfast_amd.optim import construct_optimizer
    from slowfast_amd.step import TrainStep
    from tests import model_checks as mc
    gold = mc.load_golden("slowfast_tiny")
    cfg = mc.cfg_for(gold)
    S, T = cfg.DATA.TRAIN_CROP_SIZE, cfg.DATA.NUM_FRAMES
    g = torch.Generator().manual_seed(11)
    batches = [(torch.randint(0, 256, (2, T, S, S, 3), generator=g, dtype=torch.int64).to(torch.uint8).to(gpu),
                torch.randint(0, cfg.MODEL.NUM_CLASSES, (2,), generator=g).to(gpu)) for _ in range(4)]

    def run(use_graph):
        torch.manual_seed(0)
        model = sa.MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg).to(gpu).train()
        red = GradReducer(model, bucket_mb=0.05)
        red.attach_torch_param_hooks(model.head.parameters())
        opt = construct_optimizer(model, cfg, red, loss_scale=64.0, dynamic_loss_scale=False)
        step = TrainStep(model, red, opt, F.cross_entropy, use_graph=use_graph, warmup=1)
        losses = []
        for frames, y in batches:
            st = step.static_inputs()
            if st is None:
                losses.append(float(step(sa.pack_pathways_u8(frames, cfg), y)))
            else:                                           # the loader writes into the graph's own buffers
                xs = sa.pack_pathways_u8(frames, cfg, out=st[0])
                assert all(a.data_ptr() == b.data_ptr() for a, b in zip(xs, st[0]))
                st[1].copy_(y)
                losses.append(float(step(xs, st[1])))
        out = losses, [p.detach().float().cpu().clone() for p in model.parameters()]
        red.close()
        return out

    l0, p0 = run(False)
    l1, p1 = run(True)
    assert l0 == l1, (l0, l1)
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)


def _count_calls(fn):
    """Runs fn() with a libsfamd call observer; returns (result, {entry point: calls})."""
    from slowfast_amd import lib
    counts = {}

    def observer(name, thunk, work):
        counts[name] = counts.get(name, 0) + 1
        return thunk()
    lib.set_call_observer(observer)
    try:
        return fn(), counts
    finally:
        lib.set_call_observer(None)


def test_weight_pack_plan_one_launch_same_result(sim, monkeypatch):
    """engine.WeightPackPlan: after the recorded first iteration every conv weight of the step is packed by ONE
    sf_prep_weights_batch launch -- same operands bit for bit, so same parameters after training as with per-layer packing."""
    from slowfast_amd import engine
    (params, ctl, losses), counts = _count_calls(lambda: _run_flat(sim, use_graph=False, steps=3))
    nconv = 7                                            # 2 bottleneck blocks: a, b, c (+ projection shortcut of the first)
    assert counts["sf_prep_weights"] == nconv            # first (recorded) iteration only
    assert counts["sf_prep_weights_batch"] == 2
    monkeypatch.setattr(engine, "PACK_PLAN", False)
    (params0, ctl0, losses0), counts0 = _count_calls(lambda: _run_flat(sim, use_graph=False, steps=3))
    assert counts0["sf_prep_weights"] == 3 * nconv and "sf_prep_weights_batch" not in counts0
    assert losses == losses0 and torch.equal(ctl, ctl0)
    for a, b in zip(params, params0):
        assert torch.equal(a, b)


def test_weight_pack_plan_covers_mvit_linears(sim, monkeypatch):
    from slowfast_amd import engine
    (la, pa, _, _), counts = _count_calls(lambda: _run_model(sim, "mvit_tiny", segmented=False, use_graph=False, steps=2))
    assert counts.get("sf_prep_weights_batch", 0) == 1
    monkeypatch.setattr(engine, "PACK_PLAN", False)
    (lb, pb, _, _), _ = _count_calls(lambda: _run_model(sim, "mvit_tiny", segmented=False, use_graph=False, steps=2))
    assert la == lb
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)


# ---- the step glue against the reference's own sequence: torch.optim + torch.amp.GradScaler (rows a20 / f1) ------------------
_SGD_OPTS = ["SOLVER.OPTIMIZING_METHOD", "sgd", "SOLVER.MOMENTUM", 0.9, "SOLVER.NESTEROV", True, "SOLVER.WEIGHT_DECAY", 1e-4,
             "BN.WEIGHT_DECAY", 0.0]
_ADAMW_OPTS = ["SOLVER.OPTIMIZING_METHOD", "adamw", "SOLVER.WEIGHT_DECAY", 0.05, "SOLVER.ZERO_WD_1D_PARAM", True,
               "SOLVER.CLIP_GRAD_L2NORM", 1.0]


def test_train_step_arithmetic_vs_torch_optim_sgd_hostsim(sim):
    """Comparison (A) of tests/step_checks.py on the host simulator: FlatOptimizer's SGD-Nesterov / GradScaler arithmetic, skip
    decisions and scale trajectory == torch.optim.SGD + torch.amp.GradScaler fed the engine's own gradients."""
    from tests import step_checks
    step_checks.check_train_step_vs_torch(sim, "c2d_tiny", _SGD_OPTS, steps=4, overflow_at=1, lr=1e-6, compare_oracle=False)


def test_train_step_vs_torch_optim_adamw_hostsim(sim):
    """Host-simulator twin of the GPU test below (comparisons (A) and (B), eager)."""
    from tests import step_checks
    step_checks.check_train_step_vs_torch(sim, "mvit_tiny", _ADAMW_OPTS, steps=4, overflow_at=2, lr=2e-4)


@pytest.mark.slow
def test_train_step_vs_torch_optim_sgd_hostsim(sim):
    """Host-simulator twin of test_train_step_vs_torch_optim_sgd (minutes on the CPU: SF_RUN_SLOW=1)."""
    from tests import step_checks
    step_checks.check_train_step_vs_torch(sim, "c2d_wc", _SGD_OPTS, steps=3, overflow_at=1, lr=0.02)


@pytest.mark.gpu
def test_train_step_vs_torch_optim_sgd(gpu):
    """TrainStep (HIP graphs) + FlatOptimizer (sf_flat_sumsq / sf_step_control / sf_flat_sgd) on c2d_wc: SGD-Nesterov with the
    BN / non-BN weight-decay groups, dynamic loss scale, one injected overflow -- against torch.optim.SGD +
    torch.amp.GradScaler fed (A) the engine's gradients (fp32 round-off) and (B) the fp32 oracle's (1e-3 on the parameters
    after 5 steps; identical skip decisions and scale trajectory).  tools/train_net.py:150-172, optimizer.py:100-140."""
    from tests import step_checks
    rep = {}
    try:
        step_checks.check_train_step_vs_torch(gpu, "c2d_wc", _SGD_OPTS, steps=5, overflow_at=2, lr=0.02, report=rep)
    finally:
        print("train_step sgd", rep)


@pytest.mark.gpu
def test_train_step_vs_torch_optim_adamw(gpu):
    """The same with AdamW (decoupled decay, zero-decay group from no_weight_decay() + 1-D parameters) and
    SOLVER.CLIP_GRAD_L2NORM 1.0 on mvit_tiny (gradient norm ~35: every step is clipped)."""
    from tests import step_checks
    rep = {}
    try:
        step_checks.check_train_step_vs_torch(gpu, "mvit_tiny", _ADAMW_OPTS, steps=5, overflow_at=3, lr=2e-4, report=rep)
    finally:
        print("train_step adamw", rep)
