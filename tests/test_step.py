"""TrainStep (slowfast_amd/step.py): the captured-graph iteration must be the same computation as the eager
sequence zero_grad -> forward -> loss -> backward -> finish -> optimizer.step (tools/train_net.py:104-172)."""
import copy

import pytest
import torch
import torch.nn.functional as F

from tests.kernel_checks import host_to_cl
from tests.test_data_parallel import _build


def _run(device, use_graph, steps, seed=0):
    from slowfast_amd.data_parallel import GradReducer
    from slowfast_amd.step import TrainStep
    torch.manual_seed(seed)
    net = _build().to(device).train()
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
    red = GradReducer(net)
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
