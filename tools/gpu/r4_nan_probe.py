"""Diagnostic (round 4): which gradients go non-finite in the mvit_tiny AdamW train-step test, per iteration, under the
switches given in the environment.  usage: python tools/gpu/r4_nan_probe.py [graph|eager] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F

from slowfast_amd.data_parallel import GradReducer
from slowfast_amd.optim import CTL_SCALE, construct_optimizer
from slowfast_amd.step import TrainStep
from tests import model_checks as mc

mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
case = sys.argv[3] if len(sys.argv) > 3 else "mvit_tiny"
dev = torch.device("cuda", 0)
if os.environ.get("PROBE_DIRTY", "1") == "1":          # leave NaN bit patterns in the allocator's free blocks
    junk = [torch.full((64 << 20,), float("nan"), device=dev) for _ in range(4)]
    del junk
opts = ["SOLVER.OPTIMIZING_METHOD", "adamw", "SOLVER.WEIGHT_DECAY", 0.05, "SOLVER.ZERO_WD_1D_PARAM", True, "SOLVER.CLIP_GRAD_L2NORM", 1.0]
gold = mc.load_golden(case)
cfg = mc.cfg_for(gold, extra=["TRAIN.MIXED_PRECISION", True] + opts)
model, sd, inputs, labels, *_ = mc.oracle_run(gold, cfg)
model.load_state_dict(sd)
model = model.to(dev).train()
red = GradReducer(model)
red.attach_torch_param_hooks(model.head.parameters())
opt = construct_optimizer(model, cfg, red, loss_scale=256.0, dynamic_loss_scale=True)
opt.growth_interval = 2
for g in opt.param_groups:
    g["lr"] = 2e-4
name_of = {id(p): k for k, p in model.named_parameters()}
step = TrainStep(model, red, opt, F.cross_entropy, use_graph=(mode == "graph"), warmup=1)
xs, ys = [x.to(dev) for x in inputs], labels.to(dev)
for it in range(steps):
    scale = float(opt.ctl[CTL_SCALE])
    loss = float(step(xs, ys))
    flat = red.flat.detach().float().cpu()
    bad, off = [], 0
    for p in red.params:
        n = p.numel()
        seg = flat[off:off + n]
        if not torch.isfinite(seg).all():
            bad.append((name_of[id(p)], int((~torch.isfinite(seg)).sum()), n))
        off += n
    print(f"{mode} it {it} scale {scale} loss {loss:.5f} grad_norm {float(opt.grad_norm):.4f} nonfinite params {len(bad)}: {bad[:6]}", flush=True)
red.close()
