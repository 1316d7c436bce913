"""Optimizer + loss scaling on the flat gradient memory.

Replaces, for the GradReducer / TrainStep path, what tools/train_net.py:150-172 runs between ``loss.backward()`` and the
next iteration -- ``scaler.unscale_``, ``clip_grad_norm_`` / ``clip_grad_value_``, ``optimizer.get_grad_norm_``,
``scaler.step(optimizer)``, ``scaler.update()`` and ``misc.check_nan_losses`` -- and ``optimizer.step()`` itself
(slowfast/models/optimizer.py:100-140: SGD with momentum / dampening / Nesterov, AdamW).  Three libsfamd launches per
iteration (csrc/sf_optim.h), no host synchronisation: the dynamic loss scale, the overflow flag and the gradient norm live in
an 8-word control block in device memory.

Parameters are re-pointed to views of ONE flat fp32 buffer laid out like GradReducer.flat (gradients) and the optimizer
state, so the update is a single pass over contiguous memory.  ``param_groups`` mirrors torch.optim (lists of dicts with
"lr" / "weight_decay"), so the reference's ``optim.set_lr(optimizer, lr)`` (slowfast/models/optimizer.py:143-152) keeps
working on it.
"""
from ctypes import c_float

import numpy as np
import torch

from . import engine
from .lib import get_lib

_SEG_DTYPE = np.dtype([("start", "<i8"), ("end", "<i8"), ("group", "<i4"), ("pad", "<i4")])
_BLOCK = 1024       # elements per workgroup of the update kernels (SF_OPT_BLOCK_ELEMS)
CTL_SCALE, CTL_TRACKER, CTL_FOUND_INF, CTL_GRAD_NORM, CTL_MULT, CTL_STEPS, CTL_SKIPPED = range(7)


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else None


class FlatOptimizer:
    def __init__(self, param_groups, reducer, method="sgd", momentum=0.0, dampening=0.0, nesterov=False, betas=(0.9, 0.999),
                 eps=1e-8, loss_scale=1.0, dynamic_loss_scale=False, growth_factor=2.0, backoff_factor=0.5,
                 growth_interval=2000, clip_grad_l2norm=None, clip_grad_val=None):
        assert method in ("sgd", "adamw")
        groups = [dict(g) for g in param_groups]
        assert 1 <= len(groups) <= 8, "1..8 parameter groups"
        self.param_groups = groups
        self.method, self.momentum, self.dampening, self.nesterov = method, float(momentum), float(dampening), bool(nesterov)
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        self.dynamic = bool(dynamic_loss_scale)
        self.growth, self.backoff, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        # tools/train_net.py:154-163: CLIP_GRAD_VAL takes precedence, the L2-norm clip only runs when no value clip is set
        self.clip_val = float(clip_grad_val) if clip_grad_val else 0.0
        self.clip_norm = float(clip_grad_l2norm) if (clip_grad_l2norm and not self.clip_val) else 0.0
        self.reducer = reducer
        flat_g = reducer.flat
        dev = flat_g.device
        group_of = {}
        for gi, g in enumerate(groups):
            for p in g["params"]:
                group_of[p] = gi
        # parameters become views of one flat buffer with the gradient buffer's layout
        self.flat_param = torch.empty_like(flat_g)
        segs, blk_seg, blk_off = [], [], []
        off = 0
        for p in reducer.params:
            n = p.numel()
            assert p in group_of, "every parameter of the reducer must belong to a parameter group"
            view = self.flat_param[off:off + n].view_as(p)
            view.copy_(p.data)
            p.data = view
            si = len(segs)
            segs.append((off, off + n, group_of[p], 0))
            for b in range(0, n, _BLOCK):
                blk_seg.append(si)
                blk_off.append(b)
            off += n
        assert off == flat_g.numel()
        self.m1 = torch.zeros_like(flat_g) if (method == "adamw" or self.momentum != 0.0) else None
        self.m2 = torch.zeros_like(flat_g) if method == "adamw" else None
        self.segs = torch.from_numpy(np.array(segs, dtype=_SEG_DTYPE).view(np.uint8).copy()).to(dev)
        self.blk_seg = torch.tensor(blk_seg, dtype=torch.int32, device=dev)
        self.blk_off = torch.tensor(blk_off, dtype=torch.int32, device=dev)
        self.nblocks = len(blk_seg)
        self.ctl = torch.zeros(8, dtype=torch.float32, device=dev)
        self.ctl[CTL_SCALE] = float(loss_scale)
        self.init_loss_scale = float(loss_scale)
        # partial rows of the norm / overflow pass: one range per gradient bucket (finish_and_step), rows simply add up
        lib = get_lib()
        self._bucket_rows = []
        r = 0
        for s0, e0, _ in reducer.buckets:
            nb = lib.call("sf_flat_blocks", e0 - s0)
            self._bucket_rows.append((r, nb))
            r += nb
        self._part = torch.empty((max(r, lib.call("sf_flat_blocks", flat_g.numel())), 2), dtype=torch.float32, device=dev)
        self._rows_bucketed = r

    # -- what the training loop touches ---------------------------------------------------------------------------
    @property
    def loss_scale(self):
        """0-d device tensor: multiply the loss by it (``(loss * opt.loss_scale).backward()``); under a captured graph the
        replay reads the current value from device memory, so a scale change needs no re-capture."""
        return self.ctl[CTL_SCALE]

    @property
    def grad_norm(self):
        """Global L2 norm of the unscaled, rank-averaged gradients of the last step (device tensor; inf on overflow)."""
        return self.ctl[CTL_GRAD_NORM]

    @property
    def found_inf(self):
        return self.ctl[CTL_FOUND_INF]

    def _sumsq_bucket(self, bi):
        """Bucket bi's share of the norm / overflow pass (its own rows of the partial table)."""
        s0, e0, _ = self.reducer.buckets[bi]
        r0, _ = self._bucket_rows[bi]
        g = self.reducer.flat
        get_lib().call("sf_flat_sumsq", g.data_ptr() + 4 * s0, e0 - s0, self._part.data_ptr() + 8 * r0, _stream(g),
                       work=dict(bytes=4.0 * (e0 - s0)))

    def finish_and_step(self):
        """GradReducer.finish(loss_scale=None) + step(), with the norm / overflow pass taken bucket by bucket as the collectives
        complete: the pass over the early buckets runs under the exchange of the late ones, and the update waits only for the
        control launch (only the ~30 us pass moves -- the update itself needs the global norm)."""
        self.reducer.finish(loss_scale=None, on_bucket=self._sumsq_bucket)
        self.step(_sumsq_rows=self._rows_bucketed)

    def step(self, _sumsq_rows=None):
        """Call after the gradient all-reduce finished (GradReducer.finish(loss_scale=None)): norm + overflow check,
        GradScaler update, clipped / unscaled parameter update -- skipped as a whole on overflow.  ``_sumsq_rows``: the partial
        table already holds that many rows (finish_and_step)."""
        lib, g = get_lib(), self.reducer.flat
        s = _stream(g)
        nrows = _sumsq_rows
        if nrows is None:
            nrows = lib.call("sf_flat_blocks", g.numel())
            lib.call("sf_flat_sumsq", g.data_ptr(), g.numel(), self._part.data_ptr(), s, work=dict(bytes=4.0 * g.numel()))
        lib.call("sf_step_control", self._part.data_ptr(), nrows, self.ctl.data_ptr(), float(self.reducer.world),
                 self.clip_norm, int(self.dynamic), self.growth, self.backoff, self.growth_interval, s)
        ng = len(self.param_groups)
        lr = (c_float * ng)(*[float(gp["lr"]) for gp in self.param_groups])
        wd = (c_float * ng)(*[float(gp.get("weight_decay", 0.0)) for gp in self.param_groups])
        if self.method == "sgd":
            lib.call("sf_flat_sgd", self.flat_param.data_ptr(), g.data_ptr(), self.m1.data_ptr() if self.m1 is not None else None,
                     self.segs.data_ptr(), self.blk_seg.data_ptr(), self.blk_off.data_ptr(), self.nblocks, self.ctl.data_ptr(),
                     lr, wd, ng, self.clip_val, self.momentum, self.dampening, int(self.nesterov), s,
                     work=dict(bytes=4.0 * g.numel() * (3 + 2 * int(self.m1 is not None))))
        else:
            lib.call("sf_flat_adamw", self.flat_param.data_ptr(), g.data_ptr(), self.m1.data_ptr(), self.m2.data_ptr(),
                     self.segs.data_ptr(), self.blk_seg.data_ptr(), self.blk_off.data_ptr(), self.nblocks, self.ctl.data_ptr(),
                     lr, wd, ng, self.clip_val, self.betas[0], self.betas[1], self.eps, s, work=dict(bytes=4.0 * g.numel() * 7))
        engine.PARAM_EPOCH += 1         # the kernels wrote the parameters without bumping tensor._version: packed-weight
                                        # caches of the eager path must not be reused (a captured graph re-packs anyway)

    def zero_grad(self, set_to_none=False):
        self.reducer.zero_grad()

    # -- checkpoint surface (utils/checkpoint.py saves optimizer.state_dict()) ---------------------------------------------
    def state_dict(self):
        return {"ctl": self.ctl.detach().cpu(), "m1": None if self.m1 is None else self.m1.detach().cpu(),
                "m2": None if self.m2 is None else self.m2.detach().cpu(),
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.ctl.copy_(sd["ctl"])
        if self.m1 is not None and sd.get("m1") is not None:
            self.m1.copy_(sd["m1"])
        if self.m2 is not None and sd.get("m2") is not None:
            self.m2.copy_(sd["m2"])
        for g, s in zip(self.param_groups, sd.get("param_groups", [])):
            g.update(s)


def construct_optimizer(model, cfg, reducer, loss_scale=1.0, dynamic_loss_scale=None):
    """FlatOptimizer configured as slowfast/models/optimizer.py:9-140 configures torch's (the ``LAYER_DECAY == 1.0`` branch,
    :26-92): BatchNorm parameters get BN.WEIGHT_DECAY; parameters whose dotted name contains an entry of
    ``model.no_weight_decay()`` (MViT: pos_embed* / rel_pos_* / cls_token under MVIT.ZERO_DECAY_POS_CLS) get no decay; so do
    1-D parameters and every ``*.bias`` when SOLVER.ZERO_WD_1D_PARAM; the rest SOLVER.WEIGHT_DECAY.
    SOLVER.OPTIMIZING_METHOD "sgd" (momentum / dampening / nesterov) or "adamw" / "mt_adamw" (SOLVER.BETAS, eps 1e-8);
    clipping from SOLVER.CLIP_GRAD_L2NORM / CLIP_GRAD_VAL; the dynamic loss scale defaults to TRAIN.MIXED_PRECISION
    (GradScaler's constants).  Options the fused kernels do not implement raise instead of silently changing the update rule:
    SOLVER.LAYER_DECAY != 1 (per-layer lr scale, optimizer.py:155-220), SOLVER.LARS_ON, "adam" (L2-coupled decay)."""
    layer_decay = float(cfg.SOLVER.get("LAYER_DECAY", 1.0))
    if not 0.0 < layer_decay <= 1.0:
        raise ValueError("Layer decay should be in (0, 1], but is {}".format(layer_decay))
    if layer_decay != 1.0:
        raise NotImplementedError("SOLVER.LAYER_DECAY < 1 (per-layer learning-rate scale) is not built into FlatOptimizer")
    if cfg.SOLVER.get("LARS_ON", False):
        raise NotImplementedError("SOLVER.LARS_ON is not built into FlatOptimizer")
    inner = model.module if hasattr(model, "module") and isinstance(model.module, torch.nn.Module) else model
    skip = inner.no_weight_decay() if hasattr(inner, "no_weight_decay") else ()
    bn, rest, zero = [], [], []
    seen = set()
    for name_m, m in inner.named_modules():
        is_bn = isinstance(m, torch.nn.modules.batchnorm._NormBase)
        for name_p, p in m.named_parameters(recurse=False):
            name = "{}.{}".format(name_m, name_p).strip(".")
            if not p.requires_grad or id(p) in seen:
                continue
            seen.add(id(p))
            if is_bn:
                bn.append(p)
            elif any(k in name for k in skip):
                zero.append(p)
            elif cfg.SOLVER.ZERO_WD_1D_PARAM and (p.dim() == 1 or name.endswith(".bias")):
                zero.append(p)
            else:
                rest.append(p)
    lr = cfg.SOLVER.BASE_LR
    groups = [g for g in ({"params": bn, "weight_decay": cfg.BN.WEIGHT_DECAY, "lr": lr},
                          {"params": rest, "weight_decay": cfg.SOLVER.WEIGHT_DECAY, "lr": lr},
                          {"params": zero, "weight_decay": 0.0, "lr": lr}) if g["params"]]
    dyn = bool(cfg.TRAIN.MIXED_PRECISION) if dynamic_loss_scale is None else dynamic_loss_scale
    kw = dict(loss_scale=loss_scale, dynamic_loss_scale=dyn, clip_grad_l2norm=cfg.SOLVER.CLIP_GRAD_L2NORM,
              clip_grad_val=cfg.SOLVER.CLIP_GRAD_VAL)
    method = cfg.SOLVER.OPTIMIZING_METHOD
    if method in ("adamw", "mt_adamw"):
        betas = tuple(cfg.SOLVER.get("BETAS", (0.9, 0.999)))
        return FlatOptimizer(groups, reducer, method="adamw", betas=betas, eps=1e-8, **kw)
    if method == "sgd":
        return FlatOptimizer(groups, reducer, method="sgd", momentum=cfg.SOLVER.MOMENTUM, dampening=cfg.SOLVER.DAMPENING,
                             nesterov=cfg.SOLVER.NESTEROV, **kw)
    raise NotImplementedError("Does not support {} optimizer".format(method))
