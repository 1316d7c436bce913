"""Norm-layer factory and the two non-default BatchNorm variants of the reference, as parameter
containers for the engine -- the arithmetic runs in the same libsfamd kernels as plain BatchNorm3d (partial sums in the
producing convolution's epilogue, sf_bn_finalize, normalisation applied in the consumer's operand load).

``get_norm(cfg)`` mirrors slowfast/models/batchnorm_helper.py:16-37.

``SubBatchNorm3d`` (batchnorm_helper.py:40-112; multigrid training): statistics per 1/num_splits of the batch
(sample n belongs to split n % num_splits -- the reference's ``x.view(n // S, c * S, t, h, w)``), running statistics
of the splits side by side in ``split_bn``, one shared affine pair, ``aggregate_stats()`` before eval.  BatchNorm is the
only coupling between the samples of a batch, so a training step of a network whose norms are ALL SubBatchNorm3d(S)
equals S independent passes over the sub-batches x[j::S], pass j using the j-th block of ``split_bn``'s running
statistics, with the parameter gradients summed.  That is how the engine executes it (``run_in_splits``): no kernel
needs to know about splits, and each pass keeps the fused single-statistics schedule.

``NaiveSyncBatchNorm3d`` (pytorchvideo.layers.batch_norm, imported by batchnorm_helper.py:10-13; pytorchvideo is not
vendored under /root/reference, so this restates its published algorithm -- "parity unpinned" for the cross-rank part):
statistics averaged over the ranks of a sync group with equal weights (equal per-rank batch sizes), running_var updated
with the BIASED batch variance (unlike nn.BatchNorm3d), gradient of the input through the all-reduced sums, affine
gradients local.  The engine all-reduces the [2, C] partial sums between the statistics epilogue and sf_bn_finalize
(forward) and between sf_bn_bwd_reduce and sf_bn_bwd_finalize (backward): two tiny collectives per layer per pass, as
in the reference.  With one rank per group, or in eval mode, it is plain BatchNorm3d.
"""
from functools import partial

import torch
import torch.distributed as dist
import torch.nn as nn


class SubBatchNorm3d(nn.Module):
    """Same constructor, children (bn, split_bn) and state_dict keys as the reference class."""

    def __init__(self, num_splits, **args):
        super().__init__()
        self.num_splits = num_splits
        num_features = args["num_features"]
        if args.get("affine", True):
            self.affine = True
            args["affine"] = False
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            raise NotImplementedError("SubBatchNorm3d(affine=False) is not constructed anywhere in the reference models")
        self.bn = nn.BatchNorm3d(**args)
        args["num_features"] = num_features * num_splits
        self.split_bn = nn.BatchNorm3d(**args)
        self.num_features = num_features
        self.bn.__dict__["_sf_no_bump"] = True         # only split_bn counts batches in training, as in the reference
        self.__dict__["_active_split"] = None          # set by run_in_splits() around each sub-batch pass

    # ---- what the engine reads from a norm container -----------------------------------------------------------
    eps = property(lambda self: self.bn.eps)
    momentum = property(lambda self: self.bn.momentum)
    track_running_stats = property(lambda self: self.bn.track_running_stats)
    num_batches_tracked = property(lambda self: self.split_bn.num_batches_tracked)

    def _stat(self, name):
        if self.training:
            j = self.__dict__["_active_split"]
            if j is None:
                if self.num_splits != 1:
                    raise RuntimeError("SubBatchNorm3d in training mode outside run_in_splits(): the model-level forward "
                                       "of slowfast_amd drives the sub-batch passes")
                j = 0
            C = self.num_features
            return getattr(self.split_bn, name)[j * C:(j + 1) * C]
        return getattr(self.bn, name)

    running_mean = property(lambda self: self._stat("running_mean"))
    running_var = property(lambda self: self._stat("running_var"))

    # ---- reference API -------------------------------------------------------------------------------------------
    def _get_aggregated_mean_std(self, means, stds, n):
        mean = means.view(n, -1).sum(0) / n
        std = stds.view(n, -1).sum(0) / n + ((means.view(n, -1) - mean) ** 2).view(n, -1).sum(0) / n
        return mean.detach(), std.detach()

    def aggregate_stats(self):
        """Synchronise bn.running_* from the per-split statistics; call before eval (batchnorm_helper.py:85-97)."""
        if self.split_bn.track_running_stats:
            self.bn.running_mean.data, self.bn.running_var.data = self._get_aggregated_mean_std(
                self.split_bn.running_mean, self.split_bn.running_var, self.num_splits)

    def forward(self, x):
        raise RuntimeError("norm layers of slowfast_amd are parameter containers: the fused block schedules apply them")


class NaiveSyncBatchNorm3d(nn.BatchNorm3d):
    """pytorchvideo's NaiveSyncBatchNorm3d constructor (num_sync_devices, global_sync, **BatchNorm3d args); state_dict of
    nn.BatchNorm3d.  ``sync_group()`` returns (process group, size) when statistics must be exchanged."""

    def __init__(self, num_sync_devices=None, global_sync=False, **args):
        self.global_sync = global_sync
        if self.global_sync and num_sync_devices is not None:
            raise ValueError(f"Cannot set num_sync_devices separately when global_sync = {self.global_sync}")
        if not self.global_sync and num_sync_devices is None:
            raise ValueError(f"num_sync_devices cannot be None when global_sync = {self.global_sync}")
        self.num_sync_devices = num_sync_devices
        super().__init__(**args)

    def sync_group(self):
        if not self.training or not (dist.is_available() and dist.is_initialized()):
            return None
        world = dist.get_world_size()
        if world == 1:
            return None
        if self.global_sync or self.num_sync_devices >= world:
            return (None, world)
        if self.num_sync_devices <= 1:
            return None
        return _local_groups(self.num_sync_devices)

    def forward(self, x):
        raise RuntimeError("norm layers of slowfast_amd are parameter containers: the fused block schedules apply them")


_group_cache = {}


def _local_groups(n):
    """Consecutive blocks of ``n`` ranks; every rank creates every group (torch.distributed.new_group is collective)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    assert world % n == 0, f"BN.NUM_SYNC_DEVICES {n} does not divide the world size {world}"
    if n not in _group_cache:
        groups = [dist.new_group(list(range(i * n, (i + 1) * n))) for i in range(world // n)]
        _group_cache[n] = groups
    return (_group_cache[n][rank // n], n)


def get_norm(cfg):
    """BN.NORM_TYPE -> norm layer class (slowfast/models/batchnorm_helper.py:16-37)."""
    t = cfg.BN.NORM_TYPE
    if t in {"batchnorm", "sync_batchnorm_apex"}:
        return nn.BatchNorm3d
    if t == "sub_batchnorm":
        return partial(SubBatchNorm3d, num_splits=cfg.BN.NUM_SPLITS)
    if t == "sync_batchnorm":
        return partial(NaiveSyncBatchNorm3d, num_sync_devices=cfg.BN.NUM_SYNC_DEVICES, global_sync=cfg.BN.GLOBAL_SYNC)
    raise NotImplementedError(f"Norm type {t} is not supported")


def aggregate_sub_bn_stats(module):
    """slowfast/utils/misc.py:372-387 for the drop-in class: aggregate_stats() on every SubBatchNorm3d (call before
    evaluation / checkpointing, as train_net.py:710 does); returns how many layers were aggregated."""
    count = 0
    for m in module.modules():
        if isinstance(m, SubBatchNorm3d):
            m.aggregate_stats()
            count += 1
    return count


def num_splits_of(model, full_batch=()):
    """num_splits shared by the model's SubBatchNorm3d layers (1 when it has none); cached on the model.  ``full_batch``:
    child modules that run on the whole batch after the split passes (the X3D head keeps a plain BatchNorm3d in the
    reference, video_model_builder.py:788-797 / head_helper.py:374)."""
    s = model.__dict__.get("_sf_num_splits")
    if s is None:
        vals = {m.num_splits for m in model.modules() if isinstance(m, SubBatchNorm3d)}
        assert len(vals) <= 1, f"SubBatchNorm3d layers with different num_splits: {vals}"
        outside = {id(b) for top in full_batch for b in top.modules()}
        plain = any(isinstance(m, nn.modules.batchnorm._BatchNorm) and id(m) not in outside and not _inside_sub(model, m)
                    for m in model.modules())
        s = vals.pop() if vals else 1
        assert s == 1 or not plain, "sub-batch execution needs every norm layer of the model to be SubBatchNorm3d"
        model.__dict__["_sf_num_splits"] = s
    return s


def _inside_sub(model, bn):
    return any(isinstance(m, SubBatchNorm3d) and (bn is m.bn or bn is m.split_bn) for m in model.modules())


def run_in_splits(model, forward_once, inputs, S, params=None):
    """Training forward of a SubBatchNorm3d(S) model: S passes over the sub-batches x[j::S] (see the module docstring);
    returns the outputs re-interleaved to the original sample order.  Gradient-ready notifications of the engine are
    held back until all S passes have contributed (engine.hold_notifications)."""
    from . import engine
    subs = [m for m in model.modules() if isinstance(m, SubBatchNorm3d)]
    N = inputs[0].shape[0]
    assert N % S == 0, f"batch {N} is not divisible by BN.NUM_SPLITS {S}"
    engine.hold_notifications(S, model.parameters() if params is None else params)   # params: the split-pass parameters
    outs = []
    try:
        for j in range(S):
            for m in subs:
                m.__dict__["_active_split"] = j
            outs.append(forward_once([x[j::S] for x in inputs]))
    finally:
        for m in subs:
            m.__dict__["_active_split"] = None
    return interleave(outs)


def interleave(outs):
    """[S tensors (N/S, ...)] -> (N, ...) with sample i*S + j taken from outs[j][i] (the inverse of x[j::S])."""
    out = torch.stack(outs, 1)
    return out.reshape((out.shape[0] * out.shape[1],) + tuple(out.shape[2:]))
