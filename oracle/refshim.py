"""Import shim for the UNMODIFIED reference (TEST INFRASTRUCTURE; only used in the build container).

`install()` makes ``/root/reference/slowfast/{models,config}`` importable on a machine that lacks the
reference's un-vendored third-party packages (fvcore, pytorchvideo, detectron2, iopath, ...), by
registering small stand-ins for exactly the symbols its model code imports (SURVEY.md 8c lists them):
  fvcore.common.config.CfgNode     -> slowfast_amd.config.CfgNode (yacs-compatible subset)
  fvcore.common.registry.Registry  -> slowfast_amd.registry.Registry
  fvcore.nn.weight_init            -> c2_msra_fill = kaiming_normal_(fan_out, relu); c2_xavier_fill = kaiming_uniform_(a=1)
  pytorchvideo.layers.swish.Swish, pytorchvideo.layers.batch_norm.NaiveSyncBatchNorm{1d,3d} (import-time only)
  pytorchvideo.losses.soft_target_cross_entropy.SoftTargetCrossEntropyLoss
  detectron2.layers.ROIAlign (the oracle's restatement of the published algorithm), slowfast.utils.logging -> stdlib logging
``slowfast`` and ``slowfast.models`` are pre-seeded as path-only packages so that their __init__.py
(which drags in cv2 / torchvision through the SSL models) is bypassed.  Nothing here is shipped with
or imported by the product package; it does not exist on the GPU box and no `-m gpu` test needs it.
"""
import logging as _pylogging
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("SLOWFAST_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "slowfast", "models"))


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _package(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def install():
    if "slowfast.models.video_model_builder" in sys.modules:
        return
    if not available():
        raise RuntimeError(f"reference tree not found under {REFERENCE_ROOT}")
    import torch
    import torch.nn as nn

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from slowfast_amd.config import CfgNode
    from slowfast_amd.registry import Registry

    class _RefCfgNode(CfgNode):
        """yacs semantics the reference relies on: new keys only via merge of known keys."""

    def c2_msra_fill(module):
        nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)

    def c2_xavier_fill(module):
        nn.init.kaiming_uniform_(module.weight, a=1)
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)

    class Swish(nn.Module):
        def forward(self, x):
            return x * torch.sigmoid(x)

    class SoftTargetCrossEntropyLoss(nn.Module):
        def __init__(self, ignore_index=-100, reduction="mean", normalize_targets=True):
            super().__init__()
            self.reduction, self.normalize_targets = reduction, normalize_targets

        def forward(self, x, target):
            if self.normalize_targets:
                target = target / (target.sum(-1, keepdim=True) + 1e-6)
            loss = torch.sum(-target * torch.nn.functional.log_softmax(x, dim=-1), dim=-1)
            return loss.mean() if self.reduction == "mean" else loss

    class NaiveSyncBatchNorm3d(nn.BatchNorm3d):
        """Stand-in for pytorchvideo.layers.batch_norm.NaiveSyncBatchNorm3d (un-vendored): its constructor contract and
        the single-process behaviour (plain BatchNorm3d when the sync group has one rank or in eval mode).  The
        cross-rank branch is not restated here -- tests cover it through slowfast_amd.batchnorm against plain
        BatchNorm over the concatenated batch."""

        def __init__(self, num_sync_devices=None, global_sync=False, **args):
            if global_sync and num_sync_devices is not None:
                raise ValueError(f"Cannot set num_sync_devices separately when global_sync = {global_sync}")
            if not global_sync and num_sync_devices is None:
                raise ValueError(f"num_sync_devices cannot be None when global_sync = {global_sync}")
            self.global_sync, self.num_sync_devices = global_sync, num_sync_devices
            super().__init__(**args)

    class _ImportOnly(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError("stand-in for an un-vendored dependency (import-time only)")

    _module("fvcore")
    _module("fvcore.common")
    _module("fvcore.common.config", CfgNode=_RefCfgNode)
    _module("fvcore.common.registry", Registry=Registry)
    _module("fvcore.nn")
    _module("fvcore.nn.weight_init", c2_msra_fill=c2_msra_fill, c2_xavier_fill=c2_xavier_fill)
    _module("pytorchvideo")
    _module("pytorchvideo.layers")
    _module("pytorchvideo.layers.swish", Swish=Swish)
    _module("pytorchvideo.layers.batch_norm", NaiveSyncBatchNorm1d=_ImportOnly, NaiveSyncBatchNorm3d=NaiveSyncBatchNorm3d)
    _module("pytorchvideo.losses")
    _module("pytorchvideo.losses.soft_target_cross_entropy", SoftTargetCrossEntropyLoss=SoftTargetCrossEntropyLoss)
    class ROIAlign(nn.Module):
        """detectron2.layers.ROIAlign is not installed here: the unmodified reference head runs on the oracle's
        restatement of the published algorithm (oracle/video_ref.py:roi_align) -- everything around ROIAlign is pinned
        by the reference, ROIAlign itself is 'parity unpinned'."""

        def __init__(self, output_size, spatial_scale, sampling_ratio, aligned=True):
            super().__init__()
            self.output_size = tuple(output_size) if isinstance(output_size, (list, tuple)) else (output_size, output_size)
            self.spatial_scale, self.sampling_ratio, self.aligned = spatial_scale, sampling_ratio, aligned

        def forward(self, input, rois):
            from oracle import video_ref
            return video_ref.roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned)

    _module("detectron2")
    _module("detectron2.layers", ROIAlign=ROIAlign)

    ref = os.path.join(REFERENCE_ROOT, "slowfast")
    _package("slowfast", ref)
    _package("slowfast.models", os.path.join(ref, "models"))
    _package("slowfast.utils", os.path.join(ref, "utils"))
    _package("slowfast.config", os.path.join(ref, "config"))
    log = _module("slowfast.utils.logging", get_logger=_pylogging.getLogger)
    sys.modules["slowfast.utils"].logging = log

    import slowfast.models.video_model_builder  # noqa: F401  (registers SlowFast / ResNet / X3D / MViT)


def reference_cfg(yaml_rel, opts=()):
    """get_cfg() of the reference + one of its YAML files + KEY VALUE overrides."""
    install()
    from slowfast.config.defaults import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(REFERENCE_ROOT, yaml_rel))
    if opts:
        cfg.merge_from_list(list(opts))
    return cfg


def reference_model(cfg):
    install()
    from slowfast.models.build import MODEL_REGISTRY
    return MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)
