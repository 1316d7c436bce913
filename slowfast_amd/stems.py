"""Stem drop-ins: ``ResNetBasicStem`` / ``VideoModelStem`` with the reference's constructor signatures and
state_dict keys (slowfast/models/stem_helper.py:20-201), executed by the fused engine schedule
conv -> [BN statistics in the conv epilogue] -> BN+ReLU+MaxPool in one pass (engine.StemFn)."""
import torch.nn as nn

from . import ops
from .engine import ConvUnit, StemConvUnit, StemFn, run_pathways


class ResNetBasicStem(nn.Module):
    def __init__(self, dim_in, dim_out, kernel, stride, padding, inplace_relu=True, eps=1e-5, bn_mmt=0.1,
                 norm_module=nn.BatchNorm3d):
        super().__init__()
        self.kernel, self.stride, self.padding = kernel, stride, padding
        self.inplace_relu, self.eps, self.bn_mmt = inplace_relu, eps, bn_mmt
        # parameter containers (never called): keep names/shapes of the reference checkpoint format
        self.conv = nn.Conv3d(dim_in, dim_out, tuple(kernel), stride=tuple(stride), padding=tuple(padding), bias=False)
        self.bn = norm_module(num_features=dim_out, eps=eps, momentum=bn_mmt)
        self.relu = nn.ReLU(inplace_relu)
        self.pool_layer = nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
        foldable = dim_in <= 4 and stride[2] % 2 == 0
        self._unit = (StemConvUnit if foldable else ConvUnit)(self.conv, self.bn)

    def forward(self, x):
        if not self.training and self.__dict__.get("_sf_infer"):
            return self._infer(x)
        return StemFn.apply(x, self, self.conv.weight, self.bn.weight, self.bn.bias)

    # inference fusion (slowfast_amd.inference): conv with BatchNorm folded + ReLU in one launch, then the max-pool
    def _sf_fold(self):
        self._unit.fold()

    def _infer(self, x):
        unit = self._unit
        xcl = unit.prepare_input(x) if isinstance(unit, StemConvUnit) else ops.to_cl(x)
        y = unit.infer(xcl, relu=True)
        k, s, p = self.pool_layer.kernel_size, self.pool_layer.stride, self.pool_layer.padding
        out, _ = ops.pool_fwd(y, k[1:], s[1:], p[1:], affine=None, want_argmax=False)
        return out


_STEMS = {"basic_stem": ResNetBasicStem}


def get_stem_func(name):
    if name not in _STEMS:
        raise AssertionError(f"Transformation function '{name}' not supported")
    return _STEMS[name]


class VideoModelStem(nn.Module):
    """One stem per pathway, registered as ``pathway{i}_stem``."""

    def __init__(self, dim_in, dim_out, kernel, stride, padding, inplace_relu=True, eps=1e-5, bn_mmt=0.1,
                 norm_module=nn.BatchNorm3d, stem_func_name="basic_stem"):
        super().__init__()
        lens = {len(dim_in), len(dim_out), len(kernel), len(stride), len(padding)}
        assert len(lens) == 1, f"Input pathway dimensions are not consistent: {lens}"
        self.num_pathways = len(dim_in)
        self.kernel, self.stride, self.padding = kernel, stride, padding
        self.inplace_relu, self.eps, self.bn_mmt = inplace_relu, eps, bn_mmt
        make = get_stem_func(stem_func_name)
        for i in range(self.num_pathways):
            self.add_module(f"pathway{i}_stem", make(dim_in[i], dim_out[i], kernel[i], stride[i], padding[i],
                                                     inplace_relu, eps, bn_mmt, norm_module))

    def forward(self, x):
        assert len(x) == self.num_pathways, f"Input tensor does not contain {self.num_pathways} pathway"
        return run_pathways(self.num_pathways, lambda i: getattr(self, f"pathway{i}_stem")(x[i]), x[0])
