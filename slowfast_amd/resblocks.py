"""Residual-stage drop-ins: ``BottleneckTransform``, ``ResBlock``, ``ResStage`` with the reference's
constructor signatures and state_dict keys (slowfast/models/resnet_helper.py:259-726).  ``ResBlock`` runs
the fully fused schedule of engine.ResBlockFn; ``BottleneckTransform`` on its own (API parity) runs the
three conv+BN units materialised."""
import torch.nn as nn

from .engine import ConvBNActFn, ConvUnit, ResBlockFn, as_cl, run_pathways


class BottleneckTransform(nn.Module):
    """Tx1x1 -> 1x3x3 -> 1x1x1 bottleneck; children a, a_bn, a_relu, b, b_bn, b_relu, c, c_bn."""

    def __init__(self, dim_in, dim_out, temp_kernel_size, stride, dim_inner, num_groups, stride_1x1=False,
                 inplace_relu=True, eps=1e-5, bn_mmt=0.1, dilation=1, norm_module=nn.BatchNorm3d, block_idx=0):
        super().__init__()
        assert num_groups == 1, "ResNeXt (grouped 1x3x3) is outside the SlowFast/C2D hot path built so far"
        self.temp_kernel_size = temp_kernel_size
        self._inplace_relu, self._eps, self._bn_mmt, self._stride_1x1 = inplace_relu, eps, bn_mmt, stride_1x1
        s_a, s_b = (stride, 1) if stride_1x1 else (1, stride)
        bn = dict(eps=eps, momentum=bn_mmt)
        self.a = nn.Conv3d(dim_in, dim_inner, (temp_kernel_size, 1, 1), stride=(1, s_a, s_a),
                           padding=(temp_kernel_size // 2, 0, 0), bias=False)
        self.a_bn = norm_module(num_features=dim_inner, **bn)
        self.a_relu = nn.ReLU(inplace=inplace_relu)
        self.b = nn.Conv3d(dim_inner, dim_inner, (1, 3, 3), stride=(1, s_b, s_b), padding=(0, dilation, dilation),
                           groups=num_groups, bias=False, dilation=(1, dilation, dilation))
        self.b_bn = norm_module(num_features=dim_inner, **bn)
        self.b_relu = nn.ReLU(inplace=inplace_relu)
        self.c = nn.Conv3d(dim_inner, dim_out, (1, 1, 1), bias=False)
        self.c.final_conv = True
        self.c_bn = norm_module(num_features=dim_out, **bn)
        self.c_bn.transform_final_bn = True
        self._a, self._b, self._c = ConvUnit(self.a, self.a_bn), ConvUnit(self.b, self.b_bn), ConvUnit(self.c, self.c_bn)
        self._chain = (self._a, self._b, self._c)

    def forward(self, x):
        for unit, relu in ((self._a, True), (self._b, True), (self._c, False)):
            x = ConvBNActFn.apply(x, unit, relu, self.training, *unit.params())
        return x


class BasicTransform(nn.Module):
    """Tx3x3 (stride) -> BN -> ReLU -> 1x3x3 (dilated) -> BN; children a, a_bn, a_relu, b, b_bn
    (slowfast/models/resnet_helper.py:27-115; the ResNet-18/34 style block, RESNET.TRANS_FUNC "basic_transform")."""

    def __init__(self, dim_in, dim_out, temp_kernel_size, stride, dim_inner=None, num_groups=1, stride_1x1=None,
                 inplace_relu=True, eps=1e-5, bn_mmt=0.1, dilation=1, norm_module=nn.BatchNorm3d, block_idx=0):
        super().__init__()
        self.temp_kernel_size = temp_kernel_size
        self._inplace_relu, self._eps, self._bn_mmt = inplace_relu, eps, bn_mmt
        bn = dict(eps=eps, momentum=bn_mmt)
        self.a = nn.Conv3d(dim_in, dim_out, (temp_kernel_size, 3, 3), stride=(1, stride, stride),
                           padding=(temp_kernel_size // 2, 1, 1), bias=False)
        self.a_bn = norm_module(num_features=dim_out, **bn)
        self.a_relu = nn.ReLU(inplace=inplace_relu)
        self.b = nn.Conv3d(dim_out, dim_out, (1, 3, 3), stride=(1, 1, 1), padding=(0, dilation, dilation),
                           dilation=(1, dilation, dilation), bias=False)
        self.b.final_conv = True
        self.b_bn = norm_module(num_features=dim_out, **bn)
        self.b_bn.transform_final_bn = True
        self._a, self._b = ConvUnit(self.a, self.a_bn), ConvUnit(self.b, self.b_bn)
        self._chain = (self._a, self._b)

    def forward(self, x):
        for unit, relu in ((self._a, True), (self._b, False)):
            x = ConvBNActFn.apply(x, unit, relu, self.training, *unit.params())
        return x


BottleneckTransform._block_fn = ResBlockFn
BasicTransform._block_fn = ResBlockFn
_TRANS = {"bottleneck_transform": BottleneckTransform, "basic_transform": BasicTransform}


def get_trans_func(name):
    if name not in _TRANS:
        raise AssertionError(f"Transformation function '{name}' not supported")
    return _TRANS[name]


class ResBlock(nn.Module):
    """relu(shortcut(x) + branch2(x)); projection shortcut (branch1, branch1_bn) when shape changes."""

    def __init__(self, dim_in, dim_out, temp_kernel_size, stride, trans_func, dim_inner, num_groups=1,
                 stride_1x1=False, inplace_relu=True, eps=1e-5, bn_mmt=0.1, dilation=1, norm_module=nn.BatchNorm3d,
                 block_idx=0, drop_connect_rate=0.0):
        super().__init__()
        self._inplace_relu, self._eps, self._bn_mmt = inplace_relu, eps, bn_mmt
        # The reference calls drop_path() without training=True here, i.e. it never fires
        # (resnet_helper.py:514-515, common.py:46-51); the rate is kept only for signature parity.
        self._drop_connect_rate = drop_connect_rate
        self._proj = None
        if dim_in != dim_out or stride != 1:
            self.branch1 = nn.Conv3d(dim_in, dim_out, kernel_size=1, stride=(1, stride, stride), padding=0,
                                     bias=False, dilation=1)
            self.branch1_bn = norm_module(num_features=dim_out, eps=eps, momentum=bn_mmt)
            self._proj = ConvUnit(self.branch1, self.branch1_bn)
        self.branch2 = trans_func(dim_in, dim_out, temp_kernel_size, stride, dim_inner, num_groups,
                                  stride_1x1=stride_1x1, inplace_relu=inplace_relu, dilation=dilation,
                                  norm_module=norm_module, block_idx=block_idx)
        self.relu = nn.ReLU(inplace_relu)
        self._block_fn = getattr(type(self.branch2), "_block_fn", None)   # fused schedule of this transform type

    @property
    def _param_list(self):
        plist = self.__dict__.get("_plist")
        if plist is None:
            plist = self.__dict__["_plist"] = list(self.parameters())
        return plist

    def forward(self, x):
        assert self._block_fn is not None, f"no fused schedule for {type(self.branch2).__name__}"
        if not self.training and self.__dict__.get("_sf_infer"):
            return self._infer(x)
        return self._block_fn.apply(x, self, *self._param_list)

    # inference fusion (slowfast_amd.inference): 3 launches (4 with the projection shortcut), BatchNorm folded into
    # the weights, ReLU and the residual addition in the GEMM epilogues
    def _sf_fold(self):
        t = self.branch2
        if not hasattr(t, "_chain"):
            fold = getattr(t, "_sf_fold_in_block", None)     # X3DTransform: its 1x1x1 convolutions fold, the
            return fold(self) if fold is not None else False  # depthwise / SE / Swish middle keeps running statistics
        for u in tuple(t._chain) + (self._proj,):
            if u is not None:
                u.fold()
        return True

    def _infer(self, x):
        x = as_cl(x)
        t = self.branch2
        if not hasattr(t, "_chain"):
            return t._infer_in_block(self, x)
        h = x
        for u in t._chain[:-1]:
            h = u.infer(h, relu=True)
        sc = x if self._proj is None else self._proj.infer(x)
        return t._chain[-1].infer(h, relu=True, resid=sc)


def _fold_time(x, n, t):
    """(N, C, T, H, W) channels-last tensor viewed as (n, C, t, H, W) with n*t == N*T (no copy)."""
    from . import ops
    x = ops.to_cl(x)
    N, C, T, H, W = x.shape
    rows = x.permute(0, 2, 3, 4, 1)                      # N, T, H, W, C strided view of the storage
    assert rows.stride(0) == T * rows.stride(1), "batch and time must be uniformly spaced to fold"
    return rows.reshape(n, t, H, W, C).permute(0, 4, 1, 2, 3)


class ResStage(nn.Module):
    """Per-pathway sequence of ResBlocks, registered as ``pathway{p}_res{i}``."""

    def __init__(self, dim_in, dim_out, stride, temp_kernel_sizes, num_blocks, dim_inner, num_groups,
                 num_block_temp_kernel, nonlocal_inds, nonlocal_group, nonlocal_pool, dilation,
                 instantiation="softmax", trans_func_name="bottleneck_transform", stride_1x1=False,
                 inplace_relu=True, norm_module=nn.BatchNorm3d, drop_connect_rate=0.0):
        super().__init__()
        P = len(num_blocks)
        assert all(num_block_temp_kernel[i] <= num_blocks[i] for i in range(P))
        assert len({len(v) for v in (dim_in, dim_out, temp_kernel_sizes, stride, num_blocks, dim_inner, num_groups,
                                     num_block_temp_kernel, nonlocal_inds, nonlocal_group)}) == 1
        self.num_blocks, self.nonlocal_group, self.num_pathways = num_blocks, nonlocal_group, P
        self._drop_connect_rate = drop_connect_rate
        # the first num_block_temp_kernel blocks use the stage's temporal kernel, the rest 1
        self.temp_kernel_sizes = [
            (temp_kernel_sizes[p] * num_blocks[p])[: num_block_temp_kernel[p]]
            + [1] * (num_blocks[p] - num_block_temp_kernel[p]) for p in range(P)]
        trans = get_trans_func(trans_func_name)
        for p in range(P):
            for i in range(num_blocks[p]):
                blk = ResBlock(dim_in[p] if i == 0 else dim_out[p], dim_out[p], self.temp_kernel_sizes[p][i],
                               stride[p] if i == 0 else 1, trans, dim_inner[p], num_groups[p],
                               stride_1x1=stride_1x1, inplace_relu=inplace_relu, dilation=dilation[p],
                               norm_module=norm_module, block_idx=i, drop_connect_rate=drop_connect_rate)
                self.add_module(f"pathway{p}_res{i}", blk)
                if i in nonlocal_inds[p]:
                    from .nonlocal_block import Nonlocal
                    self.add_module(f"pathway{p}_nonlocal{i}",
                                    Nonlocal(dim_out[p], dim_out[p] // 2, nonlocal_pool[p], instantiation=instantiation,
                                             norm_module=norm_module))

    def forward(self, inputs):
        # the pathways of a stage are independent: each on its own stream (engine.run_pathways)
        return run_pathways(self.num_pathways, lambda p: self._pathway(p, inputs[p]), inputs[0])

    def _pathway(self, p, x):
        for i in range(self.num_blocks[p]):
            x = getattr(self, f"pathway{p}_res{i}")(x)
            nln = getattr(self, f"pathway{p}_nonlocal{i}", None)
            if nln is not None:
                g = self.nonlocal_group[p]
                if g > 1:
                    # fold T into the batch around the block (resnet_helper.py:706-723); channels-last memory is
                    # N,T,H,W,C, so both folds are views of the same rows
                    b, c, t, h, w = x.shape
                    x = _fold_time(x, b * g, t // g)
                    x = _fold_time(nln(x), b, t)
                else:
                    x = nln(x)
        return x
