"""Model builders ``SlowFast`` and ``ResNet`` (C2D / I3D / Slow) on the fused engine.

Same cfg keys, module tree and state_dict names as slowfast/models/video_model_builder.py:172-441
(SlowFast) and :444-660 (ResNet), so reference checkpoints load unchanged; the graph is assembled
from per-stage tables instead of the reference's unrolled constructor."""
import torch
import torch.nn as nn

from . import engine, ops
from .engine import ConvUnit, FuseFn, as_cl, hold_notifications
from .heads import ResNetBasicHead, ResNetRoIHead
from .registry import MODEL_REGISTRY
from .resblocks import ResStage
from .stems import VideoModelStem

# blocks per stage (res2..res5) by depth
_STAGE_DEPTH = {18: (2, 2, 2, 2), 50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}
# temporal kernel of [conv1, res2, res3, res4, res5] per pathway
_TEMPORAL_KERNELS = {
    "2d": [[[1]]] * 5,
    "c2d": [[[1]]] * 5,
    "slow_c2d": [[[1]]] * 5,
    "i3d": [[[5]], [[3]], [[3, 1]], [[3, 1]], [[1, 3]]],
    "slow_i3d": [[[5]], [[3]], [[3, 1]], [[3, 1]], [[1, 3]]],
    "slow": [[[1]], [[1]], [[1]], [[3]], [[3]]],
    "slowfast": [[[1], [5]], [[1], [3]], [[1], [3]], [[3], [3]], [[3], [3]]],
}
# pooling applied after res2, per pathway
_POOL1 = {"2d": [[1, 1, 1]], "c2d": [[2, 1, 1]], "slow_c2d": [[1, 1, 1]], "i3d": [[2, 1, 1]],
          "slow_i3d": [[1, 1, 1]], "slow": [[1, 1, 1]], "slowfast": [[1, 1, 1], [1, 1, 1]]}


from .batchnorm import get_norm, num_splits_of, run_in_splits  # noqa: E402,F401  (batchnorm_helper.py:16-37)


def init_weights(model, fc_init_std=0.01, zero_init_final_bn=True, zero_init_final_conv=False):
    """ResNet-style init with the reference's rules (slowfast/utils/weight_init_helper.py:10-54):
    Conv3d Kaiming-normal(fan_out, relu); BN gamma 1 (0 on the block-final BN when requested), beta 0;
    Linear N(0, fc_init_std), bias 0."""
    for m in model.modules():
        if isinstance(m, nn.Conv3d):
            if getattr(m, "final_conv", False) and zero_init_final_conv:
                nn.init.zeros_(m.weight)
            else:
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, (nn.BatchNorm3d, nn.BatchNorm2d, nn.BatchNorm1d)):
            zero = getattr(m, "transform_final_bn", False) and zero_init_final_bn
            if m.weight is not None:
                nn.init.constant_(m.weight, 0.0 if zero else 1.0)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, mean=0.0, std=fc_init_std)
            if m.bias is not None:
                nn.init.zeros_(m.bias)


class FuseFastToSlow(nn.Module):
    """Lateral connection Fast -> Slow (video_model_builder.py:112-169): children conv_f2s, bn, relu."""

    def __init__(self, dim_in, fusion_conv_channel_ratio, fusion_kernel, alpha, eps=1e-5, bn_mmt=0.1,
                 inplace_relu=True, norm_module=nn.BatchNorm3d):
        super().__init__()
        self.conv_f2s = nn.Conv3d(dim_in, dim_in * fusion_conv_channel_ratio, kernel_size=(fusion_kernel, 1, 1),
                                  stride=(alpha, 1, 1), padding=(fusion_kernel // 2, 0, 0), bias=False)
        self.bn = norm_module(num_features=dim_in * fusion_conv_channel_ratio, eps=eps, momentum=bn_mmt)
        self.relu = nn.ReLU(inplace_relu)
        self._unit = ConvUnit(self.conv_f2s, self.bn)

    def forward(self, x):
        if not self.training and self.__dict__.get("_sf_infer"):
            return self._infer(x)
        x_s_fuse, x_f = FuseFn.apply(x[0], x[1], self, self.conv_f2s.weight, self.bn.weight, self.bn.bias)
        return [x_s_fuse, x_f]

    # inference fusion (slowfast_amd.inference): the lateral conv (BatchNorm folded, ReLU in the epilogue) writes its
    # channel slice of the concatenated buffer directly
    def _sf_fold(self):
        self._unit.fold()

    def _infer(self, x):
        x_s, x_f = as_cl(x[0]), as_cl(x[1])
        N, Cs, T, H, W = x_s.shape
        Cf = self.conv_f2s.out_channels
        cat = ops.cl_empty((N, Cs + Cf, T, H, W), x_s.device)
        ops.bn_act(x_s, out=cat[:, :Cs])
        self._unit.infer(x_f, relu=True, out=cat[:, Cs:])
        return [cat, x_f]


class PathwayPoolFn(torch.autograd.Function):
    """The non-overlapping nn.MaxPool3d applied after res2 (pathway{p}_pool, video_model_builder.py:330-338, 575-581:
    kernel == stride, no padding) on the sf_pool3d kernels (byte arg-max, gather backward) instead of ATen."""

    @staticmethod
    def forward(ctx, x, kernel, pool=None):
        from .nonlocal_block import pool3d_fwd
        x = as_cl(x)
        out, arg = pool3d_fwd(x, tuple(kernel))
        if engine.CAPTURE is not None:
            engine.CAPTURE.append({"kind": "pathway_pool", "mod": pool, "argmax": arg, "kernel": tuple(kernel),
                                   "in_shape": tuple(x.shape)})
        ctx.arg, ctx.in_shape, ctx.kernel = arg, tuple(x.shape), tuple(kernel)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .nonlocal_block import pool3d_bwd
        return pool3d_bwd(as_cl(dout), ctx.arg, ctx.in_shape, ctx.kernel), None, None


def _pathway_pool(pool, x):
    assert tuple(pool.stride) == tuple(pool.kernel_size) and not any(_t(pool.padding)), "pathway pools do not overlap"
    return PathwayPoolFn.apply(x, tuple(pool.kernel_size), pool)


def _t(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


def _bump_batches_tracked(model):
    """nn.BatchNorm3d.num_batches_tracked += 1 for every BN, in one foreach call per training forward."""
    bufs = model.__dict__.get("_nbt")
    if bufs is None:
        bufs = model.__dict__["_nbt"] = [m.num_batches_tracked for m in model.modules()
                                         if isinstance(m, nn.modules.batchnorm._BatchNorm)
                                         and m.num_batches_tracked is not None and not m.__dict__.get("_sf_no_bump")]
    if bufs:
        torch._foreach_add_(bufs, 1)


class _ResNetBase(nn.Module):
    def _stage(self, cfg, idx, dim_in, dim_out, dim_inner, depth):
        """res{idx+2}: idx indexes cfg lists (0 -> res2)."""
        P = self.num_pathways
        return ResStage(
            dim_in=dim_in, dim_out=dim_out, dim_inner=dim_inner,
            temp_kernel_sizes=_TEMPORAL_KERNELS[cfg.MODEL.ARCH][idx + 1],
            stride=cfg.RESNET.SPATIAL_STRIDES[idx], num_blocks=[depth] * P, num_groups=[cfg.RESNET.NUM_GROUPS] * P,
            num_block_temp_kernel=cfg.RESNET.NUM_BLOCK_TEMP_KERNEL[idx], nonlocal_inds=cfg.NONLOCAL.LOCATION[idx],
            nonlocal_group=cfg.NONLOCAL.GROUP[idx], nonlocal_pool=cfg.NONLOCAL.POOL[idx],
            instantiation=cfg.NONLOCAL.INSTANTIATION, trans_func_name=cfg.RESNET.TRANS_FUNC,
            stride_1x1=cfg.RESNET.STRIDE_1X1, inplace_relu=cfg.RESNET.INPLACE_RELU,
            dilation=cfg.RESNET.SPATIAL_DILATIONS[idx], norm_module=self.norm_module)

    def _head(self, cfg, dim_in, frames):
        pool = _POOL1[cfg.MODEL.ARCH]
        if cfg.DETECTION.ENABLE:              # video_model_builder.py:369-390, 611-622
            P = len(dim_in)
            return ResNetRoIHead(dim_in=dim_in, num_classes=cfg.MODEL.NUM_CLASSES,
                                 pool_size=[[frames[p] // pool[p][0], 1, 1] for p in range(P)],
                                 resolution=[[cfg.DETECTION.ROI_XFORM_RESOLUTION] * 2] * P,
                                 scale_factor=[cfg.DETECTION.SPATIAL_SCALE_FACTOR] * P,
                                 dropout_rate=cfg.MODEL.DROPOUT_RATE, act_func=cfg.MODEL.HEAD_ACT,
                                 aligned=cfg.DETECTION.ALIGNED, detach_final_fc=cfg.MODEL.DETACH_FINAL_FC)
        if cfg.MULTIGRID.SHORT_CYCLE or cfg.MODEL.MODEL_NAME == "ContrastiveModel":
            pool_size = [None] * len(dim_in)
        else:
            s = cfg.DATA.TRAIN_CROP_SIZE // 32
            pool_size = [[frames[p] // pool[p][0], s // pool[p][1], s // pool[p][2]] for p in range(len(dim_in))]
        return ResNetBasicHead(dim_in=dim_in, num_classes=cfg.MODEL.NUM_CLASSES, pool_size=pool_size,
                               dropout_rate=cfg.MODEL.DROPOUT_RATE, act_func=cfg.MODEL.HEAD_ACT,
                               detach_final_fc=cfg.MODEL.DETACH_FINAL_FC, cfg=cfg)

    def _finish(self, cfg):
        init_weights(self, cfg.MODEL.FC_INIT_STD, cfg.RESNET.ZERO_INIT_FINAL_BN, cfg.RESNET.ZERO_INIT_FINAL_CONV)


@MODEL_REGISTRY.register()
class SlowFast(_ResNetBase):
    """Two-pathway SlowFast network; forward(x=[slow NCTHW, fast NCTHW]) -> logits (B, num_classes)."""

    def __init__(self, cfg):
        super().__init__()
        self.norm_module = get_norm(cfg)
        self.cfg = cfg
        self.enable_detection = cfg.DETECTION.ENABLE
        self.num_pathways = 2
        arch = cfg.MODEL.ARCH
        assert arch in _POOL1 and len(_POOL1[arch]) == 2 and cfg.RESNET.DEPTH in _STAGE_DEPTH
        depths = _STAGE_DEPTH[cfg.RESNET.DEPTH]
        w = cfg.RESNET.WIDTH_PER_GROUP
        inner = cfg.RESNET.NUM_GROUPS * w
        beta_inv, ratio = cfg.SLOWFAST.BETA_INV, cfg.SLOWFAST.FUSION_CONV_CHANNEL_RATIO
        lateral = beta_inv // ratio            # slow width / lateral width
        tk = _TEMPORAL_KERNELS[arch]
        self.s1 = VideoModelStem(
            dim_in=cfg.DATA.INPUT_CHANNEL_NUM, dim_out=[w, w // beta_inv],
            kernel=[tk[0][0] + [7, 7], tk[0][1] + [7, 7]], stride=[[1, 2, 2]] * 2,
            padding=[[tk[0][0][0] // 2, 3, 3], [tk[0][1][0] // 2, 3, 3]], norm_module=self.norm_module)

        def fuse(width):
            return FuseFastToSlow(width // beta_inv, ratio, cfg.SLOWFAST.FUSION_KERNEL_SZ, cfg.SLOWFAST.ALPHA,
                                  norm_module=self.norm_module)

        self.s1_fuse = fuse(w)
        width_in = w
        for i, depth in enumerate(depths):      # res2..res5: output width w*4, w*8, w*16, w*32
            width_out = w * 4 * (2 ** i)
            stage = self._stage(cfg, i,
                                dim_in=[width_in + width_in // lateral, width_in // beta_inv],
                                dim_out=[width_out, width_out // beta_inv],
                                dim_inner=[inner * 2 ** i, inner * 2 ** i // beta_inv], depth=depth)
            setattr(self, f"s{i + 2}", stage)
            if i < 3:
                f2s = fuse(width_out)
                setattr(self, f"s{i + 2}_fuse", f2s)
                # the stage's last Slow block writes its output into the lateral connection's concatenated buffer
                # (engine.ResBlockFn / FuseFn) -- unless a Nonlocal block sits between them
                last = depth - 1
                if not hasattr(stage, f"pathway0_nonlocal{last}"):
                    getattr(stage, f"pathway0_res{last}")._cat_extra = f2s.conv_f2s.out_channels
            if i == 0:
                for p in range(2):
                    ps = _POOL1[arch][p]
                    self.add_module(f"pathway{p}_pool", nn.MaxPool3d(kernel_size=ps, stride=ps, padding=[0, 0, 0]))
            width_in = width_out
        frames = [cfg.DATA.NUM_FRAMES // cfg.SLOWFAST.ALPHA, cfg.DATA.NUM_FRAMES]
        self.head = self._head(cfg, [w * 32, w * 32 // beta_inv], frames)
        self._finish(cfg)

    def forward(self, x, bboxes=None):
        if self.training:
            _bump_batches_tracked(self)
            S = num_splits_of(self)
            hold_notifications(S, self.parameters() if S > 1 else None)
            if S > 1:                    # SubBatchNorm3d: S sub-batch passes (batchnorm.run_in_splits)
                assert bboxes is None, "detection batches are not split"
                return run_in_splits(self, self._forward, list(x), S)
        return self._forward(x, bboxes)

    def _forward(self, x, bboxes=None):
        x = self.s1_fuse(self.s1(list(x)))
        x = self.s2_fuse(self.s2(x))
        for p in range(self.num_pathways):
            pool = getattr(self, f"pathway{p}_pool")
            if tuple(pool.kernel_size) != (1, 1, 1):
                x[p] = _pathway_pool(pool, x[p])
        x = engine.cut(x)                # stage boundaries: backward segments of step.TrainStep (identity otherwise)
        x = engine.cut(self.s3_fuse(self.s3(x)))
        x = engine.cut(self.s4_fuse(self.s4(x)))
        x = self.s5(x)
        return self.head(x, bboxes) if self.enable_detection else self.head(x)


@MODEL_REGISTRY.register()
class ResNet(_ResNetBase):
    """Single-pathway ResNet video models (C2D, I3D, Slow); forward(x=[NCTHW]) -> logits."""

    def __init__(self, cfg):
        super().__init__()
        self.norm_module = get_norm(cfg)
        self.cfg = cfg
        self.enable_detection = cfg.DETECTION.ENABLE
        self.num_pathways = 1
        arch = cfg.MODEL.ARCH
        assert arch in _POOL1 and len(_POOL1[arch]) == 1 and cfg.RESNET.DEPTH in _STAGE_DEPTH
        depths = _STAGE_DEPTH[cfg.RESNET.DEPTH]
        w = cfg.RESNET.WIDTH_PER_GROUP
        inner = cfg.RESNET.NUM_GROUPS * w
        tk = _TEMPORAL_KERNELS[arch]
        self.s1 = VideoModelStem(dim_in=cfg.DATA.INPUT_CHANNEL_NUM, dim_out=[w], kernel=[tk[0][0] + [7, 7]],
                                 stride=[[1, 2, 2]], padding=[[tk[0][0][0] // 2, 3, 3]], norm_module=self.norm_module)
        width_in = w
        for i, depth in enumerate(depths):
            width_out = w * 4 * (2 ** i)
            setattr(self, f"s{i + 2}", self._stage(cfg, i, [width_in], [width_out], [inner * 2 ** i], depth))
            if i == 0:
                ps = _POOL1[arch][0]
                self.add_module("pathway0_pool", nn.MaxPool3d(kernel_size=ps, stride=ps, padding=[0, 0, 0]))
            width_in = width_out
        self.head = self._head(cfg, [w * 32], [cfg.DATA.NUM_FRAMES])
        self._finish(cfg)

    def forward(self, x, bboxes=None):
        if self.training:
            _bump_batches_tracked(self)
            S = num_splits_of(self)
            hold_notifications(S, self.parameters() if S > 1 else None)
            if S > 1:                    # SubBatchNorm3d: S sub-batch passes (batchnorm.run_in_splits)
                assert bboxes is None, "detection batches are not split"
                return run_in_splits(self, self._forward, list(x), S)
        return self._forward(x, bboxes)

    def _forward(self, x, bboxes=None):
        x = self.s2(self.s1(list(x)))
        pool = self.pathway0_pool
        if tuple(pool.kernel_size) != (1, 1, 1):
            x[0] = _pathway_pool(pool, x[0])
        x = self.s4(engine.cut(self.s3(engine.cut(x))))
        x = self.s5(engine.cut(x))
        return self.head(x, bboxes) if self.enable_detection else self.head(x)
