"""MViT drop-ins: ``PatchEmbed``, ``Mlp``, ``MultiScaleAttention``, ``MultiScaleBlock``, ``TransformerBasicHead`` and
the ``MViT`` model builder with the reference's constructor signatures, cfg keys and state_dict names
(slowfast/models/attention.py:150-514, common.py:7-34, stem_helper.py:288-320, head_helper.py:491-563,
video_model_builder.py:805-1244), executed by the token-space engine (mvit_engine.py).

Scope of this path: MODE "conv", fused qkv, POOL_FIRST False, cls token on -- i.e. the MViTv2 family of
configs/Kinetics/MVITv2_*.yaml (decomposed relative positions, residual pooling, DIM_MUL_IN_ATT), the MViTv1 family of
configs/Kinetics/MVIT_B_*_CONV.yaml (learned absolute position embeddings, dimension change after the Mlp, blocks
without q pooling) and the plain video ViT of configs/masked_ssl/k400_VIT_*_FT.yaml (no pooling, mean pooling before
the final norm).  Options outside it raise NotImplementedError instead of silently running something else.
"""
import math
from functools import partial

import torch

from . import engine
import torch.nn as nn
from torch.nn.init import trunc_normal_

from .engine import StemConvUnit
from .mvit_engine import (AttentionPlan, ClsNormFn, LinearUnit, MultiScaleBlockFn, NormUnit, PatchEmbedFn, QKVUnit,
                          TokenNormFn)
from .registry import MODEL_REGISTRY


def round_width(width, multiplier, min_width=1, divisor=1):
    """slowfast/models/utils.py:10-23."""
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    width_out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if width_out < 0.9 * width:
        width_out += divisor
    return int(width_out)


class PatchEmbed(nn.Module):
    def __init__(self, dim_in=3, dim_out=768, kernel=(1, 16, 16), stride=(1, 4, 4), padding=(1, 7, 7), conv_2d=False):
        super().__init__()
        if conv_2d:
            raise NotImplementedError("2-D patch embedding (image models) is outside the video hot path")
        self.proj = nn.Conv3d(dim_in, dim_out, kernel_size=tuple(kernel), stride=tuple(stride), padding=tuple(padding))
        self._unit = StemConvUnit(self.proj, None)

    def forward(self, x, cls_token=None, pos_embed=None):
        """(B, 3, T, H, W) fp32 clip -> tokens (B, [1 +] T'H'W', C) fp16 (+ the absolute position embedding
        [1, N, C] when given) and the (B, C, T', H', W') shape."""
        out = PatchEmbedFn.apply(x, self, cls_token, pos_embed, self.proj.weight, self.proj.bias)
        g = self._unit.geom(self._unit.prepare_shape(x.shape))
        return out, (x.shape[0], self.proj.out_channels, g.To, g.Ho, g.Wo)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop_rate=0.0):
        super().__init__()
        if drop_rate > 0.0:
            raise NotImplementedError("Mlp dropout (MVIT.DROPOUT_RATE > 0) is not on the built path")
        if act_layer is not nn.GELU:
            raise NotImplementedError("Mlp activation other than nn.GELU")
        self.drop_rate = drop_rate
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self._fc1, self._fc2 = LinearUnit(self.fc1), LinearUnit(self.fc2)


class MultiScaleAttention(nn.Module):
    def __init__(self, dim, dim_out, input_size, num_heads=8, qkv_bias=False, drop_rate=0.0, kernel_q=(1, 1, 1),
                 kernel_kv=(1, 1, 1), stride_q=(1, 1, 1), stride_kv=(1, 1, 1), norm_layer=nn.LayerNorm,
                 has_cls_embed=True, mode="conv", pool_first=False, rel_pos_spatial=False, rel_pos_temporal=False,
                 rel_pos_zero_init=False, residual_pooling=False, separate_qkv=False):
        super().__init__()
        if mode != "conv" or drop_rate > 0.0:
            raise NotImplementedError("MultiScaleAttention: only mode='conv' (depthwise conv pooling), no dropout")
        # "Skip pooling with kernel and stride size of (1, 1, 1)" (attention.py:199-203)
        if math.prod(kernel_q) == 1 and math.prod(stride_q) == 1:
            kernel_q = ()
        if math.prod(kernel_kv) == 1 and math.prod(stride_kv) == 1:
            kernel_kv = ()
        self.pool_first, self.separate_qkv, self.drop_rate = pool_first, separate_qkv, drop_rate
        self.num_heads, self.dim_in, self.dim_out = num_heads, dim, dim_out
        head_dim = dim_out // num_heads
        self.scale = head_dim ** -0.5
        self.has_cls_embed, self.mode = has_cls_embed, mode
        pad_q, pad_kv = [int(q // 2) for q in kernel_q], [int(kv // 2) for kv in kernel_kv]
        if pool_first or separate_qkv:              # attention.py:188-193
            self.q = nn.Linear(dim, dim_out, bias=qkv_bias)
            self.k = nn.Linear(dim, dim_out, bias=qkv_bias)
            self.v = nn.Linear(dim, dim_out, bias=qkv_bias)
        else:
            self.qkv = nn.Linear(dim, dim_out * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim_out, dim_out)
        # POOL_FIRST pools the block input folded into heads: dim // heads conv channels (attention.py:236-241)
        dim_conv = dim // num_heads if pool_first else head_dim
        conv = partial(nn.Conv3d, dim_conv, dim_conv, groups=dim_conv, bias=False)
        has_q, has_kv = len(kernel_q) > 0, len(kernel_kv) > 0
        self.pool_q = conv(tuple(kernel_q), stride=tuple(stride_q), padding=tuple(pad_q)) if has_q else None
        self.norm_q = norm_layer(dim_conv) if has_q else None
        self.pool_k = conv(tuple(kernel_kv), stride=tuple(stride_kv), padding=tuple(pad_kv)) if has_kv else None
        self.norm_k = norm_layer(dim_conv) if has_kv else None
        self.pool_v = conv(tuple(kernel_kv), stride=tuple(stride_kv), padding=tuple(pad_kv)) if has_kv else None
        self.norm_v = norm_layer(dim_conv) if has_kv else None
        self.rel_pos_spatial, self.rel_pos_temporal = rel_pos_spatial, rel_pos_temporal
        if rel_pos_spatial:
            assert input_size[1] == input_size[2]
            size = input_size[1]
            q_size = size // stride_q[1] if len(stride_q) > 0 else size
            kv_size = size // stride_kv[1] if len(stride_kv) > 0 else size
            rel_sp_dim = 2 * max(q_size, kv_size) - 1
            self.rel_pos_h = nn.Parameter(torch.zeros(rel_sp_dim, head_dim))
            self.rel_pos_w = nn.Parameter(torch.zeros(rel_sp_dim, head_dim))
            if not rel_pos_zero_init:
                trunc_normal_(self.rel_pos_h, std=0.02)
                trunc_normal_(self.rel_pos_w, std=0.02)
        if rel_pos_temporal:
            self.rel_pos_t = nn.Parameter(torch.zeros(2 * input_size[0] - 1, head_dim))
            if not rel_pos_zero_init:
                trunc_normal_(self.rel_pos_t, std=0.02)
        self.residual_pooling = residual_pooling
        self._proj = LinearUnit(self.proj)
        if pool_first:
            self._q, self._k, self._v = LinearUnit(self.q), LinearUnit(self.k), LinearUnit(self.v)
        else:
            self._qkv = QKVUnit(self.q, self.k, self.v) if separate_qkv else LinearUnit(self.qkv)
        self._norm_q = NormUnit(self.norm_q) if has_q else None
        self._norm_k, self._norm_v = (NormUnit(self.norm_k), NormUnit(self.norm_v)) if has_kv else (None, None)


class MultiScaleBlock(nn.Module):
    def __init__(self, dim, dim_out, num_heads, input_size, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop_rate=0.0,
                 drop_path=0.0, layer_scale_init_value=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm, up_rate=None,
                 kernel_q=(1, 1, 1), kernel_kv=(1, 1, 1), stride_q=(1, 1, 1), stride_kv=(1, 1, 1), mode="conv",
                 has_cls_embed=True, pool_first=False, rel_pos_spatial=False, rel_pos_temporal=False,
                 rel_pos_zero_init=False, residual_pooling=False, dim_mul_in_att=False, separate_qkv=False):
        super().__init__()
        if layer_scale_init_value > 0 or (up_rate is not None and up_rate > 1):
            raise NotImplementedError("layer scale / up_rate")
        self.dim, self.dim_out = dim, dim_out
        self.norm1 = norm_layer(dim)
        self.dim_mul_in_att = dim_mul_in_att
        kernel_skip = [s + 1 if s > 1 else s for s in stride_q]
        stride_skip = stride_q
        padding_skip = [int(skip // 2) for skip in kernel_skip]
        att_dim = dim_out if dim_mul_in_att else dim
        self.attn = MultiScaleAttention(
            dim, att_dim, num_heads=num_heads, input_size=input_size, qkv_bias=qkv_bias, drop_rate=drop_rate,
            kernel_q=kernel_q, kernel_kv=kernel_kv, stride_q=stride_q, stride_kv=stride_kv, norm_layer=norm_layer,
            has_cls_embed=has_cls_embed, mode=mode, pool_first=pool_first, rel_pos_spatial=rel_pos_spatial,
            rel_pos_temporal=rel_pos_temporal, rel_pos_zero_init=rel_pos_zero_init, residual_pooling=residual_pooling,
            separate_qkv=separate_qkv)
        self.drop_path_rate = drop_path
        self.drop_path = nn.Identity()          # stochastic depth: see forward()
        self.norm2 = norm_layer(att_dim)
        self.has_cls_embed = has_cls_embed
        self.mlp = Mlp(in_features=att_dim, hidden_features=int(att_dim * mlp_ratio), out_features=dim_out,
                       act_layer=act_layer, drop_rate=drop_rate)
        self.gamma_1, self.gamma_2 = None, None
        self._proj = None
        if dim != dim_out:
            self.proj = nn.Linear(dim, dim_out)
            self._proj = LinearUnit(self.proj)
        self.pool_skip = (nn.MaxPool3d(kernel_skip, stride_skip, padding_skip, ceil_mode=False)
                          if len(stride_skip) > 0 and math.prod(stride_skip) > 1 else None)
        self._norm1, self._norm2 = NormUnit(self.norm1), NormUnit(self.norm2)
        self._plans = {}

    def _plan(self, B, thw, device):
        key = (B, tuple(thw), str(device))
        p = self._plans.get(key)
        if p is None:
            p = self._plans[key] = AttentionPlan(self.attn, B, thw, device)
        return p

    @property
    def _param_list(self):
        plist = self.__dict__.get("_plist")
        if plist is None:
            plist = self.__dict__["_plist"] = list(self.parameters())
        return plist

    def _drop_scales(self, B, device):
        """Per-sample residual-branch scales of the two drop_path() calls (common.py:46-59): floor(keep + u) / keep."""
        fixed = self.__dict__.get("_fixed_drop_scales")       # tests pin the masks
        if fixed is not None:
            return tuple(t.to(device=device, dtype=torch.float32) for t in fixed)
        keep = 1.0 - self.drop_path_rate
        u = torch.rand((2, B), dtype=torch.float32, device=device)
        s = torch.floor(keep + u) / keep
        return s[0].contiguous(), s[1].contiguous()

    def forward(self, x, thw_shape=None, side=None):
        """``side`` (not in the reference's signature): the fp32 side rows of the residual stream travelling with x
        (mvit_engine.ResidSide; MViT.forward passes it from block to block, None = 16-bit stream only)."""
        drop = None
        if self.training and self.drop_path_rate > 0.0:
            drop = self._drop_scales(x.shape[0], x.device)
        out = MultiScaleBlockFn.apply(x, self, tuple(thw_shape), drop, side, *self._param_list)
        return out, list(self._plan(x.shape[0], thw_shape, x.device).q_thw)


class TransformerBasicHead(nn.Module):
    """Dropout -> Linear on the (B, C) cls features, fp32 torch ops (< 1 MMAC; head_helper.py:491-563)."""

    def __init__(self, dim_in, num_classes, dropout_rate=0.0, act_func="softmax", cfg=None):
        super().__init__()
        if dropout_rate > 0.0:
            self.dropout = nn.Dropout(dropout_rate)
        assert cfg is None or cfg.CONTRASTIVE.NUM_MLP_LAYERS == 1, "MLP heads belong to the SSL models (out of scope)"
        self.projection = nn.Linear(dim_in, num_classes, bias=True)
        self.detach_final_fc = cfg.MODEL.DETACH_FINAL_FC if cfg is not None else False
        if act_func == "softmax":
            self.act = nn.Softmax(dim=1)
        elif act_func == "sigmoid":
            self.act = nn.Sigmoid()
        elif act_func == "none":
            self.act = None
        else:
            raise NotImplementedError(f"{act_func} is not supported as an activationfunction.")

    def forward(self, x):
        x = x.float()
        if hasattr(self, "dropout"):
            x = self.dropout(x)
        if self.detach_final_fc:
            x = x.detach()
        x = self.projection(x)
        if not self.training and self.act is not None:
            x = self.act(x)
        return x.view(x.shape[0], -1)


@MODEL_REGISTRY.register()
class MViT(nn.Module):
    """MViTv2 video transformer; forward(x=[clip NCTHW]) -> logits (B, num_classes)."""

    def __init__(self, cfg):
        super().__init__()
        assert cfg.DATA.TRAIN_CROP_SIZE == cfg.DATA.TEST_CROP_SIZE
        m = cfg.MVIT
        unsupported = [k for k, bad in (
            ("PATCH_2D", m.PATCH_2D),
            ("USE_FIXED_SINCOS_POS", m.USE_FIXED_SINCOS_POS), ("NORM_STEM", m.NORM_STEM),
            ("LAYER_SCALE_INIT_VALUE", m.LAYER_SCALE_INIT_VALUE > 0), ("DROPOUT_RATE", m.DROPOUT_RATE > 0)) if bad]
        # MODEL.ACT_CHECKPOINT (video_model_builder.py:1041-1042: torch checkpoint_wrapper around every block) trades
        # recomputation for activation memory and changes no value; the engine already saves only GEMM / pooling
        # outputs and sizes for 288 GB of HBM, so the key is accepted and has no effect.
        if unsupported or m.NORM != "layernorm" or m.MODE != "conv":
            raise NotImplementedError(f"MViT options outside the built video path: {unsupported}")
        self.cfg = cfg
        self.enable_detection, self.enable_rev = bool(cfg.DETECTION.ENABLE), bool(m.REV.ENABLE)
        assert not (self.enable_detection and self.enable_rev), "rev does not support detection"
        self.patch_stride = list(m.PATCH_STRIDE)
        self.T = cfg.DATA.NUM_FRAMES // self.patch_stride[0]
        self.H = cfg.DATA.TRAIN_CROP_SIZE // self.patch_stride[1]
        self.W = cfg.DATA.TRAIN_CROP_SIZE // self.patch_stride[2]
        embed_dim, num_heads, depth = m.EMBED_DIM, m.NUM_HEADS, m.DEPTH
        self.drop_rate = m.DROPOUT_RATE
        self.cls_embed_on, self.use_mean_pooling = bool(m.CLS_EMBED_ON), m.USE_MEAN_POOLING
        self.use_abs_pos, self.sep_pos_embed = m.USE_ABS_POS, m.SEP_POS_EMBED
        self.rel_pos_spatial, self.rel_pos_temporal = m.REL_POS_SPATIAL, m.REL_POS_TEMPORAL
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.num_classes = cfg.MODEL.NUM_CLASSES
        self.patch_embed = PatchEmbed(dim_in=cfg.DATA.INPUT_CHANNEL_NUM[0], dim_out=embed_dim, kernel=m.PATCH_KERNEL,
                                      stride=m.PATCH_STRIDE, padding=m.PATCH_PADDING)
        self.input_dims = [cfg.DATA.NUM_FRAMES, cfg.DATA.TRAIN_CROP_SIZE, cfg.DATA.TRAIN_CROP_SIZE]
        self.patch_dims = [self.input_dims[i] // self.patch_stride[i] for i in range(3)]
        dpr = [x.item() for x in torch.linspace(0, m.DROPPATH_RATE, depth)]
        if self.cls_embed_on:
            self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        if self.use_abs_pos:                       # video_model_builder.py:888-911
            if self.sep_pos_embed:
                self.pos_embed_spatial = nn.Parameter(torch.zeros(1, self.patch_dims[1] * self.patch_dims[2], embed_dim))
                self.pos_embed_temporal = nn.Parameter(torch.zeros(1, self.patch_dims[0], embed_dim))
                if self.cls_embed_on:
                    self.pos_embed_class = nn.Parameter(torch.zeros(1, 1, embed_dim))
            else:
                self.pos_embed = nn.Parameter(torch.zeros(1, math.prod(self.patch_dims) + int(self.cls_embed_on), embed_dim))
        dim_mul, head_mul = torch.ones(depth + 1), torch.ones(depth + 1)
        for i, v in m.DIM_MUL:
            dim_mul[i] = v
        for i, v in m.HEAD_MUL:
            head_mul[i] = v
        pool_q, pool_kv = [[] for _ in range(depth)], [[] for _ in range(depth)]
        stride_q, stride_kv = [[] for _ in range(depth)], [[] for _ in range(depth)]
        def kernel_for(stride):                    # video_model_builder.py:926-933, 950-957
            return list(m.POOL_KVQ_KERNEL) if m.POOL_KVQ_KERNEL is not None else [v + 1 if v > 1 else v for v in stride]

        for e in m.POOL_Q_STRIDE:
            stride_q[e[0]] = list(e[1:])
            pool_q[e[0]] = kernel_for(e[1:])
        kv_list = m.POOL_KV_STRIDE
        if m.POOL_KV_STRIDE_ADAPTIVE is not None:
            _kv = list(m.POOL_KV_STRIDE_ADAPTIVE)
            kv_list = []
            for i in range(depth):
                if len(stride_q[i]) > 0:
                    _kv = [max(_kv[d] // stride_q[i][d], 1) for d in range(3)]
                kv_list.append([i] + _kv)
        for e in kv_list:
            stride_kv[e[0]] = list(e[1:])
            pool_kv[e[0]] = kernel_for(e[1:])
        self.pool_q, self.pool_kv, self.stride_q, self.stride_kv = pool_q, pool_kv, stride_q, stride_kv
        self.norm_stem = None
        input_size = self.patch_dims
        if self.enable_rev:                        # video_model_builder.py:964-978
            assert not self.cls_embed_on, "rev does not allow cls token"
            if m.REV.RESPATH_FUSE != "concat":
                raise NotImplementedError("MVIT.REV.RESPATH_FUSE: only 'concat' (norm and head on both streams) is built")
            from .rev_mvit import ReversibleMViT, TwoStreamFusion
            self.rev_backbone = ReversibleMViT(cfg, self)
            embed_dim = round_width(embed_dim, dim_mul.prod(), divisor=num_heads)
            self.fuse = TwoStreamFusion(m.REV.RESPATH_FUSE, dim=2 * embed_dim)
            embed_dim = 2 * embed_dim              # the final norm and the head see the concatenated streams
        else:
            self.blocks = nn.ModuleList()
        tokens_out = []                                # patch tokens a block hands on
        for i in range(0 if self.enable_rev else depth):
            num_heads = round_width(num_heads, head_mul[i])
            if m.DIM_MUL_IN_ATT:
                dim_out = round_width(embed_dim, dim_mul[i], divisor=round_width(num_heads, head_mul[i]))
            else:                                  # MViTv1: the block's Mlp widens to the NEXT block's dimension
                dim_out = round_width(embed_dim, dim_mul[i + 1], divisor=round_width(num_heads, head_mul[i + 1]))
            self.blocks.append(MultiScaleBlock(
                dim=embed_dim, dim_out=dim_out, num_heads=num_heads, input_size=input_size, mlp_ratio=m.MLP_RATIO,
                qkv_bias=m.QKV_BIAS, drop_rate=self.drop_rate, drop_path=dpr[i], norm_layer=norm_layer,
                kernel_q=pool_q[i], kernel_kv=pool_kv[i], stride_q=stride_q[i], stride_kv=stride_kv[i], mode=m.MODE,
                has_cls_embed=self.cls_embed_on, pool_first=m.POOL_FIRST, rel_pos_spatial=self.rel_pos_spatial,
                rel_pos_temporal=self.rel_pos_temporal, rel_pos_zero_init=m.REL_POS_ZERO_INIT,
                residual_pooling=m.RESIDUAL_POOLING, dim_mul_in_att=m.DIM_MUL_IN_ATT, separate_qkv=m.SEPARATE_QKV))
            if len(stride_q[i]) > 0:
                input_size = [size // stride for size, stride in zip(input_size, stride_q[i])]
            tokens_out.append(math.prod(input_size))
            embed_dim = dim_out
        if not self.enable_rev:
            # SF_MVIT_RESID32=full: blocks from the last q-pooling block on (MViTv2-S: 14-15, 393 tokens) keep every row of the
            # residual stream in fp32 next to the 16-bit tensor (mvit_engine.ResidSide) when that stage is small; otherwise
            # (and by default everywhere) only the class-token row
            last_pool = max([i for i in range(depth) if len(stride_q[i]) > 0 and math.prod(stride_q[i]) > 1], default=None)
            for i, blk in enumerate(self.blocks):
                blk._resid32_full = last_pool is not None and i >= last_pool and tokens_out[i] <= 1024
        self.norm = norm_layer(embed_dim)
        self._norm_unit = NormUnit(self.norm)
        if self.enable_detection:                  # video_model_builder.py:1034-1045
            from .heads import ResNetRoIHead
            self.head = ResNetRoIHead(dim_in=[embed_dim], num_classes=self.num_classes,
                                      pool_size=[[cfg.DATA.NUM_FRAMES // self.patch_stride[0], 1, 1]],
                                      resolution=[[cfg.DETECTION.ROI_XFORM_RESOLUTION] * 2],
                                      scale_factor=[cfg.DETECTION.SPATIAL_SCALE_FACTOR],
                                      dropout_rate=cfg.MODEL.DROPOUT_RATE, act_func=cfg.MODEL.HEAD_ACT,
                                      aligned=cfg.DETECTION.ALIGNED)
        else:
            self.head = TransformerBasicHead(embed_dim, self.num_classes, dropout_rate=cfg.MODEL.DROPOUT_RATE,
                                             act_func=cfg.MODEL.HEAD_ACT, cfg=cfg)
        if self.use_abs_pos:                       # video_model_builder.py:1066-1075
            if self.sep_pos_embed:
                trunc_normal_(self.pos_embed_spatial, std=0.02)
                trunc_normal_(self.pos_embed_temporal, std=0.02)
                if self.cls_embed_on:
                    trunc_normal_(self.pos_embed_class, std=0.02)
            else:
                trunc_normal_(self.pos_embed, std=0.02)
        if self.cls_embed_on:
            trunc_normal_(self.cls_token, std=0.02)
        self.apply(self._init_weights)
        self.head.projection.weight.data.mul_(m.HEAD_INIT_SCALE)
        self.head.projection.bias.data.mul_(m.HEAD_INIT_SCALE)

    def _resid_side(self, x, pos):
        """fp32 side rows of the residual stream at the first block (mvit_engine.ResidSide): the class-token rows
        cls_token (+ its position embedding), exact.  None without a class token or with SF_MVIT_RESID32=0."""
        from . import mvit_engine
        if not (self.cls_embed_on and mvit_engine.RESID32):
            return None
        c = self.cls_token.detach().view(1, -1).float()
        if pos is not None:
            c = c + pos.detach()[0, :1].float()
        return mvit_engine.ResidSide(cls32=c.expand(x.shape[0], -1).contiguous())

    def _init_weights(self, m):
        """video_model_builder.py:1085-1092."""
        if isinstance(m, (nn.Linear, nn.Conv2d, nn.Conv3d)):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if isinstance(m, nn.Linear) and m.bias is not None:
                nn.init.constant_(m.bias, 0.02)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0.02)
            nn.init.constant_(m.weight, 1.0)

    def no_weight_decay(self):
        names = []
        if self.cfg.MVIT.ZERO_DECAY_POS_CLS:       # video_model_builder.py:1095-1117
            if self.use_abs_pos:
                names += (["pos_embed_spatial", "pos_embed_temporal", "pos_embed_class"] if self.sep_pos_embed
                          else ["pos_embed"])
            if self.rel_pos_spatial:
                names += ["rel_pos_h", "rel_pos_w", "rel_pos_hw"]
            if self.rel_pos_temporal:
                names += ["rel_pos_t"]
            if self.cls_embed_on:
                names += ["cls_token"]
        return names

    def forward(self, x, bboxes=None, return_attn=False):
        pos = None
        if self.use_abs_pos:                       # video_model_builder.py:1189-1203; tiny [1, N, C] parameter algebra
            if self.sep_pos_embed:
                pos = self.pos_embed_spatial.repeat(1, self.patch_dims[0], 1) + torch.repeat_interleave(
                    self.pos_embed_temporal, self.patch_dims[1] * self.patch_dims[2], dim=1)
                if self.cls_embed_on:
                    pos = torch.cat([self.pos_embed_class, pos], 1)
            else:
                pos = self.pos_embed
        x, bcthw = self.patch_embed(x[0], self.cls_token if self.cls_embed_on else None, pos)
        T, H, W = bcthw[-3], bcthw[-2], bcthw[-1]
        assert (T, H, W) == (self.T, self.H, self.W), bcthw
        thw = [T, H, W]
        if self.enable_rev:                        # _forward_reversible, video_model_builder.py:1141-1164
            x = self.rev_backbone(x)               # fuse("concat") is the identity on the concatenated streams
        else:
            ncut = 4 if len(self.blocks) >= 8 else max(len(self.blocks) // 3, 1)      # MViTv2-S: [0-3 | 4-7 | 8-11 | 12-15 + head]
            side = self._resid_side(x, pos)
            for i, blk in enumerate(self.blocks):
                if i and i % ncut == 0:                # backward segments of step.TrainStep (identity otherwise)
                    x = engine.cut(x)
                x, thw = blk(x, thw, side)
        if self.enable_detection:                  # video_model_builder.py:1218-1226
            x = TokenNormFn.apply(x, self, self.cls_embed_on, self.norm.weight, self.norm.bias)
            B, _, C = x.shape
            x = x.view(B, thw[0], thw[1], thw[2], C).permute(0, 4, 1, 2, 3)     # channels-last (B, C, T, H, W), no copy
            return self.head([x], bboxes)
        # video_model_builder.py:1154-1161, 1226-1238: mean of the patch tokens then norm / norm of the cls rows / norm then mean
        mode = "mean_norm" if self.use_mean_pooling else ("cls" if self.cls_embed_on else "norm_mean")
        x = ClsNormFn.apply(x, self, mode, self.cls_embed_on, None if self.enable_rev else side, self.norm.weight,
                            self.norm.bias)
        return self.head(x)
