"""Reversible MViT drop-ins (slowfast/models/reversible_mvit.py): ``ReversibleMViT``, ``ReversibleBlock``,
``StageTransitionBlock``, ``AttentionSubBlock``, ``MLPSubblock`` and ``TwoStreamFusion`` (common.py:73-146) with the
reference's constructor signatures and state_dict names, executed by the token-space engine (mvit_engine.RevBlockFn /
StageTransitionFn).

Built family: configs/Kinetics/REV_MVIT_B_16x4_CONV.yaml -- MVIT.REV.PRE_Q_FUSION "avg" (the default), RES_PATH "conv",
RESPATH_FUSE "concat", no cls token (the reference's constructor asserts it, video_model_builder.py:966).  The reference
saves memory by re-deriving block inputs from block outputs in a custom backward (RevBackProp, :177-263); values and
gradients are those of the plain two-stream graph (:128-139), which is what runs here: activations are kept, sized for
288 GB of HBM.  Other fusion modes raise NotImplementedError.
"""
import torch
import torch.nn as nn

from .mvit import Mlp, MultiScaleAttention, round_width
from .mvit_engine import AttentionPlan, NormUnit, LinearUnit, RevBlockFn, StageTransitionFn


class TwoStreamFusion(nn.Module):
    """common.py:73-146, the parameter-free modes acting on the channel halves of x [B, N, 2C]."""

    def __init__(self, mode, dim=None, kernel=3, padding=1):
        super().__init__()
        self.mode = mode
        if mode not in ("add", "max", "min", "avg", "concat"):
            raise NotImplementedError(f"TwoStreamFusion mode {mode} (MLP fusion) is not on the built path")

    def forward(self, x):
        if self.mode == "concat":
            return x
        a, b = torch.chunk(x, 2, dim=2)
        if self.mode == "add":
            return a + b
        if self.mode == "avg":
            return (a + b) * 0.5
        return torch.maximum(a, b) if self.mode == "max" else torch.minimum(a, b)


class MLPSubblock(nn.Module):
    def __init__(self, dim, mlp_ratio, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm = norm_layer(dim, eps=1e-6, elementwise_affine=True)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=nn.GELU)
        self._norm = NormUnit(self.norm)


class AttentionSubBlock(nn.Module):
    def __init__(self, dim, input_size, num_heads, cfg, dim_out=None, kernel_q=(1, 1, 1), kernel_kv=(1, 1, 1),
                 stride_q=(1, 1, 1), stride_kv=(1, 1, 1), norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm = norm_layer(dim, eps=1e-6, elementwise_affine=True)
        self.thw = None                         # set by ReversibleMViT (reversible_mvit.py:119)
        m = cfg.MVIT
        self.attn = MultiScaleAttention(
            dim, dim_out, input_size=input_size, num_heads=num_heads, kernel_q=kernel_q, kernel_kv=kernel_kv,
            stride_q=stride_q, stride_kv=stride_kv, norm_layer=norm_layer, drop_rate=m.DROPOUT_RATE, qkv_bias=m.QKV_BIAS,
            has_cls_embed=m.CLS_EMBED_ON, mode=m.MODE, pool_first=m.POOL_FIRST, rel_pos_spatial=m.REL_POS_SPATIAL,
            rel_pos_temporal=m.REL_POS_TEMPORAL, rel_pos_zero_init=m.REL_POS_ZERO_INIT,
            residual_pooling=m.RESIDUAL_POOLING, separate_qkv=m.SEPARATE_QKV)
        self._norm = NormUnit(self.norm)


class _TwoStreamBlock(nn.Module):
    """Shared plumbing of the two block types: plan cache, parameter list, stochastic-depth scales."""

    def _init_engine(self):
        self._plans = {}

    def _plan(self, B, thw, device):
        key = (B, tuple(thw), str(device))
        p = self._plans.get(key)
        if p is None:
            p = self._plans[key] = AttentionPlan(self.F.attn, B, thw, device)
        return p

    def _half(self, B, device):
        key = ("half", B, str(device))
        h = self._plans.get(key)
        if h is None:
            h = self._plans[key] = torch.full((B,), 0.5, dtype=torch.float32, device=device)
        return h

    @property
    def _param_list(self):
        plist = self.__dict__.get("_plist")
        if plist is None:
            plist = self.__dict__["_plist"] = list(self.parameters())
        return plist

    def _drop_scale(self, B, device):
        """Per-sample scale floor(keep + u) / keep of drop_path() (common.py:46-59); one draw per block: the reference
        re-seeds the generator so that both branches of a ReversibleBlock use the same mask (reversible_mvit.py:500-519)."""
        if not (self.training and self.drop_path_rate > 0.0):
            return None
        fixed = self.__dict__.get("_fixed_drop_scale")        # tests pin the mask
        if fixed is not None:
            return fixed.to(device=device, dtype=torch.float32)
        keep = 1.0 - self.drop_path_rate
        return (torch.floor(keep + torch.rand((B,), dtype=torch.float32, device=device)) / keep).contiguous()


class ReversibleBlock(_TwoStreamBlock):
    def __init__(self, dim, input_size, dim_out, num_heads, mlp_ratio, qkv_bias, drop_path, kernel_q, kernel_kv, stride_q,
                 stride_kv, cfg, norm_layer=nn.LayerNorm, layer_id=0, **kwargs):
        super().__init__()
        self.drop_path_rate = drop_path
        self.F = AttentionSubBlock(dim=dim, input_size=input_size, num_heads=num_heads, cfg=cfg, dim_out=dim_out,
                                   kernel_q=kernel_q, kernel_kv=kernel_kv, stride_q=stride_q, stride_kv=stride_kv,
                                   norm_layer=norm_layer)
        self.G = MLPSubblock(dim=dim, mlp_ratio=mlp_ratio, norm_layer=norm_layer)
        self.layer_id = layer_id
        self._init_engine()

    def forward(self, X_1, X_2):
        drop = self._drop_scale(X_1.shape[0], X_1.device)
        return RevBlockFn.apply(X_1, X_2, self, tuple(self.F.thw), drop, *self._param_list)


class StageTransitionBlock(_TwoStreamBlock):
    def __init__(self, dim, input_size, dim_out, num_heads, mlp_ratio, qkv_bias, drop_path, kernel_q, kernel_kv, stride_q,
                 stride_kv, cfg, norm_layer=nn.LayerNorm, pre_q_fusion=None, layer_id=0):
        super().__init__()
        if pre_q_fusion != "avg" or cfg.MVIT.REV.RES_PATH != "conv":
            raise NotImplementedError("StageTransitionBlock: MVIT.REV.PRE_Q_FUSION 'avg' and RES_PATH 'conv' are built "
                                      f"(got {pre_q_fusion!r}, {cfg.MVIT.REV.RES_PATH!r})")
        self.drop_path_rate = drop_path
        self.F = AttentionSubBlock(dim=dim, input_size=input_size, num_heads=num_heads, cfg=cfg, dim_out=dim_out,
                                   kernel_q=kernel_q, kernel_kv=kernel_kv, stride_q=stride_q, stride_kv=stride_kv,
                                   norm_layer=norm_layer)
        self.G = MLPSubblock(dim=dim_out, mlp_ratio=mlp_ratio, norm_layer=norm_layer)
        assert self.F.attn.pool_q is not None, "a stage transition pools q (MVIT.POOL_Q_STRIDE at every BUFFER_LAYERS index)"
        self.layer_id = layer_id
        self.is_proj, self.has_cls_embed = False, cfg.MVIT.CLS_EMBED_ON
        self.is_conv, self.pool_first, self.mode = False, cfg.MVIT.POOL_FIRST, cfg.MVIT.MODE
        self.pre_q_fuse = TwoStreamFusion(pre_q_fusion, dim=dim)
        self.res_conv = True
        self._res_proj = None
        if dim != dim_out:
            self.is_proj = True
            self.res_proj = nn.Linear(dim, dim_out, bias=True)
            self._res_proj = LinearUnit(self.res_proj)
        self._init_engine()

    def forward(self, x):
        """x = (X_1, X_2): the two streams of the preceding reversible sequence (their channel concatenation in the
        reference)."""
        x1, x2 = x
        drop = self._drop_scale(x1.shape[0], x1.device)
        return StageTransitionFn.apply(x1, x2, self, tuple(self.F.thw), drop, *self._param_list)


class ReversibleMViT(nn.Module):
    """The two-stream encoder (reversible_mvit.py:12-174): layer construction follows :75-126."""

    def __init__(self, config, model):
        super().__init__()
        self.cfg = config
        m = config.MVIT
        embed_dim, depth, num_heads = m.EMBED_DIM, m.DEPTH, m.NUM_HEADS
        self.dropout = m.DROPOUT_RATE
        self.pre_q_fusion = m.REV.PRE_Q_FUSION
        dpr = [x.item() for x in torch.linspace(0, m.DROPPATH_RATE, depth)]
        input_size = model.patch_dims
        self.layers = nn.ModuleList([])
        self.no_custom_backward = False
        if m.NORM != "layernorm":
            raise NotImplementedError("Only supports layernorm.")
        from functools import partial
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        dim_mul, head_mul = torch.ones(depth + 1), torch.ones(depth + 1)
        for i, v in m.DIM_MUL:
            dim_mul[i] = v
        for i, v in m.HEAD_MUL:
            head_mul[i] = v
        buffers = list(m.REV.BUFFER_LAYERS)
        assert 0 not in buffers and (depth - 1) not in buffers and all(b + 1 not in buffers for b in buffers), \
            "every stage transition sits between reversible blocks (its input is the two-stream concatenation)"
        for i in range(depth):
            num_heads = round_width(num_heads, head_mul[i])
            embed_dim = round_width(embed_dim, dim_mul[i - 1] if i > 0 else 1.0, divisor=num_heads)
            dim_out = round_width(embed_dim, dim_mul[i], divisor=round_width(num_heads, head_mul[i + 1]))
            if i in buffers:
                layer_type = StageTransitionBlock
                input_mult = 2 if "concat" in self.pre_q_fusion else 1
            else:
                layer_type = ReversibleBlock
                input_mult = 1
            dimout_correction = 2 if (input_mult == 2 and "concat" in self.pre_q_fusion) else 1
            self.layers.append(layer_type(
                dim=embed_dim * input_mult, input_size=input_size, dim_out=dim_out * input_mult // dimout_correction,
                num_heads=num_heads, cfg=config, mlp_ratio=m.MLP_RATIO, qkv_bias=m.QKV_BIAS, drop_path=dpr[i],
                norm_layer=norm_layer, kernel_q=model.pool_q[i] if len(model.pool_q) > i else [],
                kernel_kv=model.pool_kv[i] if len(model.pool_kv) > i else [],
                stride_q=model.stride_q[i] if len(model.stride_q) > i else [],
                stride_kv=model.stride_kv[i] if len(model.stride_kv) > i else [], layer_id=i,
                pre_q_fusion=self.pre_q_fusion))
            self.layers[-1].F.thw = input_size
            if len(model.stride_q[i]) > 0:
                input_size = [size // stride for size, stride in zip(input_size, model.stride_q[i])]

    def forward(self, x):
        """x [B, N, C] -> the two streams' channel concatenation [B, N', 2C'] (reversible_mvit.py:141-174)."""
        assert self.dropout == 0.0
        a = h = x                                              # torch.cat([x, x], -1), split again by the blocks
        for layer in self.layers:
            if isinstance(layer, StageTransitionBlock):
                a = h = layer((a, h))
            else:
                a, h = layer(a, h)
        return torch.cat([a, h], dim=-1)
