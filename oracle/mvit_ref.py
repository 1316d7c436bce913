"""CPU restatement (plain torch fp32/fp64) of the reference's MViT forward graph: the MViTv2 configuration family
(conv pooling, cls token, decomposed relative positions, residual pooling, DIM_MUL_IN_ATT) and the MViTv1 / ViT one
(learned absolute position embedding, dimension change after the Mlp, blocks without q or k/v pooling, mean pooling).

TEST INFRASTRUCTURE: the oracle the HIP engine's MViT path is checked against.  Pinned to the unmodified reference
by oracle/make_golden.py (golden fixtures tests/golden/mvit_*.json).  Each function cites the reference lines it
follows.  Driven by a flat ``state_dict`` with the reference's key names; autograd gives the reference backward.
"""
import math

import torch
import torch.nn.functional as F

from . import video_ref as _vr


def _store(x):
    return _vr._STORE(x)


# Storage-model policy of the RESIDUAL STREAM (tools/mvit_logits_bisect.py, tests/model_checks.py): None = the two residual
# sums of every block are stored like any other tensor; otherwise a callable (x [B, N, C], block index, which in {0, 1}) ->
# stored x, so a test can model an engine that keeps part of the stream (the class-token row, the last stage) in fp32.
RESID_POLICY = None


def _store_resid(x, prefix, which):
    if RESID_POLICY is None or _vr._STORE is _vr._ident:
        return _store(x)
    return RESID_POLICY(x, int(prefix.split(".")[1]), which)


def engine_resid_policy(cls_fp32=True, fp32_from_block=10 ** 6):
    """The residual-stream storage of slowfast_amd.mvit_engine (round 4): the class-token row of both residual sums of every
    block stays fp32 (a [B, C] side buffer); blocks >= ``fp32_from_block`` keep the whole stream in fp32 (the engine's
    SF_MVIT_RESID32=full option: MViTv2-S blocks 14-15)."""
    def policy(x, i, which):
        if i >= fp32_from_block:
            return x
        if cls_fp32:
            return torch.cat([x[:, :1], _store(x[:, 1:])], 1)
        return _store(x)
    return policy


def _linear(x, sd, prefix):
    return _store(F.linear(x, _store(sd[prefix + ".weight"]), sd.get(prefix + ".bias")))


def _ln(x, sd, prefix, eps=1e-6):
    """nn.LayerNorm(eps=1e-6) (video_model_builder.py:857-858 partial(nn.LayerNorm, eps=1e-6))."""
    return _store(F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps))


def attention_pool(tensor, weight, stride, thw, has_cls, norm=None, sd=None, pool_mode="conv"):
    """attention_pool (attention.py:13-45) with a depthwise Conv3d (kernel 3x3x3 / padding 1 per cfg POOL_KVQ_KERNEL)
    or a MaxPool3d for the skip path (weight None).  tensor: (B, heads, L, C) or (B, L, C)."""
    dim3 = tensor.ndim == 3
    if dim3:
        tensor = tensor.unsqueeze(1)
    if has_cls:
        cls_tok, tensor = tensor[:, :, :1, :], tensor[:, :, 1:, :]
    B, N, L, C = tensor.shape
    T, H, W = thw
    t = tensor.reshape(B * N, T, H, W, C).permute(0, 4, 1, 2, 3).contiguous()
    if pool_mode == "conv":
        k = weight.shape[2:]
        t = F.conv3d(t, _store(weight), None, tuple(stride), tuple(int(v // 2) for v in k), 1, C)
        t = _store(t)
    else:   # max: kernel = stride + 1 where stride > 1 (attention.py:430-432)
        k = [s + 1 if s > 1 else s for s in stride]
        t = F.max_pool3d(t, k, tuple(stride), [int(v // 2) for v in k])
    thw_new = [t.shape[2], t.shape[3], t.shape[4]]
    t = t.reshape(B, N, C, -1).transpose(2, 3)
    if has_cls:
        t = torch.cat((cls_tok, t), dim=2)
    if norm is not None:
        t = _ln(t, sd, norm)
    if dim3:
        t = t.squeeze(1)
    return t, thw_new


def rel_pos_index(q_size, k_size):
    """dist tables of cal_rel_pos_spatial / cal_rel_pos_temporal (attention.py:76-88, 123-130) as int64."""
    q_ratio = max(k_size / q_size, 1.0)
    k_ratio = max(q_size / k_size, 1.0)
    dist = torch.arange(q_size)[:, None] * q_ratio - torch.arange(k_size)[None, :] * k_ratio
    dist += (k_size - 1) * k_ratio
    return dist.long()


def get_rel_pos(rel_pos, d):
    """get_rel_pos (attention.py:48-61): a table whose row count differs from 2*max(q,k)-1 is resampled along its rows
    with F.interpolate(mode="linear").  (Happens when a pooled extent is odd: the constructor sizes the tables with
    size // stride, the pooling convs produce ceil(size / stride), e.g. MViTv2-L at 312^2: 39 -> 20, not 19.)"""
    ori_d = rel_pos.shape[0]
    if ori_d == d:
        return rel_pos
    new = F.interpolate(rel_pos.reshape(1, ori_d, -1).permute(0, 2, 1), size=d, mode="linear")
    return new.reshape(-1, d).permute(1, 0)


def _rel_pos_bias(attn, q, has_cls, q_shape, k_shape, rel_h, rel_w, rel_t):
    """cal_rel_pos_spatial + cal_rel_pos_temporal (attention.py:64-147); tables constructed with 2*max(q,k)-1 rows
    (attention.py:271-287) are used as they are, others go through get_rel_pos."""
    sp = 1 if has_cls else 0
    q_t, q_h, q_w = q_shape
    k_t, k_h, k_w = k_shape
    B, nh, qN, dim = q.shape
    r_q = q[:, :, sp:].reshape(B, nh, q_t, q_h, q_w, dim)
    out = attn[:, :, sp:, sp:].reshape(B, nh, q_t, q_h, q_w, k_t, k_h, k_w)
    if rel_h is not None:
        Rh = get_rel_pos(rel_h, 2 * max(q_h, k_h) - 1)[rel_pos_index(q_h, k_h)]
        Rw = get_rel_pos(rel_w, 2 * max(q_w, k_w) - 1)[rel_pos_index(q_w, k_w)]
        rel_h_q = torch.einsum("bythwc,hkc->bythwk", r_q, Rh)
        rel_w_q = torch.einsum("bythwc,wkc->bythwk", r_q, Rw)
        out = out + rel_h_q[:, :, :, :, :, None, :, None] + rel_w_q[:, :, :, :, :, None, None, :]
    if rel_t is not None:
        Rt = get_rel_pos(rel_t, 2 * max(q_t, k_t) - 1)[rel_pos_index(q_t, k_t)]
        rel_t_q = torch.einsum("bythwc,tkc->bythwk", r_q, Rt)
        out = out + rel_t_q[:, :, :, :, :, :, None, None]
    out = out.reshape(B, nh, q_t * q_h * q_w, k_t * k_h * k_w)
    if sp:
        top = attn[:, :, :1, :]
        left = attn[:, :, 1:, :1]
        out = torch.cat([top, torch.cat([left, out], dim=3)], dim=2)
    return out


def attention(x, sd, prefix, thw, heads, stride_q, stride_kv, has_cls=True, residual_pooling=True, pool_first=False):
    """MultiScaleAttention.forward (attention.py:293-392), mode "conv".  Fused qkv projection, or separate q / k / v
    Linears (SEPARATE_QKV, attention.py:188-191, 310-314), or POOL_FIRST (attention.py:296-301, 339-351: the block input
    is folded into heads and pooled BEFORE the q / k / v Linears; the pooling convs then have dim // heads channels).
    q (k / v) is left un-pooled and un-normed when the block has no pool_q (pool_k / pool_v) -- attention.py:199-203,
    236-262: kernel () when both kernel and stride are all ones (MViTv1 blocks outside POOL_Q_STRIDE)."""
    B, N, _ = x.shape
    if pool_first:
        q = k = v = x.reshape(B, N, heads, -1).permute(0, 2, 1, 3)
    elif prefix + ".qkv.weight" in sd:
        qkv = _linear(x, sd, prefix + ".qkv").reshape(B, N, 3, heads, -1).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
    else:
        q, k, v = (_linear(x, sd, prefix + "." + n).reshape(B, N, heads, -1).permute(0, 2, 1, 3) for n in "qkv")
    q_shape = k_shape = list(thw)
    if prefix + ".pool_q.weight" in sd:
        q, q_shape = attention_pool(q, sd[prefix + ".pool_q.weight"], stride_q, thw, has_cls, prefix + ".norm_q", sd)
    if prefix + ".pool_k.weight" in sd:
        k, k_shape = attention_pool(k, sd[prefix + ".pool_k.weight"], stride_kv, thw, has_cls, prefix + ".norm_k", sd)
        v, _ = attention_pool(v, sd[prefix + ".pool_v.weight"], stride_kv, thw, has_cls, prefix + ".norm_v", sd)
    if pool_first:
        q, k, v = (_linear(t.permute(0, 2, 1, 3).reshape(B, t.shape[2], -1), sd, prefix + "." + n)
                   .reshape(B, t.shape[2], heads, -1).permute(0, 2, 1, 3) for t, n in ((q, "q"), (k, "k"), (v, "v")))
    head_dim = q.shape[-1]
    scale = head_dim ** -0.5
    attn = (q * scale) @ k.transpose(-2, -1)
    attn = _store(attn)
    if prefix + ".rel_pos_h" in sd or prefix + ".rel_pos_t" in sd:
        attn = _rel_pos_bias(attn, q, has_cls, q_shape, k_shape, sd.get(prefix + ".rel_pos_h"), sd.get(prefix + ".rel_pos_w"),
                             sd.get(prefix + ".rel_pos_t"))
    attn = _store(attn.softmax(dim=-1))
    o = attn @ v
    if residual_pooling:
        if has_cls:
            o = torch.cat([o[:, :, :1, :], o[:, :, 1:, :] + q[:, :, 1:, :]], dim=2)
        else:
            o = o + q
    o = _store(o)
    o = o.transpose(1, 2).reshape(B, -1, heads * head_dim)
    return _linear(o, sd, prefix + ".proj"), q_shape


def block(x, sd, prefix, thw, heads, stride_q, stride_kv, has_cls=True, drop=None, residual_pooling=True,
          dim_mul_in_att=True, pool_first=False):
    """MultiScaleBlock.forward (attention.py:491-514).  The dimension change (``proj``) acts on the normed block input
    with DIM_MUL_IN_ATT (MViTv2) and on the normed Mlp input without it (MViTv1, attention.py:507-508).
    drop = (s1, s2): per-sample scales mask/keep_prob of the two drop_path() calls (common.py:46-59), None = off."""
    x_norm = _ln(x, sd, prefix + ".norm1")
    x_block, thw_new = attention(x_norm, sd, prefix + ".attn", thw, heads, stride_q, stride_kv, has_cls, residual_pooling,
                                 pool_first)
    has_proj = prefix + ".proj.weight" in sd
    if has_proj and dim_mul_in_att:
        x = _linear(x_norm, sd, prefix + ".proj")
    if math.prod(stride_q) > 1:
        x_res, _ = attention_pool(x, None, stride_q, thw, has_cls, pool_mode="max")
    else:
        x_res = x
    if drop is not None:
        x_block = _store(x_block) * drop[0].view(-1, 1, 1)
    x = _store_resid(x_res + x_block, prefix, 0)
    x_norm = _ln(x, sd, prefix + ".norm2")
    h = _linear(x_norm, sd, prefix + ".mlp.fc1")
    h = _store(F.gelu(h))
    x_mlp = F.linear(h, _store(sd[prefix + ".mlp.fc2.weight"]), sd[prefix + ".mlp.fc2.bias"])
    if has_proj and not dim_mul_in_att:
        x = _linear(x_norm, sd, prefix + ".proj")
    if drop is not None:
        x_mlp = _store(x_mlp) * drop[1].view(-1, 1, 1)
    return _store_resid(x + x_mlp, prefix, 1), thw_new


def _mlp(x, sd, prefix):
    """Mlp.forward (common.py:25-34), GELU, no dropout."""
    h = _store(F.gelu(_linear(x, sd, prefix + ".fc1")))
    return F.linear(h, _store(sd[prefix + ".fc2.weight"]), sd[prefix + ".fc2.bias"])


def rev_backbone(x, sd, cfg, thw, drop=None):
    """ReversibleMViT.forward (reversible_mvit.py:141-174) in its eval / "vanilla" form (:128-139), which computes the same
    values as the RevBackProp path (:177-263; that one only re-derives activations in backward instead of storing them).
    ReversibleBlock (:491-526): Y1 = X1 + F(X2), Y2 = X2 + G(Y1), F = attention(norm(.)), G = mlp(norm(.)) (:593-672).
    StageTransitionBlock (:350-409) for MVIT.REV.PRE_Q_FUSION "avg" and RES_PATH "conv": x = mean of the two streams;
    residual = [res_proj] -> pool_q conv + norm_q (the attention's own q pooling parameters) -> [res_proj if POOL_FIRST];
    x = residual + F(x); x = x + G(x); drop_path(x).  drop[i] = per-sample scale of layer i's stochastic depth (both
    branches of a ReversibleBlock draw the same mask: the generator is re-seeded with the "droppath" seed, :500-519)."""
    assert cfg.MVIT.REV.PRE_Q_FUSION == "avg" and cfg.MVIT.REV.RES_PATH == "conv" and not cfg.MVIT.CLS_EMBED_ON
    plan = mvit_plan(cfg)
    pool_first, res_pool = cfg.MVIT.POOL_FIRST, cfg.MVIT.RESIDUAL_POOLING
    a = h = None
    for i, (heads, sq, skv) in enumerate(plan):
        pre = f"rev_backbone.layers.{i}"
        s = None if drop is None else drop[i].view(-1, 1, 1)

        def f_att(t):
            return attention(_ln(t, sd, pre + ".F.norm"), sd, pre + ".F.attn", thw, heads, sq, skv, False, res_pool,
                             pool_first)

        def g_mlp(t):
            return _mlp(_ln(t, sd, pre + ".G.norm"), sd, pre + ".G.mlp")

        if i in cfg.MVIT.REV.BUFFER_LAYERS:
            if a is not None:
                x = _store((a + h) * 0.5)                     # TwoStreamFusion("avg") (common.py:99-100)
            x_res = x
            has_proj = pre + ".res_proj.weight" in sd
            if has_proj and not pool_first:
                x_res = _linear(x_res, sd, pre + ".res_proj")
            B, L, C = x_res.shape
            t = x_res.reshape(B, L, heads, C // heads).permute(0, 2, 1, 3)
            t, _ = attention_pool(t, sd[pre + ".F.attn.pool_q.weight"], sq, thw, False, pre + ".F.attn.norm_q", sd)
            x_res = t.permute(0, 2, 1, 3).reshape(B, t.shape[2], C)
            if has_proj and pool_first:
                x_res = _linear(x_res, sd, pre + ".res_proj")
            xa, thw_new = f_att(x)
            x = _store(x_res + xa)
            x = _store(x + g_mlp(x))
            if s is not None:
                x = _store(x * s)
            thw = thw_new
            a = h = None
        else:
            if a is None:
                a = h = x                                   # torch.cat([x, x], -1) then chunk (:157, :135)
            fa, _ = f_att(h)
            y1 = _store(a + (fa if s is None else _store(fa) * s))
            g = g_mlp(y1)
            y2 = _store(h + (g if s is None else _store(g) * s))
            a, h = y1, y2
    return torch.cat([a, h], dim=-1) if a is not None else x


def mvit_plan(cfg):
    """Per-block (heads, stride_q, stride_kv) from the cfg, as MViT.__init__ derives them
    (video_model_builder.py:906-1004)."""
    depth = cfg.MVIT.DEPTH
    head_mul = [1.0] * (depth + 1)
    for i, m in cfg.MVIT.HEAD_MUL:
        head_mul[i] = m
    stride_q = [[1, 1, 1] for _ in range(depth)]
    for e in cfg.MVIT.POOL_Q_STRIDE:
        stride_q[e[0]] = list(e[1:])
    stride_kv = [[1, 1, 1] for _ in range(depth)]
    if cfg.MVIT.POOL_KV_STRIDE_ADAPTIVE is not None:
        _kv = list(cfg.MVIT.POOL_KV_STRIDE_ADAPTIVE)
        for i in range(depth):
            if math.prod(stride_q[i]) > 1:
                _kv = [max(_kv[d] // stride_q[i][d], 1) for d in range(3)]
            stride_kv[i] = list(_kv)
    else:
        for e in cfg.MVIT.POOL_KV_STRIDE:
            stride_kv[e[0]] = list(e[1:])
    heads, plan = cfg.MVIT.NUM_HEADS, []
    for i in range(depth):
        heads = int(round(heads * head_mul[i]))
        plan.append((heads, stride_q[i], stride_kv[i]))
    return plan


def mvit_forward(sd, cfg, inputs, training=True, drop=None, bboxes=None):
    """MViT.forward (video_model_builder.py:1166-1244) with or without the cls token (CLS_EMBED_ON), learned absolute
    position embedding (joint or SEP_POS_EMBED) or none, no dropout, + TransformerBasicHead.forward
    (head_helper.py:538-563).  drop = per-block
    (s1, s2) stochastic-depth scales (the sampled masks of drop_path(), common.py:46-59, divided by keep_prob) or None."""
    x = inputs[0]
    w = sd["patch_embed.proj.weight"]
    stride, pad = tuple(cfg.MVIT.PATCH_STRIDE), tuple(cfg.MVIT.PATCH_PADDING)
    x = _store(F.conv3d(_store(x), _store(w), sd["patch_embed.proj.bias"], stride, pad))
    B, C, T, H, W = x.shape
    x = x.flatten(2).transpose(1, 2)
    has_cls = bool(cfg.MVIT.CLS_EMBED_ON)
    if has_cls:
        x = torch.cat((sd["cls_token"].expand(B, -1, -1), x), dim=1)
    if cfg.MVIT.USE_ABS_POS:            # video_model_builder.py:1189-1203 (same clip size as constructed: no interpolation)
        if cfg.MVIT.SEP_POS_EMBED:
            pos = sd["pos_embed_spatial"].repeat(1, T, 1) + torch.repeat_interleave(sd["pos_embed_temporal"], H * W, dim=1)
            if has_cls:
                pos = torch.cat([sd["pos_embed_class"], pos], 1)
        else:
            pos = sd["pos_embed"]
        x = x + pos
    x = _store(x)
    thw = [T, H, W]
    if cfg.MVIT.REV.ENABLE:             # MViT._forward_reversible (video_model_builder.py:1141-1164), RESPATH_FUSE "concat"
        assert cfg.MVIT.REV.RESPATH_FUSE == "concat"
        x = rev_backbone(x, sd, cfg, thw, drop)
        x = _ln(x.mean(1), sd, "norm") if cfg.MVIT.USE_MEAN_POOLING else _ln(x, sd, "norm").mean(1)
        z = F.linear(x, sd["head.projection.weight"], sd["head.projection.bias"])
        if not training and cfg.MODEL.HEAD_ACT == "softmax":
            z = F.softmax(z, dim=1)
        return z
    for i, (heads, sq, skv) in enumerate(mvit_plan(cfg)):
        x, thw = block(x, sd, f"blocks.{i}", thw, heads, sq, skv, has_cls=has_cls, drop=None if drop is None else drop[i],
                       residual_pooling=cfg.MVIT.RESIDUAL_POOLING, dim_mul_in_att=cfg.MVIT.DIM_MUL_IN_ATT,
                       pool_first=cfg.MVIT.POOL_FIRST)
    if cfg.DETECTION.ENABLE:            # video_model_builder.py:1218-1226: norm, drop cls, tokens -> (B, C, T, H, W), RoI head
        x = _ln(x, sd, "norm")
        x = x[:, 1:] if has_cls else x
        x = x.transpose(1, 2).reshape(x.shape[0], x.shape[2], thw[0], thw[1], thw[2])
        return _vr.roi_head([x], sd, cfg, bboxes, training)
    if cfg.MVIT.USE_MEAN_POOLING:       # video_model_builder.py:1228-1232: mean over the patch tokens, then norm
        x = _ln(x[:, 1:].mean(1) if has_cls else x.mean(1), sd, "norm")
    elif has_cls:
        x = _ln(x[:, 0], sd, "norm")
    else:                               # video_model_builder.py:1236-1238: norm, then mean over all tokens
        x = _ln(x, sd, "norm").mean(1)
    z = F.linear(x, sd["head.projection.weight"], sd["head.projection.bias"])
    if not training and cfg.MODEL.HEAD_ACT == "softmax":
        z = F.softmax(z, dim=1)
    return z


def randomize_state(shapes, seed, dtype=torch.float32):
    """Deterministic generic parameters keyed by the reference's MViT state_dict names."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in shapes.items():
        if name.endswith("weight") and len(shape) == 1:          # LayerNorm gamma
            sd[name] = (torch.rand(shape, generator=g) + 0.5).to(dtype)
        elif name.endswith("bias"):
            sd[name] = (torch.randn(shape, generator=g) * 0.1).to(dtype)
        elif "rel_pos" in name:
            sd[name] = (torch.randn(shape, generator=g) * 0.2).to(dtype)
        elif name.startswith("pos_embed"):
            sd[name] = (torch.randn(shape, generator=g) * 0.5).to(dtype)
        elif name == "cls_token":
            sd[name] = (torch.randn(shape, generator=g) * 0.5).to(dtype)
        elif len(shape) == 5:                                     # conv weights: fan-in scaled
            fan_in = shape[1] * shape[2] * shape[3] * shape[4]
            sd[name] = (torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5).to(dtype)
        elif len(shape) == 2:                                     # Linear weights
            sd[name] = (torch.randn(shape, generator=g) * (1.0 / shape[1]) ** 0.5).to(dtype)
        else:
            sd[name] = (torch.randn(shape, generator=g) * 0.1).to(dtype)
    return sd


def loss_and_grads(sd, cfg, inputs, labels, dtype=torch.float32, drop=None, bboxes=None, device=None,
                   autocast_dtype=None, loss_scale=1.0):
    """fp32 oracle by default; ``device`` / ``autocast_dtype`` / ``loss_scale``: the same graph under torch.autocast with
    a fixed loss scale, as the reference trains with TRAIN.MIXED_PRECISION (see video_ref.loss_and_grads)."""
    import contextlib
    dev = torch.device(device or "cpu")
    params = {k: v.detach().to(dev, dtype).clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    ctx = torch.autocast(dev.type, dtype=autocast_dtype) if autocast_dtype is not None else contextlib.nullcontext()
    if drop is not None and dev.type != "cpu":
        drop = [tuple(s.to(dev) for s in d) if isinstance(d, (tuple, list)) else d.to(dev) for d in drop]
    with ctx:
        logits = mvit_forward(params, cfg, [x.to(dev, dtype) for x in inputs], training=True, drop=drop,
                              bboxes=bboxes.to(dev) if bboxes is not None else None)
    labels = labels.to(dev)
    # detection head: BCE on the activated outputs (losses.py:61-69 "bce")
    loss = F.binary_cross_entropy(logits.float(), labels.float()) if bboxes is not None else F.cross_entropy(logits.float(), labels)
    (loss * loss_scale if loss_scale != 1.0 else loss).backward()
    grads = {k: (v.grad.float() / loss_scale).cpu() if (loss_scale != 1.0 or dev.type != "cpu") else v.grad
             for k, v in params.items() if v.grad is not None}
    return logits.detach().float().cpu(), loss.detach().float().cpu(), grads, {}
