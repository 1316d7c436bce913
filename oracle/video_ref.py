"""CPU restatement (plain torch fp32/fp64) of the reference's ResNet / SlowFast forward graph.

TEST INFRASTRUCTURE: the oracle the HIP engine is checked against.  It is pinned to the real
reference by oracle/make_golden.py (run in the build container, where /root/reference is importable):
logits, loss and every parameter gradient agree with the unmodified reference model to fp32 round-off,
and the values are committed under tests/golden/.  Each function cites the reference lines it follows.

The graph is driven by a flat ``state_dict`` (reference key names) and the cfg; every op is the torch
functional the reference's nn.Module would dispatch to, so autograd gives the reference backward.
"""
import math

import torch
import torch.nn.functional as F


STORAGE_DTYPE = torch.float16     # the storage model's 16-bit type; a bf16 test process sets torch.bfloat16


class _RoundFp16(torch.autograd.Function):
    """y = fp16(x) in forward AND fp16(dy) in backward: emulates an fp16 HBM round trip of an activation and of
    its gradient.  Used only by the "fp16 storage model" of the oracle (``video_forward(..., store=round_fp16)``),
    which tells apart rounding noise of ANY fp16-storage implementation from real kernel defects."""

    @staticmethod
    def forward(ctx, x):
        return x.to(STORAGE_DTYPE).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(STORAGE_DTYPE).to(g.dtype)


def round_fp16(x):
    return _RoundFp16.apply(x)


def _ident(x):
    return x


_STORE = _ident   # identity = the reference's fp32 graph; round_fp16 = fp16 storage model
# Storage-model policy of the dot-product Nonlocal affinity theta^T phi / P and of its gradient: True = one 16-bit tensor (what
# slowfast_amd.nonlocal_block stores), False = kept exact (a 16-bit hi + lo pair; tools/nl_storage_probe.py measures what that buys)
NL_AFFINITY_16BIT = True
NL_EXACT = frozenset()      # probe only: names among theta / phi / g / y / out (conv_out's result) kept exact by the storage model


class fp16_storage_model:
    """Context manager: every tensor that an fp16-storage implementation would round (conv operands and outputs,
    block outputs, their gradients) is rounded to fp16 at that point; arithmetic stays fp32.  This is NOT the
    oracle's reference mode (default: exact fp32 restatement of the reference); it is the yardstick that says how
    far from fp32 a correct fp16-storage / fp32-accumulate engine is expected to land."""

    def __enter__(self):
        global _STORE
        self._old, _STORE = _STORE, round_fp16

    def __exit__(self, *exc):
        global _STORE
        _STORE = self._old


class _Recorder(dict):
    """Per-module mask table being FILLED by a pass (recording_masks): the ReLU masks / max-pool routes the pass itself used."""


class _ReluFixedMask(torch.autograd.Function):
    """relu(x) in forward; the BACKWARD mask is supplied by the caller instead of being x > 0.  Parity tests of fused
    blocks take the masks the engine under test actually used: an element whose pre-activation lies within fp16 round-off
    of zero may legitimately land on either side in two correct fp16 realisations, and its O(1) effect on the gradients
    would otherwise drown the 2e-3 comparison of everything else (tests/block_checks.py)."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return x.clamp_min(0)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask.to(g.dtype), None


def _relu(x, masks=None, key=None):
    if isinstance(masks, _Recorder):
        masks[key] = (x > 0).detach()
        return F.relu(x)
    if masks is not None and masks.get(key) is not None:
        return _ReluFixedMask.apply(x, masks[key])
    return F.relu(x)


class _MaxPoolRouted(torch.autograd.Function):
    """max_pool3d(x) in forward (the true maximum); the BACKWARD sends each output's gradient to the input element the
    caller names (``index``: int64 (N, C, To, Ho, Wo), flat t*H*W + h*W + w) instead of to the arg-max.  Same purpose as
    _ReluFixedMask: a window whose two largest entries differ by less than fp16 round-off may be routed to either by a correct
    fp16 implementation, which moves an O(1) gradient contribution between two positions."""

    @staticmethod
    def forward(ctx, x, kernel, stride, padding, index):
        ctx.save_for_backward(index)
        ctx.in_shape = tuple(x.shape)
        return F.max_pool3d(x, kernel, stride, padding)

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        N, C, T, H, W = ctx.in_shape
        dx = torch.zeros((N, C, T * H * W), dtype=g.dtype, device=g.device)
        dx.scatter_add_(2, index.reshape(N, C, -1), g.reshape(N, C, -1))
        return dx.reshape(ctx.in_shape), None, None, None, None


class _AmaxRouted(torch.autograd.Function):
    """x.amax(-1) in forward; the backward sends the gradient to the element ``index`` names instead of to the arg-max (the
    RoI head's max over the ROIAlign bins, head_helper.py:97 MaxPool2d(resolution); see _MaxPoolRouted)."""

    @staticmethod
    def forward(ctx, x, index):
        ctx.save_for_backward(index)
        ctx.n = x.shape[-1]
        return x.amax(-1)

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        dx = torch.zeros(tuple(g.shape) + (ctx.n,), dtype=g.dtype, device=g.device)
        dx.scatter_(-1, index.unsqueeze(-1), g.unsqueeze(-1))
        return dx, None


def _max_pool(x, kernel, stride, padding, masks=None, key="pool_route"):
    if isinstance(masks, _Recorder):
        y, idx = F.max_pool3d(x, tuple(kernel), tuple(stride), tuple(padding), return_indices=True)
        masks[key] = idx.detach()
        return y
    if masks is not None and masks.get(key) is not None:
        return _MaxPoolRouted.apply(x, tuple(kernel), tuple(stride), tuple(padding), masks[key])
    return F.max_pool3d(x, tuple(kernel), tuple(stride), tuple(padding))


# Whole-model hand-over (tests only): {module prefix: {key: mask / route}} of the engine under test, looked up by every
# block below when its own ``masks`` argument is None.  See tests/model_checks.py:engine_masks.
_HANDED = None


class handed_masks:
    """Context manager: the BACKWARD of every ReLU / max-pool whose module prefix appears in ``table`` runs through the mask /
    routes given there (the forward stays the exact fp32 reference)."""

    def __init__(self, table):
        self.table = table

    def __enter__(self):
        global _HANDED
        self._old, _HANDED = _HANDED, self.table

    def __exit__(self, *exc):
        global _HANDED
        _HANDED = self._old


class recording_masks:
    """Context manager: the pass inside records the ReLU masks / max-pool routes IT uses into ``self.table`` ({module prefix:
    {key: mask / route}}, the shape handed_masks takes) -- two oracle passes (fp32 / a storage model) can then be compared
    under identical masks without an engine run (tools/nl_storage_probe.py)."""

    def __enter__(self):
        global _HANDED
        self.table = _RecorderTable()
        self._old, _HANDED = _HANDED, self.table
        return self

    def __exit__(self, *exc):
        global _HANDED
        _HANDED = self._old
        self.table = {k: dict(v) for k, v in self.table.items()}


class _RecorderTable(dict):
    def get(self, prefix, default=None):
        if prefix not in self:
            self[prefix] = _Recorder()
        return self[prefix]


def _handed(prefix, masks):
    if masks is not None:
        return masks
    return _HANDED.get(prefix) if _HANDED is not None else None


def _conv(x, w, *args):
    return _STORE(F.conv3d(x, _STORE(w), *args))   # args = (bias, stride, padding, dilation)


def _bn(x, sd, prefix, training, stats_out, momentum=0.1, eps=1e-5):
    """nn.BatchNorm3d (slowfast/models/batchnorm_helper.py:24-25 -> torch BatchNorm3d; eps/momentum as passed at
    every construction site, e.g. resnet_helper.py:340-342)."""
    if prefix + ".split_bn.running_mean" in sd:
        return _sub_bn(x, sd, prefix, training, stats_out, momentum, eps)
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if training:
        rm, rv = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"], True, momentum, eps)
        if stats_out is not None:
            stats_out[prefix + ".running_mean"], stats_out[prefix + ".running_var"] = rm, rv
        return y
    return F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"], False, momentum, eps)


def _sub_bn(x, sd, prefix, training, stats_out, momentum, eps):
    """SubBatchNorm3d.forward (slowfast/models/batchnorm_helper.py:99-112): training = BatchNorm (no affine) over the
    batch viewed as (n / S, c * S, t, h, w), i.e. statistics per split of samples n % S; eval = ``bn`` with the
    aggregated running statistics; then the shared affine pair."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if training:
        rm, rv = sd[prefix + ".split_bn.running_mean"].clone(), sd[prefix + ".split_bn.running_var"].clone()
        S = rm.numel() // w.numel()
        n, c, t, h, ww = x.shape
        y = F.batch_norm(x.reshape(n // S, c * S, t, h, ww), rm, rv, None, None, True, momentum, eps).reshape(n, c, t, h, ww)
        if stats_out is not None:
            stats_out[prefix + ".split_bn.running_mean"], stats_out[prefix + ".split_bn.running_var"] = rm, rv
    else:
        y = F.batch_norm(x, sd[prefix + ".bn.running_mean"], sd[prefix + ".bn.running_var"], None, None, False, momentum, eps)
    return y * w.view(-1, 1, 1, 1) + b.view(-1, 1, 1, 1)


def stem(x, sd, prefix, training, stats_out, masks=None):
    """ResNetBasicStem.forward: conv -> bn -> relu -> MaxPool3d([1,3,3],[1,2,2],[0,1,1]) (stem_helper.py:182-201).
    ``masks`` (tests only, see _ReluFixedMask): {"relu": 0/1 tensor, "pool_index": int64 (N, C, T, Ho, Wo) flat h*W + w of the
    element each pooled output is routed to} -- the backward runs through the routes the engine under test took."""
    masks = _handed(prefix, masks)
    w = sd[prefix + ".conv.weight"]
    kt = w.shape[2]
    x = _conv(_STORE(x), w, None, (1, 2, 2), (kt // 2, 3, 3))
    x = _STORE(_relu(_bn(x, sd, prefix + ".bn", training, stats_out), masks, "relu"))
    if masks is not None and masks.get("pool_index") is not None:
        N, C, T, H, W = x.shape
        idx = masks["pool_index"]
        return x.reshape(N, C, T, H * W).gather(3, idx.reshape(N, C, T, -1)).reshape(idx.shape)
    return _max_pool(x, (1, 3, 3), (1, 2, 2), (0, 1, 1), masks)      # "pool_route": true maximum forward, routed backward


def fuse(xs, xf, sd, prefix, alpha, training, stats_out, masks=None):
    """FuseFastToSlow.forward (video_model_builder.py:162-169): time-strided conv on Fast, BN, ReLU, concat."""
    masks = _handed(prefix, masks)
    w = sd[prefix + ".conv_f2s.weight"]
    k = w.shape[2]
    f = _conv(xf, w, None, (alpha, 1, 1), (k // 2, 0, 0))
    f = _STORE(_relu(_bn(f, sd, prefix + ".bn", training, stats_out), masks, "relu"))
    return torch.cat([xs, f], 1)


def res_block(x, sd, prefix, stride, dilation, stride_1x1, training, stats_out, masks=None):
    """ResBlock.forward (resnet_helper.py:512-521) around BottleneckTransform.forward (:377-392) or, when the block has
    no ``c`` convolution, BasicTransform.forward (:105-115: Tx3x3 stride s -> BN -> ReLU -> 1x3x3 dilated -> BN).
    ``masks`` ({"a", "b", "out"} -> 0/1 tensors): backward masks of the three ReLUs (see _ReluFixedMask; tests only)."""
    masks = _handed(prefix, masks)
    s_a, s_b = (stride, 1) if stride_1x1 else (1, stride)
    b2 = prefix + ".branch2"
    wa = sd[b2 + ".a.weight"]
    kt = wa.shape[2]
    if b2 + ".c.weight" not in sd:
        y = _conv(x, wa, None, (1, stride, stride), (kt // 2, 1, 1))
        y = _STORE(_relu(_bn(y, sd, b2 + ".a_bn", training, stats_out), masks, "a"))
        y = _conv(y, sd[b2 + ".b.weight"], None, (1, 1, 1), (0, dilation, dilation), (1, dilation, dilation))
        y = _bn(y, sd, b2 + ".b_bn", training, stats_out)
    else:
        y = _conv(x, wa, None, (1, s_a, s_a), (kt // 2, 0, 0))
        y = _STORE(_relu(_bn(y, sd, b2 + ".a_bn", training, stats_out), masks, "a"))
        y = _conv(y, sd[b2 + ".b.weight"], None, (1, s_b, s_b), (0, dilation, dilation), (1, dilation, dilation))
        y = _STORE(_relu(_bn(y, sd, b2 + ".b_bn", training, stats_out), masks, "b"))
        y = _conv(y, sd[b2 + ".c.weight"])
        y = _bn(y, sd, b2 + ".c_bn", training, stats_out)
    if prefix + ".branch1.weight" in sd:
        sc = _conv(x, sd[prefix + ".branch1.weight"], None, (1, stride, stride))
        sc = _bn(sc, sd, prefix + ".branch1_bn", training, stats_out)
    else:
        sc = x
    return _STORE(_relu(sc + y, masks, "out"))


def nonlocal_block(x, sd, prefix, pool_size, instantiation, training, stats_out):
    """Nonlocal.forward (nonlocal_helper.py:103-144): theta/phi/g 1x1x1 convs (+bias), phi and g on the max-pooled
    input, affinity theta^T phi normalised by softmax(./sqrt(C)) or by 1/N ("dot_product"), out conv + BN, residual."""
    N, C, T, H, W = x.shape

    def st(name):       # storage-model probe: tensors named in NL_EXACT stay exact (tools/nl_storage_probe.py)
        return _ident if name in NL_EXACT else _STORE

    def conv(name, inp):
        return st(name)(F.conv3d(inp, _STORE(sd[prefix + ".conv_" + name + ".weight"]), sd[prefix + ".conv_" + name + ".bias"]))
    theta = conv("theta", x)
    xp = _max_pool(x, pool_size, pool_size, (0, 0, 0), _handed(prefix, None)) if any(s > 1 for s in pool_size) else x
    phi = conv("phi", xp)
    g = conv("g", xp)
    ci = theta.shape[1]
    theta, phi, g = theta.view(N, ci, -1), phi.view(N, ci, -1), g.view(N, ci, -1)
    store_a = _STORE if (NL_AFFINITY_16BIT or instantiation != "dot_product") else _ident
    a = store_a(torch.einsum("nct,ncp->ntp", theta, phi))
    if instantiation == "softmax":
        a = F.softmax(a * (ci ** -0.5), dim=2)
    elif instantiation == "dot_product":
        a = a / a.shape[2]
    else:
        raise NotImplementedError(instantiation)
    a = store_a(a)
    y = st("y")(torch.einsum("ntg,ncg->nct", a, g)).view(N, ci, T, H, W)
    p = conv("out", y)
    p = _bn(p, sd, prefix + ".bn", training, stats_out)
    return _STORE(x + p)


def res_stage(xs, sd, name, strides, dilations, stride_1x1, training, stats_out, nonlocal_pool=None,
              instantiation="dot_product", nonlocal_group=None):
    """ResStage.forward (resnet_helper.py:697-726); Nonlocal blocks after the listed blocks, with NONLOCAL.GROUP > 1
    folding T into the batch around the block (:706-723)."""
    out = []
    for p, x in enumerate(xs):
        i = 0
        while f"{name}.pathway{p}_res{i}.branch2.a.weight" in sd:
            x = res_block(x, sd, f"{name}.pathway{p}_res{i}", strides[p] if i == 0 else 1, dilations[p], stride_1x1,
                          training, stats_out)
            nl = f"{name}.pathway{p}_nonlocal{i}"
            if nl + ".conv_theta.weight" in sd:
                g = nonlocal_group[p] if nonlocal_group is not None else 1
                b, c, t, h, w = x.shape
                if g > 1:
                    x = x.permute(0, 2, 1, 3, 4).reshape(b * g, t // g, c, h, w).permute(0, 2, 1, 3, 4)
                x = nonlocal_block(x, sd, nl, nonlocal_pool[p], instantiation, training, stats_out)
                if g > 1:
                    x = x.permute(0, 2, 1, 3, 4).reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)
            i += 1
        out.append(x)
    return out


# ------------------------------------------------------------------------------------------------
# X3D
def x3d_stem(x, sd, prefix, training, stats_out):
    """X3DStem.forward (stem_helper.py:279-285): conv_xy (1,3,3)/(1,2,2) -> depthwise (5,1,1) -> BN -> ReLU."""
    w_xy, w_t = sd[prefix + ".conv_xy.weight"], sd[prefix + ".conv.weight"]
    x = _conv(_STORE(x), w_xy, None, (1, 2, 2), (0, 1, 1))
    x = _STORE(F.conv3d(x, _STORE(w_t), None, 1, (w_t.shape[2] // 2, 0, 0), 1, w_t.shape[0]))
    return _STORE(_relu(_bn(x, sd, prefix + ".bn", training, stats_out), _handed(prefix, None), "relu"))


def x3d_block(x, sd, prefix, stride, training, stats_out):
    """ResBlock.forward (resnet_helper.py:512-521) around X3DTransform.forward (:253-256): 1x1x1 -> BN -> ReLU ->
    depthwise 3x3x3 (stride) -> BN -> [SE] -> Swish -> 1x1x1 -> BN; SE = operators.py:53-59."""
    masks = _handed(prefix, None)
    b2 = prefix + ".branch2"
    y = _conv(x, sd[b2 + ".a.weight"])
    y = _STORE(_relu(_bn(y, sd, b2 + ".a_bn", training, stats_out), masks, "a"))
    wb = sd[b2 + ".b.weight"]
    y = _STORE(F.conv3d(y, _STORE(wb), None, (1, stride, stride), (wb.shape[2] // 2, 1, 1), 1, wb.shape[0]))
    y = _bn(y, sd, b2 + ".b_bn", training, stats_out)
    if b2 + ".se.fc1.weight" in sd:
        g = y.mean((2, 3, 4), keepdim=True)
        g = _relu(F.conv3d(g, sd[b2 + ".se.fc1.weight"], sd[b2 + ".se.fc1.bias"]), masks, "se")
        g = torch.sigmoid(F.conv3d(g, sd[b2 + ".se.fc2.weight"], sd[b2 + ".se.fc2.bias"]))
        y = y * g
    y = _STORE(y * torch.sigmoid(y))                   # Swish
    y = _conv(y, sd[b2 + ".c.weight"])
    y = _bn(y, sd, b2 + ".c_bn", training, stats_out)
    if prefix + ".branch1.weight" in sd:
        sc = _conv(x, sd[prefix + ".branch1.weight"], None, (1, stride, stride))
        sc = _bn(sc, sd, prefix + ".branch1_bn", training, stats_out)
    else:
        sc = x
    return _STORE(_relu(sc + y, masks, "out"))


def x3d_forward(sd, cfg, inputs, training=True, stats_out=None):
    """X3D.forward (video_model_builder.py:799-802) + X3DHead.forward (head_helper.py:461-488); with X3D.BN_LIN5 the
    head has a BatchNorm between lin_5 and its ReLU (head_helper.py:440-443, 470-471)."""
    x = x3d_stem(inputs[0], sd, "s1.pathway0_stem", training, stats_out)
    for s in range(2, 6):
        i = 0
        while f"s{s}.pathway0_res{i}.branch2.a.weight" in sd:
            x = x3d_block(x, sd, f"s{s}.pathway0_res{i}", 2 if i == 0 else 1, training, stats_out)
            i += 1
    x = _conv(x, sd["head.conv_5.weight"])
    x = _STORE(_relu(_bn(x, sd, "head.conv_5_bn", training, stats_out), _handed("head", None), "conv_5"))
    # nn.AvgPool3d([NUM_FRAMES, ceil(crop/32), ceil(crop/32)], stride=1) (video_model_builder.py:783-797): the whole
    # extent at the training crop, a sliding window (fully-convolutional inference) at a larger test crop
    spat = -(-cfg.DATA.TRAIN_CROP_SIZE // 32)
    x = F.avg_pool3d(x, (cfg.DATA.NUM_FRAMES, spat, spat), 1)
    x = F.conv3d(x, sd["head.lin_5.weight"])
    if "head.lin_5_bn.weight" in sd:
        x = _bn(x, sd, "head.lin_5_bn", training, stats_out)
    x = _relu(x, _handed("head", None), "lin_5")
    z = F.linear(x.permute(0, 2, 3, 4, 1), sd["head.projection.weight"], sd["head.projection.bias"])
    if not training:
        z = F.softmax(z, dim=4).mean([1, 2, 3])
    return z.reshape(z.shape[0], -1)


_POOL1_T = {"2d": 1, "c2d": 2, "slow_c2d": 1, "i3d": 2, "slow_i3d": 1, "slow": 1, "slowfast": 1}


def roi_align(feat, rois, out_size, spatial_scale, sampling_ratio=0, aligned=True):
    """ROIAlign on a (B, C, H, W) map for rois (R, 5) = [batch index, x1, y1, x2, y2] -> (R, C, out_h, out_w).

    PARITY UNPINNED: the reference takes this op from detectron2.layers.ROIAlign (-> torchvision.ops.roi_align), which is
    not vendored under /root/reference and not installed here.  This restates the published algorithm (Mask R-CNN
    ROIAlign as implemented in torchvision/csrc/ops/cpu/roi_align_kernel.cpp): continuous coordinates scaled by
    spatial_scale, shifted by -0.5 when `aligned`; legacy (aligned=False) clamps the RoI size to >= 1; every output bin
    averages a grid of ceil(roi_size / out_size) (or sampling_ratio) bilinear samples per axis; samples outside
    [-1, size] contribute 0, coordinates are clamped into the map.  Call sites in the reference:
    head_helper.py:88-94, 126-127."""
    R = rois.shape[0]
    B, C, H, W = feat.shape
    oh, ow = out_size
    out = []
    off = 0.5 if aligned else 0.0
    for r in range(R):
        b = int(rois[r, 0])
        x1, y1, x2, y2 = [float(v) * spatial_scale - off for v in rois[r, 1:5]]
        rw, rh = x2 - x1, y2 - y1
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bh, bw = rh / oh, rw / ow
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / oh))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / ow))
        count = max(gh * gw, 1)
        ys = y1 + (torch.arange(oh, dtype=feat.dtype, device=feat.device)[:, None] * bh + (torch.arange(gh, dtype=feat.dtype, device=feat.device)[None, :] + 0.5) * bh / max(gh, 1))
        xs = x1 + (torch.arange(ow, dtype=feat.dtype, device=feat.device)[:, None] * bw + (torch.arange(gw, dtype=feat.dtype, device=feat.device)[None, :] + 0.5) * bw / max(gw, 1))
        ys, xs = ys.reshape(-1), xs.reshape(-1)                  # (oh*gh), (ow*gw)
        vy = ((ys >= -1.0) & (ys <= H)).to(feat.dtype)
        vx = ((xs >= -1.0) & (xs <= W)).to(feat.dtype)
        yc, xc = ys.clamp(min=0.0), xs.clamp(min=0.0)
        y0, x0 = yc.floor().long(), xc.floor().long()
        top, left = y0 >= H - 1, x0 >= W - 1
        y0, x0 = torch.where(top, torch.full_like(y0, H - 1), y0), torch.where(left, torch.full_like(x0, W - 1), x0)
        y1i, x1i = torch.where(top, y0, y0 + 1), torch.where(left, x0, x0 + 1)
        ly = torch.where(top, torch.zeros_like(yc), yc - y0.to(feat.dtype))
        lx = torch.where(left, torch.zeros_like(xc), xc - x0.to(feat.dtype))
        hy, hx = 1.0 - ly, 1.0 - lx
        f = feat[b]                                              # (C, H, W)
        v = (f[:, y0][:, :, x0] * (hy[:, None] * hx[None, :]) + f[:, y0][:, :, x1i] * (hy[:, None] * lx[None, :])
             + f[:, y1i][:, :, x0] * (ly[:, None] * hx[None, :]) + f[:, y1i][:, :, x1i] * (ly[:, None] * lx[None, :]))
        v = v * (vy[:, None] * vx[None, :])
        v = v.reshape(C, oh, gh, ow, gw).sum((2, 4)) / count if gh * gw > 0 else torch.zeros((C, oh, ow), dtype=feat.dtype, device=feat.device)
        out.append(v)
    return torch.stack(out, 0) if out else feat.new_zeros((0, C, oh, ow))


def roi_head(x, sd, cfg, bboxes, training):
    """ResNetRoIHead.forward (head_helper.py:116-144): temporal average pool -> ROIAlign(7x7, 1/16) -> MaxPool2d(7) per
    pathway, concat, (dropout off), Linear, activation (applied in training too)."""
    res = cfg.DETECTION.ROI_XFORM_RESOLUTION
    pooled = []
    handed = _handed("head", None)
    for p, v in enumerate(x):
        m = _STORE(v).mean(2)                                    # AvgPool3d([T, 1, 1]) + squeeze
        r = roi_align(m, bboxes, (res, res), 1.0 / cfg.DETECTION.SPATIAL_SCALE_FACTOR, 0, cfg.DETECTION.ALIGNED)
        if handed is not None and handed.get(f"roi_bin{p}") is not None:      # tests only: the engine's arg-max bins
            pooled.append(_AmaxRouted.apply(r.flatten(2), handed[f"roi_bin{p}"]))
        else:
            pooled.append(r.amax((2, 3)))
    z = torch.cat(pooled, 1)
    z = F.linear(z, sd["head.projection.weight"], sd["head.projection.bias"])
    return torch.sigmoid(z) if cfg.MODEL.HEAD_ACT == "sigmoid" else F.softmax(z, dim=1)


def video_forward(sd, cfg, inputs, training=True, stats_out=None, bboxes=None):
    """SlowFast.forward (video_model_builder.py:423-441) / ResNet.forward (:645-660) + ResNetBasicHead.forward
    (head_helper.py:305-350).  Dropout is not applied (the parity harness sets MODEL.DROPOUT_RATE 0)."""
    P = len(inputs)
    two = P == 2
    x = [stem(inputs[p], sd, f"s1.pathway{p}_stem", training, stats_out) for p in range(P)]
    if two:
        x[0] = fuse(x[0], x[1], sd, "s1_fuse", cfg.SLOWFAST.ALPHA, training, stats_out)
    for i in range(4):
        name = f"s{i + 2}"
        x = res_stage(x, sd, name, cfg.RESNET.SPATIAL_STRIDES[i], cfg.RESNET.SPATIAL_DILATIONS[i],
                      cfg.RESNET.STRIDE_1X1, training, stats_out, cfg.NONLOCAL.POOL[i], cfg.NONLOCAL.INSTANTIATION,
                      cfg.NONLOCAL.GROUP[i])
        if i == 0:
            pt = _POOL1_T[cfg.MODEL.ARCH]
            if pt != 1:
                x = [_max_pool(v, (pt, 1, 1), (pt, 1, 1), (0, 0, 0), _handed(f"pathway{p}_pool", None)) for p, v in enumerate(x)]
        if two and i < 3:
            x[0] = fuse(x[0], x[1], sd, f"{name}_fuse", cfg.SLOWFAST.ALPHA, training, stats_out)
    if cfg.DETECTION.ENABLE:
        return roi_head(x, sd, cfg, bboxes, training)
    crop = cfg.DATA.TRAIN_CROP_SIZE // 32
    frames = [cfg.DATA.NUM_FRAMES // cfg.SLOWFAST.ALPHA, cfg.DATA.NUM_FRAMES] if two else \
        [cfg.DATA.NUM_FRAMES // _POOL1_T[cfg.MODEL.ARCH]]
    if cfg.MULTIGRID.SHORT_CYCLE:      # pool_size None -> nn.AdaptiveAvgPool3d((1,1,1)) (head_helper.py:251-252)
        feats = [v.mean((2, 3, 4), keepdim=True) for v in x]
    else:
        feats = [F.avg_pool3d(v, (frames[p], crop, crop), 1) for p, v in enumerate(x)]
    z = torch.cat(feats, 1).permute(0, 2, 3, 4, 1)
    z = F.linear(z, sd["head.projection.weight"], sd["head.projection.bias"])
    if not training:
        if cfg.MODEL.HEAD_ACT == "softmax":
            z = F.softmax(z, dim=4)
        elif cfg.MODEL.HEAD_ACT == "sigmoid":
            z = torch.sigmoid(z)
        z = z.mean([1, 2, 3])
    return z.reshape(z.shape[0], -1)


# ------------------------------------------------------------------------------------------------
def randomize_state(shapes, seed, dtype=torch.float32):
    """Deterministic non-degenerate parameters/buffers for parity runs, keyed by reference state_dict names.

    Fresh reference init would zero every block-final BN gamma (RESNET.ZERO_INIT_FINAL_BN), which
    makes gradients vanish; this fills BN affine/running statistics with generic values instead."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in shapes.items():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith("running_mean"):
            sd[name] = (torch.randn(shape, generator=g) * 0.1).to(dtype)
        elif name.endswith("running_var"):
            sd[name] = (torch.rand(shape, generator=g) + 0.5).to(dtype)
        elif len(shape) == 5:   # Conv3d weight: fan-in scaled so activations stay O(1)
            fan_in = shape[1] * shape[2] * shape[3] * shape[4]
            sd[name] = (torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5).to(dtype)
        elif len(shape) == 2:   # Linear weight
            sd[name] = (torch.randn(shape, generator=g) * 0.05).to(dtype)
        elif "bn" in name and name.endswith(".weight"):
            sd[name] = (torch.rand(shape, generator=g) + 0.5).to(dtype)
        else:                   # biases
            sd[name] = (torch.randn(shape, generator=g) * 0.1).to(dtype)
    return sd


def scale_final_bn(sd, factor):
    """Multiplies the block-final BatchNorm gammas (BottleneckTransform c_bn, Nonlocal bn) in place.  Deep parity cases
    (R101: 33 residual blocks) use a factor < 1 so the residual stream stays O(1): with O(1) gammas it grows like
    sqrt(depth), and the un-normalised dot-product Nonlocal (theta^T phi g ~ |x|^3) then exceeds the fp16 range that
    the engine -- and the reference under autocast -- store activations in."""
    for k in sd:
        if k.endswith("c_bn.weight") or ("nonlocal" in k and k.endswith(".bn.weight")):
            sd[k] = sd[k] * factor
    return sd


def calibrate_running_stats(sd, cfg, inputs, bboxes=None):
    """Returns a copy of ``sd`` whose BatchNorm running statistics are the batch statistics of ``inputs`` (the fixed
    point of the momentum average on that batch).  Random running statistics do not match the activations' real
    scale, so an eval-mode forward would saturate the softmax and test conditioning instead of kernels; calibrated
    statistics give eval outputs as well spread as the training-mode ones.  One training-mode oracle forward with the
    reference's momentum 0.1: new = 0.9 * old + 0.1 * batch  =>  batch = (new - 0.9 * old) / 0.1."""
    stats = {}
    fwd = x3d_forward if cfg.MODEL.MODEL_NAME == "X3D" else video_forward
    with torch.no_grad():
        if bboxes is not None:
            fwd(sd, cfg, inputs, training=True, stats_out=stats, bboxes=bboxes)
        else:
            fwd(sd, cfg, inputs, training=True, stats_out=stats)
    out = dict(sd)
    for k, new in stats.items():
        out[k] = (new - 0.9 * sd[k]) / 0.1
        if k.endswith("running_var"):
            out[k] = out[k].clamp_min(1e-4)
    return out


def synthetic_batch(cfg, batch, seed, num_classes=None, crop=None):
    """Kinetics-shaped synthetic clips: x = randn(B,3,T,S,S); SlowFast slow pathway =
    index_select(fast, 2, linspace(0, T-1, T//alpha)) (slowfast/datasets/utils.py:96-102); integer labels.
    ``crop`` overrides the spatial size (DATA.TEST_CROP_SIZE clips of the multi-view test path)."""
    g = torch.Generator().manual_seed(seed)
    T, S = cfg.DATA.NUM_FRAMES, crop or cfg.DATA.TRAIN_CROP_SIZE
    fast = torch.randn((batch, 3, T, S, S), generator=g)
    labels = torch.randint(0, num_classes or cfg.MODEL.NUM_CLASSES, (batch,), generator=g)
    if len(cfg.DATA.INPUT_CHANNEL_NUM) == 2:
        idx = torch.linspace(0, T - 1, T // cfg.SLOWFAST.ALPHA).long()
        return [torch.index_select(fast, 2, idx), fast], labels
    return [fast], labels


def synthetic_boxes(cfg, batch, seed, per_clip=3):
    """(R, 5) boxes [batch index, x1, y1, x2, y2] in crop pixels (slowfast/datasets/loader.py:66-75 layout)."""
    g = torch.Generator().manual_seed(seed)
    S = float(cfg.DATA.TRAIN_CROP_SIZE)
    rows = []
    for b in range(batch):
        for _ in range(per_clip):
            x1, y1 = (torch.rand(2, generator=g) * 0.6 * S).tolist()
            w, h = ((torch.rand(2, generator=g) * 0.35 + 0.05) * S).tolist()
            rows.append([float(b), x1, y1, min(x1 + w, S - 1.0), min(y1 + h, S - 1.0)])
    return torch.tensor(rows, dtype=torch.float32)


def loss_and_grads(sd, cfg, inputs, labels, dtype=torch.float32, bboxes=None, device=None, autocast_dtype=None,
                   loss_scale=1.0):
    """Training-mode forward + mean cross-entropy (BCE on the activated outputs for the detection head, losses.py:61-69
    "bce") + backward; returns logits, loss, {name: grad}, new running stats (all on the CPU).

    ``device`` / ``autocast_dtype`` / ``loss_scale`` run the SAME graph the way the reference trains under
    TRAIN.MIXED_PRECISION (tools/train_net.py:113-172): forward and loss under ``torch.autocast``, ``scale * loss``
    backward, gradients unscaled afterwards (GradScaler with a fixed scale).  That mode is the reference-derived
    yardstick for what fp16 compute costs on a case (tools/autocast_yardstick.py); the default is the fp32 oracle."""
    import contextlib
    dev = torch.device(device or "cpu")
    params = {k: v.detach().to(dev, dtype).clone().requires_grad_(v.is_floating_point() and "running" not in k)
              for k, v in sd.items() if v.is_floating_point()}
    stats = {}
    fwd = x3d_forward if cfg.MODEL.MODEL_NAME == "X3D" else video_forward
    ctx = torch.autocast(dev.type, dtype=autocast_dtype) if autocast_dtype is not None else contextlib.nullcontext()
    xs = [x.to(dev, dtype) for x in inputs]
    labels = labels.to(dev)
    with ctx:
        if bboxes is not None:
            logits = fwd(params, cfg, xs, training=True, stats_out=stats, bboxes=bboxes.to(dev))
        else:
            logits = fwd(params, cfg, xs, training=True, stats_out=stats)
    if bboxes is not None:
        loss = F.binary_cross_entropy(logits.float(), labels.float())
    else:
        loss = F.cross_entropy(logits.float(), labels)
    (loss * loss_scale if loss_scale != 1.0 else loss).backward()
    grads = {k: (v.grad.float() / loss_scale).cpu() if (loss_scale != 1.0 or dev.type != "cpu") else v.grad
             for k, v in params.items() if v.requires_grad}
    stats = {k: v.detach().float().cpu() for k, v in stats.items()}
    return logits.detach().float().cpu(), loss.detach().float().cpu(), grads, stats


def grad_norm(grads):
    """Global L2 norm of all parameter gradients (slowfast/models/optimizer.py:362-379 get_grad_norm_)."""
    return torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values()))
