"""Block-level parity (well-conditioned: every BatchNorm sees >= 128 samples): the fused engine schedules
(engine.StemFn / FuseFn / ResBlockFn / ConvBNActFn) against the oracle's functional restatement."""
import torch

from oracle import video_ref
from slowfast_amd.resblocks import BottleneckTransform, ResBlock
from slowfast_amd.stems import ResNetBasicStem
from slowfast_amd.video_models import FuseFastToSlow
from tests.kernel_checks import ACT, F16_EPS, cl_to_host, host_to_cl

EPS_SCALE = F16_EPS / 2.0 ** -10     # 1 in an fp16 process, 8 under SF_ACT_DTYPE=bf16: every bound below is stated for fp16
TOL = 2e-3 * EPS_SCALE   # relative L2, 16-bit storage + fp32 accumulation -- asserted on EVERY compared quantity, no fallback

# ReLU masks.  An element whose pre-activation lies within fp16 round-off of zero lands on either side of the ReLU in two
# correct fp16 realisations, and although its forward effect is O(round-off) its backward effect is O(1) (its gradient is
# switched on or off): ~0.05 % of the elements of a layer, i.e. a few per cent of every gradient in relative L2 -- for ANY
# fp16 implementation, the reference under autocast included.  To test the fused schedules tightly, the oracle's BACKWARD
# is run with the masks the engine actually used (engine.CAPTURE exposes the tensors that decide them;
# oracle.video_ref._ReluFixedMask): flipped elements are thereby excluded from the comparison and everything else -- all
# the convolution / BatchNorm / residual arithmetic of the block, forward and backward -- must agree to TOL.  The number of
# flipped elements is itself bounded (a wrong BatchNorm or a wrong tap flips a large fraction, not 0.1 %).
MAX_FLIP_FRACTION = 3e-3 * EPS_SCALE


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-12))


def _load(mod, seed):
    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    sd = video_ref.randomize_state(shapes, seed)
    mod.load_state_dict(sd)
    return sd


def sd_prefixed(sd, prefix):
    return {prefix + k: v for k, v in sd.items()}


def _oracle_params(sd, prefix):
    return {prefix + k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}


def _oracle_run(fn):
    """fn(params, stats) -> dict name -> tensor (outputs, input grads); returns (results, parameter grads, stats)."""
    p, st = fn.params(), {}
    res = fn(p, st)
    grads = {k: v.grad for k, v in p.items() if v.requires_grad and v.grad is not None}
    return res, grads, st


YARD_BLOCK = 1.5    # engine deviation <= max(TOL, YARD_BLOCK x the oracle's own fp16-storage deviation of that quantity)


def _storage_yardstick(case, ref, ref_grads, zero_floor=0.0):
    """Per-quantity deviation of the ORACLE's fp16 storage model (torch fp32 arithmetic on the reference graph, every tensor an
    fp16-storage implementation keeps in HBM rounded to fp16, video_ref.fp16_storage_model) from its fp32 run, with the same
    masks / routes handed to both backward passes.  A few quantities of these blocks are ill-conditioned against STORAGE
    rounding itself -- e.g. the weight gradient of a convolution in front of a BatchNorm picks up the rounding error of the
    BatchNorm-backward projection coefficient coherently: 1.06e-2 on Nonlocal(dot_product).conv_out.weight for torch fp32
    arithmetic with fp16-stored tensors, 1.07e-2 for the engine -- so a flat 2e-3 cannot hold for ANY fp16-storage
    implementation there; everywhere else the yardstick is far below TOL and TOL decides."""
    with video_ref.fp16_storage_model():
        res, grads, _ = _oracle_run(case)
    yard = {k: rel(res[k], ref[k]) for k in ref}
    floor = 0.0
    if zero_floor:
        floor = zero_floor * (sum(float(g.double().pow(2).sum()) for g in ref_grads.values()) / len(ref_grads)) ** 0.5
    gsq = sum(float(g.double().pow(2).sum()) for g in ref_grads.values())
    esq = sum(float(grads[k].double().pow(2).sum()) for k in ref_grads)
    for k, r in ref_grads.items():
        yard["grad:" + k] = float((grads[k].double() - r.double()).norm() / (max(float(r.double().norm()), floor) + 1e-12))
    yard["grad_norm"] = abs(esq ** 0.5 - gsq ** 0.5) / gsq ** 0.5
    return yard


def _compare(mod, prefix, got, ref, ref_grads, stats, tol=TOL, zero_floor=0.0, yard=None):
    """``zero_floor``: a parameter whose gradient vanishes identically in exact arithmetic (a bias in front of a BatchNorm or
    of a softmax over the axis it is constant on) has a reference gradient of pure round-off; such gradients are compared
    against ``zero_floor`` x the RMS parameter-gradient norm of the block instead of against their own (meaningless) norm."""
    errs = {k: rel(got[k], ref[k]) for k in ref}
    gsq, esq = 0.0, 0.0
    named = list(mod.named_parameters())
    floor = 0.0
    if zero_floor:
        floor = zero_floor * (sum(float(ref_grads[prefix + k].double().pow(2).sum()) for k, _ in named) / len(named)) ** 0.5
    for k, prm in named:
        r = ref_grads[prefix + k]
        g = prm.grad.cpu()
        errs["grad:" + prefix + k] = float((g.double() - r.double()).norm() / (max(float(r.double().norm()), floor) + 1e-12))
        gsq += float(r.double().pow(2).sum())
        esq += float(g.double().pow(2).sum())
    errs["grad_norm"] = abs(esq ** 0.5 - gsq ** 0.5) / gsq ** 0.5
    msd = mod.state_dict()
    for k, v in stats.items():
        errs["stat:" + k] = rel(msd[k[len(prefix):]].cpu(), v)
    bad = {k: (v, None if yard is None else yard.get(k)) for k, v in errs.items()
           if v > max(tol, YARD_BLOCK * (yard or {}).get(k, 0.0))}
    assert not bad, bad
    if yard is not None:
        errs["above_tol_by_yardstick"] = sorted(k for k, v in errs.items() if isinstance(v, float) and v > tol)
    return errs


def _flip_fraction(mask_engine, mask_ref):
    return float((mask_engine != mask_ref).float().mean())


class _Case:
    """Callable oracle evaluation with fresh leaf parameters per run."""

    def __init__(self, sd, prefix, body):
        self.sd, self.prefix, self.body = sd, prefix, body

    def params(self):
        return _oracle_params(self.sd, self.prefix)

    def __call__(self, p, st):
        return self.body(p, st)


def check_resblock(device, dim_in, dim_out, temp_k, stride, inner, shape, dilation=1, seed=3):
    from slowfast_amd import engine
    torch.manual_seed(seed)
    blk = ResBlock(dim_in, dim_out, temp_k, stride, BottleneckTransform, inner, dilation=dilation)
    sd = _load(blk, seed)
    blk = blk.to(device).train()
    x = torch.randn(shape).to(ACT).float()
    with torch.no_grad():
        oshape = video_ref.res_block(x, sd_prefixed(sd, "blk."), "blk", stride, dilation, False, True, None).shape
    dout = torch.randn(oshape).to(ACT).float()

    # the engine first: its masks (decided by the raw convolution outputs it stored and its BatchNorm scale / shift)
    xc = host_to_cl(x, device).requires_grad_(True)
    engine.CAPTURE = []
    try:
        out = blk(xc)
        cap = engine.CAPTURE[-1]
    finally:
        engine.CAPTURE = None
    out.backward(host_to_cl(dout, device))
    masks = {}
    for key, raw, (sc, sh) in zip(("a", "b"), cap["raw"], cap["bn"]):
        pre = cl_to_host(raw) * sc.float().cpu().view(1, -1, 1, 1, 1) + sh.float().cpu().view(1, -1, 1, 1, 1)
        masks[key] = (pre > 0).float()
    masks["out"] = (cl_to_host(cap["out"]) > 0).float()

    seen = {}

    def body(p, st, use_masks=True):
        xr = x.clone().requires_grad_(True)
        o = video_ref.res_block(xr, p, "blk", stride, dilation, False, True, st, masks=masks if use_masks else None)
        o.backward(dout)
        return {"out": o.detach(), "dx": xr.grad}

    ref, rg, st = _oracle_run(_Case(sd, "blk.", body))
    # how many elements actually flipped against the fp32 reference's own masks (diagnostic + sanity bound)
    with torch.no_grad():
        o32 = video_ref.res_block(x, sd_prefixed(sd, "blk."), "blk", stride, dilation, False, True, None)
    flips = _flip_fraction(masks["out"], (o32 > 0).float())
    assert flips <= MAX_FLIP_FRACTION, flips
    errs = _compare(blk, "blk.", {"out": cl_to_host(out), "dx": cl_to_host(xc.grad)}, ref, rg, st)
    errs["flipped_out_fraction"] = flips
    return errs


def _pre_mask(raw, scale, shift):
    pre = cl_to_host(raw) * scale.float().cpu().view(1, -1, 1, 1, 1) + shift.float().cpu().view(1, -1, 1, 1, 1)
    return (pre > 0).float()


def check_stem(device, dim_out, kernel, shape, seed=5):
    """conv -> BN -> ReLU -> max-pool with the engine's ReLU mask AND its pooling routes (byte arg-max table) handed to the
    oracle's backward: a pooled window whose two largest entries differ by less than fp16 round-off may route its
    gradient to either, which moves an O(1) contribution of the weight gradient between two input patches."""
    from slowfast_amd import engine
    torch.manual_seed(seed)
    stem = ResNetBasicStem(3, dim_out, kernel, [1, 2, 2], [kernel[0] // 2, 3, 3])
    sd = _load(stem, seed)
    stem = stem.to(device).train()
    x = torch.randn(shape)
    with torch.no_grad():
        o32 = video_ref.stem(x.to(ACT).float(), sd_prefixed(sd, "st."), "st", True, None)
    dout = torch.randn(o32.shape).to(ACT).float()
    engine.CAPTURE = []
    try:
        out = stem(x.to(device))
        cap = engine.CAPTURE[-1]
    finally:
        engine.CAPTURE = None
    out.backward(host_to_cl(dout, device))
    # byte arg-max table [N, T, Ho, Wo, C]: window-local index kh*3 + kw of the routed element -> flat h*W + w
    raw = cap["raw"][0]
    N, C, T, H, W = raw.shape
    arg = cap["argmax"].cpu().long()                       # (N, T, Ho, Wo, C)
    Ho, Wo = arg.shape[2], arg.shape[3]
    ho = torch.arange(Ho).view(1, 1, Ho, 1, 1)
    wo = torch.arange(Wo).view(1, 1, 1, Wo, 1)
    h = (ho * 2 - 1 + arg // 3).clamp(0, H - 1)
    w = (wo * 2 - 1 + arg % 3).clamp(0, W - 1)
    pool_index = (h * W + w).permute(0, 4, 1, 2, 3).contiguous()
    masks = {"relu": _pre_mask(raw, *cap["bn"][0]), "pool_index": pool_index}

    def body(p, st):
        o = video_ref.stem(x.to(ACT).float(), p, "st", True, st, masks=masks)
        o.backward(dout)
        return {"out": o.detach()}

    ref, rg, st = _oracle_run(_Case(sd, "st.", body))
    # the routed forward must reproduce the true max-pool (a wrong route shows here at fp32 accuracy)
    assert rel(ref["out"], o32) < 2e-3, rel(ref["out"], o32)
    flips = _flip_fraction((cl_to_host(out) > 0).float(), (o32 > 0).float())
    assert flips <= MAX_FLIP_FRACTION, flips
    return _compare(stem, "st.", {"out": cl_to_host(out)}, ref, rg, st)


def check_fuse(device, dim_in, ratio, kernel, alpha, shape_fast, seed=9):
    torch.manual_seed(seed)
    fz = FuseFastToSlow(dim_in, ratio, kernel, alpha)
    sd = _load(fz, seed)
    fz = fz.to(device).train()
    N, C, T, H, W = shape_fast
    xf = torch.randn(shape_fast).to(ACT).float()
    xs = torch.randn((N, dim_in * 4, T // alpha, H, W)).to(ACT).float()
    dcat = torch.randn((N, dim_in * 4 + dim_in * ratio, T // alpha, H, W)).to(ACT).float()
    dpass = torch.randn(shape_fast).to(ACT).float()      # gradient reaching x_f from the Fast pathway itself
    xfc, xsc = host_to_cl(xf, device).requires_grad_(True), host_to_cl(xs, device).requires_grad_(True)
    cat, xf_out = fz([xsc, xfc])
    torch.autograd.backward([cat, xf_out], [host_to_cl(dcat, device), host_to_cl(dpass, device)])
    got = {"cat": cl_to_host(cat), "dx_s": cl_to_host(xsc.grad), "dx_f": cl_to_host(xfc.grad)}
    masks = {"relu": (got["cat"][:, dim_in * 4:] > 0).float()}      # the lateral branch's ReLU mask is its own output's sign

    def body(p, st):
        xfr, xsr = xf.clone().requires_grad_(True), xs.clone().requires_grad_(True)
        o = video_ref.fuse(xsr, xfr, p, "fz", alpha, True, st, masks=masks)
        (o * dcat).sum().backward()
        return {"cat": o.detach(), "dx_s": xsr.grad, "dx_f": xfr.grad.to(ACT).float() + dpass}

    ref, rg, st = _oracle_run(_Case(sd, "fz.", body))
    return _compare(fz, "fz.", got, ref, rg, st)


def check_bottleneck_alone(device, shape, seed=11):
    """BottleneckTransform.forward on its own (a -> b -> c with every unit materialised)."""
    from slowfast_amd import engine
    torch.manual_seed(seed)
    t = BottleneckTransform(shape[1], 32, 3, 1, 8, 1)
    sd = _load(t, seed)
    t = t.to(device).train()
    x = torch.randn(shape).to(ACT).float()
    dout = torch.randn((shape[0], 32) + tuple(shape[2:])).to(ACT).float()
    xc = host_to_cl(x, device).requires_grad_(True)
    engine.CAPTURE = []
    try:
        out = t(xc)
        caps = list(engine.CAPTURE)
    finally:
        engine.CAPTURE = None
    out.backward(host_to_cl(dout, device))
    assert len(caps) == 3
    masks = {"a": _pre_mask(caps[0]["raw"][0], *caps[0]["bn"][0]), "b": _pre_mask(caps[1]["raw"][0], *caps[1]["bn"][0])}

    def body(p, st):
        xr = x.clone().requires_grad_(True)
        y = video_ref._conv(xr, p["t.a.weight"], None, 1, (1, 0, 0))
        y = video_ref._relu(video_ref._bn(y, p, "t.a_bn", True, st), masks, "a")
        y = video_ref._conv(y, p["t.b.weight"], None, 1, (0, 1, 1))
        y = video_ref._relu(video_ref._bn(y, p, "t.b_bn", True, st), masks, "b")
        y = video_ref._conv(y, p["t.c.weight"])
        o = video_ref._bn(y, p, "t.c_bn", True, st)
        o.backward(dout)
        return {"out": o.detach(), "dx": xr.grad}

    ref, rg, st = _oracle_run(_Case(sd, "t.", body))
    return _compare(t, "t.", {"out": cl_to_host(out), "dx": cl_to_host(xc.grad)}, ref, rg, st)


# ---- the other fused block schedules: X3DBlockFn, NonlocalFn, MultiScaleBlockFn -------------------------------------------------
def _capture_forward(fn):
    from slowfast_amd import engine
    engine.CAPTURE = []
    try:
        out = fn()
        caps = list(engine.CAPTURE)
    finally:
        engine.CAPTURE = None
    return out, caps


def check_x3d_block(device, dim_in, dim_out, stride, inner, shape, block_idx=0, seed=13):
    """relu(shortcut(x) + X3DTransform(x)) (engine X3DBlockFn: 1x1x1 igemm, depthwise 3x3x3 stencil + BN statistics, SE,
    gate * BN -> Swish, channel padding) against oracle.video_ref.x3d_block with the engine's two ReLU masks handed to its
    backward: outputs, input gradient, EVERY parameter gradient, gradient norm and running statistics at TOL."""
    from slowfast_amd.x3d import X3DTransform
    torch.manual_seed(seed)
    blk = ResBlock(dim_in, dim_out, 3, stride, X3DTransform, inner, num_groups=inner, block_idx=block_idx)
    sd = _load(blk, seed)
    blk = blk.to(device).train()
    x = torch.randn(shape).to(ACT).float()
    with torch.no_grad():
        o32 = video_ref.x3d_block(x, sd_prefixed(sd, "blk."), "blk", stride, True, None)
    dout = torch.randn(o32.shape).to(ACT).float()
    xc = host_to_cl(x, device).requires_grad_(True)
    out, caps = _capture_forward(lambda: blk(xc))
    dpad = torch.nn.functional.pad(dout, (0, 0, 0, 0, 0, 0, 0, out.shape[1] - dim_out))     # 54 -> 56: pad channels stay zero
    out.backward(host_to_cl(dpad, device))
    cap = [c for c in caps if c["kind"] == "x3d_block"][-1]
    masks = {"a": _pre_mask(cap["raw"][0], *cap["bn"][0])[:, :inner], "out": (cl_to_host(cap["out"]) > 0).float()[:, :dim_out]}
    if cap.get("se_h") is not None:
        masks["se"] = (cap["se_h"] > 0).float().cpu().view(cap["se_h"].shape[0], -1, 1, 1, 1)

    def body(p, st):
        xr = x.clone().requires_grad_(True)
        with video_ref.handed_masks({"blk": masks}):
            o = video_ref.x3d_block(xr, p, "blk", stride, True, st)
        o.backward(dout)
        return {"out": o.detach(), "dx": xr.grad}

    case = _Case(sd, "blk.", body)
    ref, rg, st = _oracle_run(case)
    flips = _flip_fraction(masks["out"], (o32 > 0).float())
    assert flips <= MAX_FLIP_FRACTION, flips
    got = {"out": cl_to_host(out)[:, :dim_out], "dx": cl_to_host(xc.grad)[:, :dim_in]}
    errs = _compare(blk, "blk.", got, ref, rg, st, yard=_storage_yardstick(case, ref, rg))
    errs["flipped_out_fraction"] = flips
    return errs


def check_nonlocal(device, dim, dim_inner, pool_size, shape, instantiation="softmax", seed=17):
    """Nonlocal block (engine NonlocalFn: bias convolutions, MaxPool3d with byte arg-max, the two batched affinity GEMMs,
    softmax | 1/N, BatchNorm + residual) against oracle.video_ref.nonlocal_block with the engine's max-pool routes handed to
    its backward: every quantity at TOL."""
    from slowfast_amd.nonlocal_block import Nonlocal
    from tests.model_checks import _window_route
    torch.manual_seed(seed)
    nl = Nonlocal(dim, dim_inner, pool_size, instantiation=instantiation)
    sd = _load(nl, seed)
    nl = nl.to(device).train()
    x = torch.randn(shape).to(ACT).float()
    dout = torch.randn(shape).to(ACT).float()
    xc = host_to_cl(x, device).requires_grad_(True)
    out, caps = _capture_forward(lambda: nl(xc))
    out.backward(host_to_cl(dout, device))
    table = {}
    for c in caps:
        if c["kind"] == "nonlocal":
            table["nl"] = {"pool_route": _window_route(c["argmax"], c["kernel"], c["kernel"], (0, 0, 0), c["in_shape"][2:])}
    assert table or not nl.use_pool

    def body(p, st):
        xr = x.clone().requires_grad_(True)
        with video_ref.handed_masks(table):
            o = video_ref.nonlocal_block(xr, p, "nl", pool_size, instantiation, True, st)
        o.backward(dout)
        return {"out": o.detach(), "dx": xr.grad}

    case = _Case(sd, "nl.", body)
    ref, rg, st = _oracle_run(case)
    # conv_g / conv_out bias (in front of the BatchNorm) and, with softmax, conv_phi bias have identically vanishing gradients
    return _compare(nl, "nl.", {"out": cl_to_host(out), "dx": cl_to_host(xc.grad)}, ref, rg, st, zero_floor=0.1,
                    yard=_storage_yardstick(case, ref, rg, zero_floor=0.1))


def check_multiscale_block(device, dim, dim_out, heads, thw, stride_q, stride_kv, B=2, cls=True, seed=19, tol=TOL):
    """MultiScaleBlock (engine MultiScaleBlockFn: LayerNorm, qkv Linear, depthwise pooling + per-head LayerNorm, fused
    attention with decomposed relative positions and residual pooling, proj, max-pooled skip, Mlp with GELU, residuals in
    the GEMM epilogues) against oracle.mvit_ref.block -- MViTv2 options (DIM_MUL_IN_ATT, rel-pos, residual pooling).  No
    ReLU in this block; the skip path's max-pool windows overlap, ties within fp16 round-off are rare enough at this size to
    stay inside TOL: outputs, input gradient, EVERY parameter gradient and the gradient norm at max(TOL, YARD_BLOCK x the
    oracle's own fp16-storage deviation) -- see _storage_yardstick."""
    from oracle import mvit_ref
    from slowfast_amd.mvit import MultiScaleBlock
    torch.manual_seed(seed)
    blk = MultiScaleBlock(dim, dim_out, heads, thw, qkv_bias=True, norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6),
                          kernel_q=(3, 3, 3), kernel_kv=(3, 3, 3), stride_q=stride_q, stride_kv=stride_kv, has_cls_embed=cls,
                          rel_pos_spatial=True, rel_pos_temporal=True, residual_pooling=True, dim_mul_in_att=True)
    shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
    sd = mvit_ref.randomize_state(shapes, seed)
    blk.load_state_dict(sd)
    blk = blk.to(device).train()
    N = thw[0] * thw[1] * thw[2] + (1 if cls else 0)
    x = torch.randn((B, N, dim)).to(ACT).float()
    psd = sd_prefixed(sd, "blk.")
    with torch.no_grad():
        o32, thw_new = mvit_ref.block(x, psd, "blk", list(thw), heads, list(stride_q), list(stride_kv), cls, None, True, True)
    dout = torch.randn(o32.shape).to(ACT).float()
    xc = x.to(device).to(ACT).requires_grad_(True)
    out, thw_e = blk(xc, list(thw))
    assert list(thw_e) == list(thw_new), (thw_e, thw_new)
    out.backward(dout.to(device).to(out.dtype))

    def body(p, st):
        xr = x.clone().requires_grad_(True)
        o, _ = mvit_ref.block(xr, p, "blk", list(thw), heads, list(stride_q), list(stride_kv), cls, None, True, True)
        o.backward(dout)
        return {"out": o.detach(), "dx": xr.grad}

    case = _Case(sd, "blk.", body)
    ref, rg, st = _oracle_run(case)
    got = {"out": out.detach().float().cpu(), "dx": xc.grad.float().cpu()}
    # norm_k.bias: the same vector added to every key leaves the softmax unchanged -- its gradient vanishes identically
    return _compare(blk, "blk.", got, ref, rg, st, tol=tol, zero_floor=0.1,
                    yard=_storage_yardstick(case, ref, rg, zero_floor=0.1))


def check_bn_part_tag_second_consumer(device, seed=5):
    """Two consecutive identity ResBlocks whose intermediate output has a SECOND autograd consumer (a feature tap): the fused
    BatchNorm-backward partial sums the second block tags onto its input gradient describe only ITS contribution; autograd may
    add the tap's gradient into the same tensor in place.  engine.tagged_bn_part must notice (storage + version counter) and the
    first block must fall back to the separate reduction: identical gradients with the fusion on and off."""
    from slowfast_amd import engine
    assert engine.BN_FUSE_REDUCE
    torch.manual_seed(seed)
    b1 = ResBlock(32, 32, 1, 1, BottleneckTransform, 8)
    b2 = ResBlock(32, 32, 1, 1, BottleneckTransform, 8)
    _load(b1, seed), _load(b2, seed + 1)
    b1, b2 = b1.to(device).train(), b2.to(device).train()
    x = torch.randn((2, 32, 2, 8, 8)).to(ACT).float()
    tapw = torch.randn((2, 32, 2, 8, 8)).to(ACT).float()
    d2 = torch.randn((2, 32, 2, 8, 8)).to(ACT).float()
    # unit level: a tag stops describing a tensor the moment the tensor is written to
    t = torch.zeros(4)
    y0 = torch.zeros(3)
    engine.tag_bn_part(t, y0, "part")
    assert engine.tagged_bn_part(t, y0) == "part" and engine.tagged_bn_part(t, torch.zeros(3)) is None
    t.add_(1.0)
    assert engine.tagged_bn_part(t, y0) is None

    def run(fuse, tap_first):
        engine.BN_FUSE_REDUCE = fuse
        try:
            for b in (b1, b2):
                for p in b.parameters():
                    p.grad = None
            xc = host_to_cl(x, device).requires_grad_(True)
            y1 = b1(xc)
            tap = (y1.float() * host_to_cl(tapw, device).float()).sum()
            y2 = b2(y1)
            main = (y2.float() * host_to_cl(d2, device).float()).sum()
            (tap + main if tap_first else main + tap).backward()
            return [p.grad.detach().float().cpu().clone() for p in b1.parameters()] + [cl_to_host(xc.grad)]
        finally:
            engine.BN_FUSE_REDUCE = True

    for tap_first in (False, True):
        ref, got = run(False, tap_first), run(True, tap_first)
        for a, b in zip(ref, got):
            assert rel(b, a) < 1e-5, (tap_first, rel(b, a))
