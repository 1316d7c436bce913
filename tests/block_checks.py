"""Block-level parity (well-conditioned: every BatchNorm sees >= 128 samples): the fused engine schedules
(engine.StemFn / FuseFn / ResBlockFn / ConvBNActFn) against the oracle's functional restatement."""
import torch

from oracle import video_ref
from slowfast_amd.resblocks import BottleneckTransform, ResBlock
from slowfast_amd.stems import ResNetBasicStem
from slowfast_amd.video_models import FuseFastToSlow
from tests.kernel_checks import cl_to_host, host_to_cl

TOL = 2e-3   # relative L2, fp16 storage + fp32 accumulation -- asserted on EVERY compared quantity, no fallback

# ReLU masks.  An element whose pre-activation lies within fp16 round-off of zero lands on either side of the ReLU in two
# correct fp16 realisations, and although its forward effect is O(round-off) its backward effect is O(1) (its gradient is
# switched on or off): ~0.05 % of the elements of a layer, i.e. a few per cent of every gradient in relative L2 -- for ANY
# fp16 implementation, the reference under autocast included.  To test the fused schedules tightly, the oracle's BACKWARD
# is run with the masks the engine actually used (engine.CAPTURE exposes the tensors that decide them;
# oracle.video_ref._ReluFixedMask): flipped elements are thereby excluded from the comparison and everything else -- all
# the convolution / BatchNorm / residual arithmetic of the block, forward and backward -- must agree to TOL.  The number of
# flipped elements is itself bounded (a wrong BatchNorm or a wrong tap flips a large fraction, not 0.1 %).
MAX_FLIP_FRACTION = 3e-3


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


def _compare(mod, prefix, got, ref, ref_grads, stats, tol=TOL):
    errs = {k: rel(got[k], ref[k]) for k in ref}
    gsq, esq = 0.0, 0.0
    for k, prm in mod.named_parameters():
        r = ref_grads[prefix + k]
        errs["grad:" + prefix + k] = rel(prm.grad.cpu(), r)
        gsq += float(r.double().pow(2).sum())
        esq += float(prm.grad.cpu().double().pow(2).sum())
    errs["grad_norm"] = abs(esq ** 0.5 - gsq ** 0.5) / gsq ** 0.5
    msd = mod.state_dict()
    for k, v in stats.items():
        errs["stat:" + k] = rel(msd[k[len(prefix):]].cpu(), v)
    bad = {k: v for k, v in errs.items() if v > tol}
    assert not bad, bad
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
    x = torch.randn(shape).half().float()
    with torch.no_grad():
        oshape = video_ref.res_block(x, sd_prefixed(sd, "blk."), "blk", stride, dilation, False, True, None).shape
    dout = torch.randn(oshape).half().float()

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
        o32 = video_ref.stem(x.half().float(), sd_prefixed(sd, "st."), "st", True, None)
    dout = torch.randn(o32.shape).half().float()
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
        o = video_ref.stem(x.half().float(), p, "st", True, st, masks=masks)
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
    xf = torch.randn(shape_fast).half().float()
    xs = torch.randn((N, dim_in * 4, T // alpha, H, W)).half().float()
    dcat = torch.randn((N, dim_in * 4 + dim_in * ratio, T // alpha, H, W)).half().float()
    dpass = torch.randn(shape_fast).half().float()      # gradient reaching x_f from the Fast pathway itself
    xfc, xsc = host_to_cl(xf, device).requires_grad_(True), host_to_cl(xs, device).requires_grad_(True)
    cat, xf_out = fz([xsc, xfc])
    torch.autograd.backward([cat, xf_out], [host_to_cl(dcat, device), host_to_cl(dpass, device)])
    got = {"cat": cl_to_host(cat), "dx_s": cl_to_host(xsc.grad), "dx_f": cl_to_host(xfc.grad)}
    masks = {"relu": (got["cat"][:, dim_in * 4:] > 0).float()}      # the lateral branch's ReLU mask is its own output's sign

    def body(p, st):
        xfr, xsr = xf.clone().requires_grad_(True), xs.clone().requires_grad_(True)
        o = video_ref.fuse(xsr, xfr, p, "fz", alpha, True, st, masks=masks)
        (o * dcat).sum().backward()
        return {"cat": o.detach(), "dx_s": xsr.grad, "dx_f": xfr.grad.half().float() + dpass}

    ref, rg, st = _oracle_run(_Case(sd, "fz.", body))
    return _compare(fz, "fz.", got, ref, rg, st)


def check_bottleneck_alone(device, shape, seed=11):
    """BottleneckTransform.forward on its own (a -> b -> c with every unit materialised)."""
    from slowfast_amd import engine
    torch.manual_seed(seed)
    t = BottleneckTransform(shape[1], 32, 3, 1, 8, 1)
    sd = _load(t, seed)
    t = t.to(device).train()
    x = torch.randn(shape).half().float()
    dout = torch.randn((shape[0], 32) + tuple(shape[2:])).half().float()
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
