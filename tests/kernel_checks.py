"""Kernel parity checks shared by the CPU (host simulator) and GPU (-m gpu) test files.

Reference = the plain torch fp32 op the reference model calls (F.conv3d, F.batch_norm, F.max_pool3d,
autograd) evaluated on the CPU on the SAME fp16-rounded operands, so the only differences are the
accumulation order and the fp16 rounding of the stored result (tolerances below are stated in units
of fp16 epsilon 2^-10 ~ 1e-3).
"""
import torch
import torch.nn.functional as F

from oracle import video_ref
from slowfast_amd import lib as _sflib
from slowfast_amd import ops

ACT = _sflib.act_dtype()            # 16-bit storage type of this process: fp16, or bf16 under SF_ACT_DTYPE=bf16
F16_EPS = 2 * _sflib.act_eps()       # one ulp of that type at 1.0: 2^-10 (fp16) / 2^-7 (bf16); every tolerance below is a multiple
EPS_SCALE = F16_EPS / 2.0 ** -10    # 1 (fp16) / 8 (bf16): scales the tolerances that are written as plain numbers for fp16
video_ref.STORAGE_DTYPE = ACT        # the oracle's storage-model yardstick rounds to the type this process stores in


def host_to_cl(x, device):
    """NCTHW float (cpu) -> channels-last fp16 on `device` (test utility, not the product path)."""
    x = x.to(ACT).permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)
    return x.to(device)


def cl_to_host(x):
    return x.detach().float().cpu().contiguous()


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def assert_close(name, got, ref, tol):
    e = rel_err(got, ref)
    assert e <= tol, f"{name}: max-abs error / max|ref| = {e:.3e} > {tol:.1e}"
    return e


def make_conv_case(seed, in_shape, Co, k, s, p, d, Cw=None):
    g = torch.Generator().manual_seed(seed)
    N, Ci, T, H, W = in_shape
    Cw = Ci if Cw is None else Cw
    x = torch.randn(in_shape, generator=g)
    if Cw < Ci:
        x[:, Cw:] = 0
    x = x.to(ACT).float()
    w = (torch.randn((Co, Cw) + tuple(k), generator=g) / (Cw * k[0] * k[1] * k[2]) ** 0.5)
    return x, w


def check_conv_fwd(device, in_shape, Co, k, s, p, d=(1, 1, 1), Cw=None, affine=False, ldx_extra=0, seed=0):
    x, w = make_conv_case(seed, in_shape, Co, k, s, p, d, Cw)
    geom = ops.ConvGeom(in_shape, Co, k, s, p, d, Cw=Cw)
    wf, wd = ops.prep_weights(w.to(device), geom)
    w16 = w.to(ACT).float()
    xin = x
    in_affine = None
    if affine:
        g = torch.Generator().manual_seed(seed + 1)
        sc = torch.randn(in_shape[1], generator=g)
        sh = torch.randn(in_shape[1], generator=g) * 0.5
        xin = F.relu(x * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)).to(ACT).float()
        in_affine = (sc.to(device), sh.to(device), True)
    if ldx_extra:
        N, Ci, T, H, W = in_shape
        big = ops.cl_empty((N, Ci + ldx_extra, T, H, W), device, zero=True)
        xc = big[:, ldx_extra:]
        xc.copy_(host_to_cl(x, device))
    else:
        xc = host_to_cl(x, device)
    y, part = ops.conv_fwd(xc, wf, geom, in_affine=in_affine, stats=True)
    ref = F.conv3d(xin[:, : w.shape[1]], w16, None, s, p, d)
    assert tuple(y.shape) == tuple(ref.shape)
    e = assert_close("conv_fwd", cl_to_host(y), ref, 2 * F16_EPS)
    tot = part.sum(0).cpu()
    assert_close("conv_fwd stats sum", tot[0], ref.sum((0, 2, 3, 4)), 1e-4 * max(1.0, float(ref.abs().sum((0, 2, 3, 4)).max() / (ref.sum((0, 2, 3, 4)).abs().max() + 1e-9))))
    assert_close("conv_fwd stats sumsq", tot[1], (ref * ref).sum((0, 2, 3, 4)), 1e-4)
    return e


def check_conv_fwd_fused(device, in_shape, Co, k, s, p, d=(1, 1, 1), resid=False, relu=True, bias=True, seed=0):
    """sf_conv_fwd_fused (eval-mode conv with BatchNorm folded into weights + bias, residual / ReLU epilogue) vs
    relu(F.conv3d(x, w, b) + resid) on the same fp16-rounded operands."""
    x, w = make_conv_case(seed, in_shape, Co, k, s, p, d)
    geom = ops.ConvGeom(in_shape, Co, k, s, p, d)
    wf, _ = ops.prep_weights(w.to(device), geom, need_dgrad=False)
    g = torch.Generator().manual_seed(seed + 5)
    b = torch.randn(Co, generator=g) * 0.5 if bias else None
    ref = F.conv3d(x, w.to(ACT).float(), b, s, p, d)
    r = None
    if resid:
        rr = torch.randn(ref.shape, generator=g).to(ACT).float()
        ref = ref.to(ACT).float() + rr          # the epilogue adds the residual to the fp16-rounded tile
        r = host_to_cl(rr, device)
    if relu:
        ref = F.relu(ref)
    y = ops.conv_fwd_fused(host_to_cl(x, device), wf, geom, bias=None if b is None else b.to(device), resid=r, relu=relu)
    assert tuple(y.shape) == tuple(ref.shape)
    got = cl_to_host(y)
    if relu:
        assert float(got.min()) >= 0.0
    return assert_close("conv_fwd_fused", got, ref, 2 * F16_EPS)


def check_conv_dgrad(device, in_shape, Co, k, s, p, d=(1, 1, 1), resid=False, seed=0):
    x, w = make_conv_case(seed, in_shape, Co, k, s, p, d)
    geom = ops.ConvGeom(in_shape, Co, k, s, p, d)
    wf, wd = ops.prep_weights(w.to(device), geom)
    w16 = w.to(ACT).float()
    g = torch.Generator().manual_seed(seed + 2)
    dy = torch.randn(geom.out_shape, generator=g).to(ACT).float()
    xr = x.clone().requires_grad_(True)
    F.conv3d(xr, w16, None, s, p, d).backward(dy)
    ref = xr.grad
    r = None
    if resid:
        rr = torch.randn(in_shape, generator=g).to(ACT).float()
        ref = (ref.to(ACT).float() + rr)
        r = host_to_cl(rr, device)
    dx = ops.conv_dgrad(host_to_cl(dy, device), wd, geom, resid=r)
    e = assert_close("conv_dgrad", cl_to_host(dx), ref, 2 * F16_EPS)
    if resid:           # masked residual: only elements whose mask bit is set are added
        N, Ci, T, H, W = in_shape
        Mi = N * T * H * W
        bits = torch.randint(0, 256, (Mi, Ci // 8), generator=g, dtype=torch.int32)
        keep = ((bits.unsqueeze(-1) >> torch.arange(8, dtype=torch.int32)) & 1).reshape(Mi, Ci).bool()
        keep5 = keep.reshape(N, T, H, W, Ci).permute(0, 4, 1, 2, 3)
        ref2 = xr.grad.to(ACT).float() + rr * keep5
        dx2 = ops.conv_dgrad(host_to_cl(dy, device), wd, geom, resid=r, resid_bits=bits.to(torch.uint8).to(device))
        assert_close("conv_dgrad masked residual", cl_to_host(dx2), ref2, 2 * F16_EPS)
    return e


def check_conv_dgrad_bn(device, in_shape, Co, k, p, d=(1, 1, 1), resid=False, seed=0):
    """sf_conv_dgrad_bn: the data gradient is unchanged (bit for bit) and the fused epilogue's partial table sums to what
    sf_bn_bwd_reduce computes from the STORED gradient in a separate pass: sum g and sum g * y per channel, g = dx masked by
    (y * scale + shift > 0) -- compared against fp64 sums over the same fp16 tensors."""
    s = (1, 1, 1)
    x, w = make_conv_case(seed, in_shape, Co, k, s, p, d)
    geom = ops.ConvGeom(in_shape, Co, k, s, p, d)
    g = torch.Generator().manual_seed(seed + 2)
    dy = torch.randn(geom.out_shape, generator=g).to(ACT).float()
    r = torch.randn(in_shape, generator=g).to(ACT).float() if resid else None
    ybn = torch.randn(in_shape, generator=g).to(ACT).float()                 # raw output of the producer convolution
    sc = torch.rand(in_shape[1], generator=g) + 0.5
    sh = torch.randn(in_shape[1], generator=g) * 0.3
    _, wd = ops.prep_weights(w.to(device), geom)
    dyc, rc = host_to_cl(dy, device), (host_to_cl(r, device) if resid else None)
    base = ops.conv_dgrad(dyc, wd, geom, resid=rc)
    dx, part = ops.conv_dgrad(dyc, wd, geom, resid=rc, bn=(host_to_cl(ybn, device), sc.to(device), sh.to(device)))
    assert part is not None and part.shape[1:] == (2, in_shape[1])
    assert torch.equal(cl_to_host(dx), cl_to_host(base)), "the fused epilogue must not change the stored gradient"
    dxh = cl_to_host(dx).double()
    mask = (ybn * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1) > 0).double()

    def close(tot, m):
        ref_g, ref_gy = (dxh * m).sum((0, 2, 3, 4)), (dxh * m * ybn.double()).sum((0, 2, 3, 4))
        assert float((tot[0] - ref_g).abs().max()) <= 2e-6 * float((dxh * m).abs().sum((0, 2, 3, 4)).max())
        assert float((tot[1] - ref_gy).abs().max()) <= 2e-6 * float((dxh * m * ybn.double()).abs().sum((0, 2, 3, 4)).max())
    close(part.double().sum(0).cpu(), mask)
    # block-input form: the mask is a bit image (the previous block's output > 0)
    bmask = torch.rand(in_shape, generator=g) < 0.6
    Mrows = in_shape[0] * in_shape[2] * in_shape[3] * in_shape[4]
    bl = bmask.permute(0, 2, 3, 4, 1).reshape(Mrows, in_shape[1] // 8, 8).to(torch.int32)
    bits = (bl << torch.arange(8, dtype=torch.int32)).sum(-1).to(torch.uint8).contiguous().to(device)
    dx2, p0 = ops.conv_dgrad(dyc, wd, geom, resid=rc, bn={"bits": bits, "y0": host_to_cl(ybn, device)})
    assert torch.equal(cl_to_host(dx2), cl_to_host(base)) and p0 is not None
    close(p0.double().sum(0).cpu(), bmask.double())
    return part.shape[0]


def check_conv_wgrad(device, in_shape, Co, k, s, p, d=(1, 1, 1), Cw=None, affine=False, out_scale=1.0, seed=0):
    x, w = make_conv_case(seed, in_shape, Co, k, s, p, d, Cw)
    geom = ops.ConvGeom(in_shape, Co, k, s, p, d, Cw=Cw)
    g = torch.Generator().manual_seed(seed + 3)
    dy = torch.randn(geom.out_shape, generator=g).to(ACT).float()
    xin = x
    in_affine = None
    if affine:
        sc = torch.randn(in_shape[1], generator=g)
        sh = torch.randn(in_shape[1], generator=g) * 0.5
        xin = F.relu(x * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)).to(ACT).float()
        in_affine = (sc.to(device), sh.to(device), True)
    wr = w.clone().requires_grad_(True)
    F.conv3d(xin[:, : w.shape[1]], wr, None, s, p, d).backward(dy)
    ref = wr.grad * out_scale
    dw = torch.full(w.shape, 7.0, dtype=torch.float32, device=device)  # must be cleared by zero_first
    ops.conv_wgrad(host_to_cl(x, device), host_to_cl(dy, device), geom, dw, in_affine=in_affine, out_scale=out_scale,
                   zero_first=True)
    return assert_close("conv_wgrad", dw.cpu(), ref, 1e-4)


def check_bn_chain(device, shape, relu=True, residual=None, seed=0):
    """conv-stat partials -> finalize -> bn_act, and the backward (reduce/finalize/apply)."""
    g = torch.Generator().manual_seed(seed)
    N, C, T, H, W = shape
    y = (torch.randn(shape, generator=g) * 1.5 + 0.3).to(ACT).float()
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.2
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    dz = torch.randn(shape, generator=g).to(ACT).float()
    M = N * T * H * W
    # reference
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    zr = F.batch_norm(yr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    res = None
    if residual is not None:
        res = torch.randn(shape, generator=g).to(ACT).float()
        zr = zr + res
    if relu:
        zr = F.relu(zr)
    zr.backward(dz)
    # ours: partial sums as the conv epilogue would leave them (2 tiles)
    yf = y.permute(0, 2, 3, 4, 1).reshape(M, C)
    half = M // 2
    part = torch.stack([torch.stack([yf[:half].sum(0), (yf[:half] ** 2).sum(0)]),
                        torch.stack([yf[half:].sum(0), (yf[half:] ** 2).sum(0)])]).to(device)
    gd, bd, rmd, rvd = gamma.to(device), beta.to(device), rm.clone().to(device), rv.clone().to(device)
    scale, shift, mean, rstd = ops.bn_finalize(part, M, gd, bd, rmd, rvd, 0.1, 1e-5, training=True)
    assert_close("running_mean", rmd.cpu(), rm_ref, 1e-5)
    assert_close("running_var", rvd.cpu(), rv_ref, 1e-5)
    yc = host_to_cl(y, device)
    rc = host_to_cl(res, device) if res is not None else None
    z, bits = ops.bn_act(yc, scale, shift, relu=relu, resid=rc, want_mask=True)
    assert_close("bn_act", cl_to_host(z), zr.detach(), 2 * F16_EPS)
    # the 1-bit mask is exactly the sign of the stored activation
    zrow = z.permute(0, 2, 3, 4, 1).reshape(M, C).cpu()
    ref_bits = ((zrow > 0).reshape(M, C // 8, 8).to(torch.int32) << torch.arange(8, dtype=torch.int32)).sum(-1)
    assert torch.equal(bits.cpu().to(torch.int32), ref_bits), "bn_act bit mask"
    dgamma = torch.empty(C, device=device)
    dbeta = torch.empty(C, device=device)
    dzc = host_to_cl(dz, device)
    if residual is not None:
        dy, gm = ops.bn_bwd(dzc, yc, gd, mean, rstd, dgamma, dbeta, zmask=z if relu else None, want_g=True,
                            inv_loss_scale=0.5)
        gref = dz * (zr.detach() > 0) if relu else dz
        assert_close("bn_bwd g", cl_to_host(gm), gref, 1e-6)
        if relu:        # the bit-mask form must give the same gradients bit for bit
            dg2, db2 = torch.empty(C, device=device), torch.empty(C, device=device)
            dy2 = ops.bn_bwd(dzc, yc, gd, mean, rstd, dg2, db2, zmask=bits, inv_loss_scale=0.5)
            assert torch.equal(dy2.cpu(), dy.cpu()) and torch.equal(dg2.cpu(), dgamma.cpu()) and torch.equal(db2.cpu(), dbeta.cpu())
    else:
        dy = ops.bn_bwd(dzc, yc, gd, mean, rstd, dgamma, dbeta, relu_affine=(scale, shift) if relu else None,
                        inv_loss_scale=0.5)
    assert_close("bn_bwd dy", cl_to_host(dy), yr.grad, 3 * F16_EPS)
    assert_close("bn_bwd dgamma", dgamma.cpu(), gr.grad * 0.5, 2e-3 * EPS_SCALE)
    assert_close("bn_bwd dbeta", dbeta.cpu(), br.grad * 0.5, 2e-3 * EPS_SCALE)
    # eval mode uses the running statistics
    sc_e, sh_e, _, _ = ops.bn_finalize(None, M, gd, bd, rmd, rvd, 0.1, 1e-5, training=False)
    ze = ops.bn_act(yc, sc_e, sh_e, relu=False)
    zref = F.batch_norm(y, rm_ref, rv_ref, gamma, beta, False, 0.1, 1e-5)
    assert_close("bn eval", cl_to_host(ze), zref, 2 * F16_EPS)


def check_pool(device, shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    N, C, T, H, W = shape
    y = torch.randn(shape, generator=g).to(ACT).float()
    sc = torch.rand(C, generator=g) + 0.5
    sh = torch.randn(C, generator=g) * 0.3
    yr = y.clone()
    z = F.relu(yr * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)).to(ACT).float().requires_grad_(True)
    pr = F.max_pool3d(z, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    dout = torch.randn(pr.shape, generator=g).to(ACT).float()
    pr.backward(dout)
    gref = z.grad * (z.detach() > 0)
    yc = host_to_cl(y, device)
    aff = (sc.to(device), sh.to(device), True)
    out, arg = ops.pool_fwd(yc, (3, 3), (2, 2), (1, 1), affine=aff)
    assert_close("pool_fwd", cl_to_host(out), pr.detach(), 2 * F16_EPS)  # fused multiply-add vs mul+add before fp16 rounding
    gm = ops.pool_bwd(tuple(yc.shape), out, arg, host_to_cl(dout, device), (3, 3), (2, 2), (1, 1), relu=True)
    assert_close("pool_bwd", cl_to_host(gm), gref, 2 * F16_EPS)


def check_head_mean(device, shape, seed=0):
    """heads._GlobalMeanFn (sf_tmean_fwd as [8][rows / 8] partial means + sf_tmean_bwd as a broadcast) against the AvgPool3d over
    the whole extent of the reference head (head_helper.py:293-300) on the same 16-bit operand: fp32 sums in another order."""
    from slowfast_amd import heads
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g).to(ACT).float()
    dout = torch.randn((shape[0], shape[1], 1, 1, 1), generator=g)
    xr = x.clone().requires_grad_(True)
    ref = F.avg_pool3d(xr, shape[2:], stride=1)
    ref.backward(dout)
    xc = host_to_cl(x, device).requires_grad_(True)
    assert heads._GlobalMeanFn.eligible(xc)
    out = heads._GlobalMeanFn.apply(xc)
    assert out.dtype == torch.float32 and tuple(out.shape) == tuple(ref.shape)
    out.backward(dout.to(device))
    assert_close("head mean", out.detach().cpu(), ref.detach(), 1e-6)
    assert ops.is_cl(xc.grad)
    assert_close("head mean dx", cl_to_host(xc.grad), xr.grad.to(ACT).float(), F16_EPS)  # dout * (1 / rows) against dout / rows, rounded once


def check_layout(device, shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g)
    xc = ops.ncthw_to_cl(x.to(device))
    assert xc.shape[1] == (shape[1] + 7) // 8 * 8
    assert_close("ncthw_to_cl", cl_to_host(xc)[:, : shape[1]], x.to(ACT).float(), 1e-6)
    assert float(cl_to_host(xc)[:, shape[1]:].abs().max()) == 0.0 if xc.shape[1] > shape[1] else True
    back = ops.cl_to_ncthw(xc)
    assert_close("cl_to_ncthw", back.cpu()[:, : shape[1]], x.to(ACT).float(), 1e-6)
    if shape[1] <= 4 and shape[4] % 2 == 0:
        # the W-pair-folded stem operand (Cp = 4): logical channel (w & 1) * 4 + c of column w / 2, bit-exact against the rounding
        xw = cl_to_host(ops.ncthw_to_cl_wpairs(x.to(device)))                        # [N, 8, T, H, W / 2]
        N, C, T, H, W = shape
        ref = torch.zeros(N, 8, T, H, W // 2)
        for par in range(2):
            ref[:, par * 4: par * 4 + C] = x.to(ACT).float()[..., par::2]
        assert torch.equal(xw, ref), "ncthw_to_cl_wpairs"


def check_bn_finalize_long(device, nblk, C, seed=0):
    """Long partial tables: 1024-thread finalize blocks up to 16384 rows, the in-place fold stage (sf_part_fold_kernel)
    beyond, in the forward (sf_bn_finalize) and the backward (sf_bn_bwd_finalize) statistics."""
    from slowfast_amd.lib import get_lib
    g = torch.Generator().manual_seed(seed)
    rows = 128
    s = torch.randn((nblk, C), generator=g) * rows ** 0.5 + 0.3 * rows
    q = torch.rand((nblk, C), generator=g) * rows + rows
    part = torch.stack([s, q], 1).contiguous()
    count = float(nblk * rows)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    mean = s.double().sum(0) / count
    var = q.double().sum(0) / count - mean * mean
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    scale, shift, m, r = ops.bn_finalize(part.clone().to(device), count, gamma.to(device), beta.to(device), None, None,
                                         0.1, 1e-5, training=True)
    assert_close("finalize mean", m.cpu(), mean.float(), 1e-6)
    assert_close("finalize rstd", r.cpu(), rstd.float(), 1e-5)
    assert_close("finalize scale", scale.cpu(), (gamma.double() * rstd).float(), 1e-5)
    assert_close("finalize shift", shift.cpu(), (beta.double() - mean * gamma.double() * rstd).float(), 1e-5)
    # backward sums
    dgamma, dbeta = torch.empty(C, device=device), torch.empty(C, device=device)
    coef = torch.empty((3, C), device=device)
    pd = part.clone().to(device)
    stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else None
    get_lib().call("sf_bn_bwd_finalize", pd.data_ptr(), nblk, C, C, count, gamma.to(device).data_ptr(), m.data_ptr(),
                   r.data_ptr(), 1.0, dgamma.data_ptr(), dbeta.data_ptr(), 0, coef.data_ptr(), stream)
    sg, sgy = s.double().sum(0), q.double().sum(0)
    assert_close("bwd dbeta", dbeta.cpu(), sg.float(), 1e-6)
    assert_close("bwd dgamma", dgamma.cpu(), (rstd * (sgy - mean * sg)).float(), 1e-5)


def check_roi_pool(device, shape=(2, 16, 3, 10, 12), aligned=True, seed=0):
    """AvgPool3d([T,1,1]) -> ROIAlign(7x7, 1/16, adaptive sampling) -> MaxPool2d(7) forward and backward
    (sf_tmean_*, sf_roi_align_max_*) against the oracle's ROIAlign restatement, incl. boxes that leave the map."""
    from oracle import video_ref
    from slowfast_amd.heads import _RoiPoolFn
    g = torch.Generator().manual_seed(seed)
    N, C, T, H, W = shape
    x = torch.randn(shape, generator=g).to(ACT).float()
    S = 16.0 * max(H, W)
    rois = torch.tensor([[0, 0.06 * S, 0.12 * S, 0.8 * S, 0.7 * S], [N - 1, 0., 0., 16. * W - 1, 16. * H - 1],
                         [N - 1, 0.3 * S, 0.2 * S, 0.45 * S, 0.36 * S], [0, -0.1 * S, 0.15 * S, 0.4 * S, 1.5 * S],
                         [0, 0.5 * S, 0.5 * S, 0.5 * S + 3.0, 0.5 * S + 2.0]])
    xr = x.clone().requires_grad_(True)
    ref = video_ref.roi_align(xr.mean(2), rois, (7, 7), 1 / 16., 0, aligned).amax((2, 3))
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    xc = host_to_cl(x, device).requires_grad_(True)
    out = _RoiPoolFn.apply(xc, rois.to(device), 7, 1 / 16., aligned)
    (out * w.to(device)).sum().backward()
    assert_close("roi pooled", out.cpu(), ref.detach(), 1e-5)
    assert_close("roi d(features)", cl_to_host(xc.grad), xr.grad, 2 * F16_EPS)


def check_roi_known_answer(device):
    """sf_roi_align_max_fwd against detectron2's PUBLISHED ROIAlign vectors (tests/golden/roi_align_detectron2.json:
    detectron2 tests/layers/test_roi_align.py, arange 5x5 map, box [1, 1, 3, 3], 4x4 output, aligned False / True) -- the
    pin for the op the reference imports from detectron2 (head_helper.py:11, 88-94).  The kernel fuses the max-pool over the
    bins, so (a) the 4x4 call must return the table's maximum for +ramp and minus its minimum for -ramp in both modes, and
    (b) every one of the 32 published bin values is read back through a 1x1 call on the box of that bin (aligned mode, box
    = the bin's sample cell shifted by +0.5: same single bilinear sample as the published 4x4 call takes for that bin).
    Also test_empty_box (zero-height box -> zeros)."""
    import json
    import os
    from slowfast_amd.heads import _RoiPoolFn
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "roi_align_detectron2.json")))
    f = gold["forward_output"]
    H, W = f["input_hw"]
    ramp = torch.arange(H * W, dtype=torch.float32).reshape(H, W)
    x = torch.zeros((1, 8, 1, H, W))
    x[0, 0, 0], x[0, 1, 0], x[0, 2, 0] = ramp, -ramp, 2.0 * ramp
    xc = host_to_cl(x, device)
    box = torch.tensor([f["box"]], dtype=torch.float32)
    for aligned, key in ((False, "aligned_false"), (True, "aligned_true")):
        tab = torch.tensor(f[key])
        out = _RoiPoolFn.apply(xc, box.to(device), f["output_size"][0], f["spatial_scale"], aligned).cpu()
        assert abs(float(out[0, 0]) - float(tab.max())) < 1e-6, (key, out[0, :3], tab.max())
        assert abs(float(out[0, 1]) + float(tab.min())) < 1e-6, (key, out[0, :3], tab.min())
        assert abs(float(out[0, 2]) - 2.0 * float(tab.max())) < 1e-6, (key, out[0, :3])
        # every published bin through a 1x1 aligned call on that bin's cell
        x1, y1 = float(f["box"][1]) - (0.5 if aligned else 0.0), float(f["box"][2]) - (0.5 if aligned else 0.0)
        bw = (f["box"][3] - f["box"][1]) / f["output_size"][1]
        bh = (f["box"][4] - f["box"][2]) / f["output_size"][0]
        cells = [[0.0, x1 + j * bw + 0.5, y1 + i * bh + 0.5, x1 + (j + 1) * bw + 0.5, y1 + (i + 1) * bh + 0.5]
                 for i in range(f["output_size"][0]) for j in range(f["output_size"][1])]
        out = _RoiPoolFn.apply(xc, torch.tensor(cells).to(device), 1, 1.0, True).cpu()
        assert_close("published bins " + key, out[:, 0], tab.reshape(-1), 1e-6)
    e = gold["empty_box"]
    g = torch.Generator().manual_seed(0)
    xr = torch.zeros((1, 8, 1, 5, 5))
    xr[0, :, 0] = torch.rand((8, 5, 5), generator=g).to(ACT).float() + 0.5
    out = _RoiPoolFn.apply(host_to_cl(xr, device), torch.tensor([e["box"]], dtype=torch.float32).to(device),
                           e["output_size"][0], 1.0, e["aligned"]).cpu()
    assert float(out.abs().max()) == e["expected_all"], out


def check_pack_clip(device, arch="slowfast", reverse=False, seed=0):
    """sf_pack_clip_u8 (uint8 frames -> normalised fp16 W-pair clips per pathway) is bit-exact with the fp16 rounding
    of the reference's tensor_normalize + permute + pack_pathway_output (oracle/data_ref.py)."""
    import slowfast_amd as sa
    from oracle import data_ref
    cfg = sa.get_preset("SLOWFAST_8x8_R50" if arch == "slowfast" else "C2D_8x8_R50",
                        ["DATA.MEAN", [0.45, 0.40, 0.35], "DATA.STD", [0.225, 0.25, 0.2],
                         "DATA.REVERSE_INPUT_CHANNEL", reverse])
    g = torch.Generator().manual_seed(seed)
    frames = torch.randint(0, 256, (2, 8, 6, 10, 3), generator=g, dtype=torch.int64).to(torch.uint8)
    ref = data_ref.pack_pathways(frames, cfg)
    got = sa.pack_pathways_u8(frames.to(device), cfg)
    assert len(got) == len(ref)
    for x, r in zip(got, ref):
        N, C8, T, H, W2 = x.shape
        assert C8 == 8 and getattr(x, "_sf_wpairs", False)
        # (N, 8, T, H, W/2) view of the N,T,H,W,4 buffer: channel = (w & 1) * 4 + c
        buf = x.permute(0, 2, 3, 4, 1).reshape(N, T, H, W2 * 2, 4).cpu()
        assert torch.equal(buf[..., :3].permute(0, 4, 1, 2, 3).contiguous(), r.to(ACT)), "normalised clip differs"
        assert float(buf[..., 3].abs().max()) == 0.0
    # out=: the next batch written straight into the tensors of a previous call (a captured step's static input buffers)
    frames2 = torch.randint(0, 256, (2, 8, 6, 10, 3), generator=g, dtype=torch.int64).to(torch.uint8)
    ptrs = [x.data_ptr() for x in got]
    again = sa.pack_pathways_u8(frames2.to(device), cfg, out=got)
    fresh = sa.pack_pathways_u8(frames2.to(device), cfg)
    assert [x.data_ptr() for x in again] == ptrs and all(torch.equal(a, f) for a, f in zip(again, fresh))


def check_prep_weights_batch(device, cases, seed=0):
    """sf_prep_weights_batch (every weight of a model packed by one launch, LDS bricks) == sf_prep_weights per layer, bit for
    bit, including the zero padding of both operands.  cases: (Co_real, Cw, Ci_padded, kernel, need_dgrad)."""
    from ctypes import byref
    import numpy as np
    from slowfast_amd.lib import PrepItem, get_lib
    lib = get_lib()
    g = torch.Generator().manual_seed(seed)
    items = (PrepItem * len(cases))()
    blk_item, blk_off, keep, want = [], [], [], []
    for i, (Co, Cw, Ci, k, need_dgrad) in enumerate(cases):
        geom = ops.ConvGeom((1, Ci, 4, 8, 8), Co, k, 1, tuple(x // 2 for x in k), Cw=Cw)
        w = torch.randn((Co, Cw) + tuple(k), generator=g).to(device)
        want.append(ops.prep_weights(w, geom, need_dgrad=need_dgrad))
        wf = torch.full((geom.Co, geom.ldf), float("nan"), dtype=ACT, device=device)
        wd = torch.full((geom.Ci, geom.ldd), float("nan"), dtype=ACT, device=device) if need_dgrad else None
        lib.call("sf_prep_item_fill", byref(geom.desc(geom.Ci, geom.Co)), w.data_ptr(), wf.data_ptr(),
                 None if wd is None else wd.data_ptr(), byref(items[i]))
        nb = lib.call("sf_prep_item_blocks", byref(items[i]))
        blk_item += [i] * nb
        blk_off += list(range(nb))
        keep.append((w, wf, wd))
    tab = torch.from_numpy(np.frombuffer(bytes(items), dtype=np.uint8).copy()).to(device)
    bi = torch.tensor(blk_item, dtype=torch.int32, device=device)
    bo = torch.tensor(blk_off, dtype=torch.int32, device=device)
    lib.call("sf_prep_weights_batch", tab.data_ptr(), bi.data_ptr(), bo.data_ptr(), len(blk_item), ops._stream(tab))
    for (w, wf, wd), (rf, rd), case in zip(keep, want, cases):
        assert torch.equal(wf.cpu(), rf.cpu()), ("forward operand", case)
        if wd is not None:
            assert torch.equal(wd.cpu(), rd.cpu()), ("dgrad operand", case)
