"""Kernel parity checks of the token-space entry points (tokens.py) against the torch fp32 ops the reference
calls (F.linear, F.layer_norm, F.gelu, F.conv3d with groups, F.max_pool3d, softmax, matmul) on fp16-rounded data."""
import torch
import torch.nn.functional as F

from oracle import mvit_ref
from slowfast_amd import tokens
from tests.kernel_checks import ACT, EPS_SCALE, F16_EPS, assert_close


def _h(x, device):
    return x.to(ACT).to(device)


def check_gemm(device, M, K, N, bias=True, resid=True, seed=0):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn((M, K), generator=g).to(ACT).float()
    w = (torch.randn((N, K), generator=g) / K ** 0.5).to(ACT).float()
    b = torch.randn(N, generator=g) if bias else None
    r = torch.randn((M, N), generator=g).to(ACT).float() if resid else None
    ref = F.linear(a, w, b) + (r if resid else 0)
    out = tokens.gemm(_h(a, device), _h(w, device), bias=b.to(device) if bias else None,
                      resid=_h(r, device) if resid else None)
    assert_close("gemm", out.float().cpu(), ref, 2 * F16_EPS)
    # weight / bias gradients
    dy = torch.randn((M, N), generator=g).to(ACT).float()
    dw = torch.full((N, K), 3.0, device=device)
    tokens.linear_wgrad(_h(a, device), _h(dy, device), dw, zero_first=True)
    assert_close("linear_wgrad", dw.cpu(), dy.t() @ a, 1e-4)
    db = torch.empty(N, device=device)
    tokens.bias_grad(_h(dy, device), db)
    assert_close("bias_grad", db.cpu(), dy.sum(0), 1e-4)


def check_layernorm(device, M, C, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn((M, C), generator=g) * 1.3 + 0.2).to(ACT).float()
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    dy = torch.randn((M, C), generator=g).to(ACT).float()
    res = torch.randn((M, C), generator=g).to(ACT).float()
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = F.layer_norm(xr, (C,), gr, br, 1e-6)
    y.backward(dy)
    yk, mean, rstd = tokens.layernorm_fwd(_h(x, device), gamma.to(device), beta.to(device), 1e-6)
    assert_close("ln fwd", yk.float().cpu(), y.detach(), 2 * F16_EPS)
    dg, db = torch.empty(C, device=device), torch.empty(C, device=device)
    dx = tokens.layernorm_bwd(_h(dy, device), _h(x, device), gamma.to(device), mean, rstd, dg, db, resid=_h(res, device))
    assert_close("ln dx", dx.float().cpu(), xr.grad + res, 3 * F16_EPS)
    assert_close("ln dgamma", dg.cpu(), gr.grad, 1e-3 * EPS_SCALE)
    assert_close("ln dbeta", db.cpu(), br.grad, 1e-3 * EPS_SCALE)
    # the same pass with the column sums of the residual operand and of the stored result (bias gradients of the neighbouring
    # Linear layers): same dx bit for bit, sums of exactly the 16-bit values that were read / written
    dg2, db2 = torch.empty(C, device=device), torch.empty(C, device=device)
    sr, sx = torch.full((C,), 7.0, device=device), torch.full((C,), 3.0, device=device)
    dx2 = tokens.layernorm_bwd(_h(dy, device), _h(x, device), gamma.to(device), mean, rstd, dg2, db2, resid=_h(res, device),
                               sums=((sr, False), (sx, True)))
    assert torch.equal(dx2, dx) and torch.equal(dg2, dg) and torch.equal(db2, db)
    assert_close("ln sum resid", sr.cpu(), res.sum(0), 1e-4 * EPS_SCALE)
    assert_close("ln sum dx (accumulated onto 3)", sx.cpu(), dx.float().cpu().sum(0) + 3.0, 1e-4 * EPS_SCALE)


def check_gelu(device, n, seed=0):
    g = torch.Generator().manual_seed(seed)
    h = (torch.randn(n, generator=g) * 2).to(ACT).float()
    da = torch.randn(n, generator=g).to(ACT).float()
    hr = h.clone().requires_grad_(True)
    a = F.gelu(hr)
    a.backward(da)
    assert_close("gelu fwd", tokens.gelu_fwd(_h(h, device)).float().cpu(), a.detach(), 2 * F16_EPS)
    assert_close("gelu bwd", tokens.gelu_bwd(_h(h, device), _h(da, device)).float().cpu(), hr.grad, 2 * F16_EPS)


def check_dwconv(device, B, heads, Cw, thw, kernel, stride, cls, seed=0):
    """Depthwise conv over tokens [B, cls+THW, 3*heads*Cw] (middle slice), weight shared by the heads."""
    g = torch.Generator().manual_seed(seed)
    T, H, W = thw
    C = heads * Cw
    N = cls + T * H * W
    big = torch.randn((B, N, 3 * C), generator=g).to(ACT).float()
    x = big[..., C:2 * C]
    w = (torch.randn((Cw, 1) + tuple(kernel), generator=g) * 0.3)
    pad = tuple(k // 2 for k in kernel)
    # reference: attention_pool on (B, heads, N, Cw)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    t = xr.reshape(B, N, heads, Cw).permute(0, 2, 1, 3)
    sd = {"w": wr}
    y, thw_o = mvit_ref.attention_pool(t, wr, stride, thw, bool(cls))
    y = y.permute(0, 2, 1, 3).reshape(B, -1, C)
    dy = torch.randn(y.shape, generator=g).to(ACT).float()
    y.backward(dy)
    geom = tokens.DwGeom(B, C, Cw, thw, kernel, stride, pad, cls)
    assert list(geom.out_thw) == list(thw_o)
    bigd = _h(big, device)
    xd = bigd[..., C:2 * C]
    yk = tokens.dwconv_fwd(xd, w.to(device), geom).view(B, -1, C)
    assert_close("dwconv fwd", yk.float().cpu(), y.detach(), 2 * F16_EPS)
    dbig = torch.zeros((B, N, 3 * C), dtype=ACT, device=device)
    tokens.dwconv_dgrad(_h(dy, device).view(-1, C), w.to(device), geom, out=dbig[..., C:2 * C])
    assert_close("dwconv dgrad", dbig[..., C:2 * C].float().cpu(), xr.grad, 2 * F16_EPS)
    assert float(dbig[..., :C].abs().max()) == 0.0 and float(dbig[..., 2 * C:].abs().max()) == 0.0
    # ... and the column sums of dx the plane sweeps leave as a by-product (the qkv bias gradient of MViT), cls row included
    dbig2 = torch.zeros((B, N, 3 * C), dtype=ACT, device=device)
    _, spart = tokens.dwconv_dgrad(_h(dy, device).view(-1, C), w.to(device), geom, out=dbig2[..., C:2 * C], sums=True)
    assert torch.equal(dbig2, dbig)
    if spart is not None:
        assert_close("dwconv dgrad column sums", spart[:, 0].sum(0).cpu(), xr.grad.reshape(-1, C).sum(0), 2e-3 * EPS_SCALE)
    dw = torch.full(w.shape, 5.0, device=device)
    tokens.dwconv_wgrad(xd, _h(dy, device).view(-1, C), geom, dw, zero_first=True)
    assert_close("dwconv wgrad", dw.cpu(), wr.grad, 1e-3 * EPS_SCALE)
    # BatchNorm statistics epilogue (X3D)
    y2, part = tokens.dwconv_fwd(xd, w.to(device), geom, stats=True)
    tot = part.sum(0).cpu()
    yf = y.detach().reshape(-1, C)
    assert_close("dwconv stats sum", tot[0], yf.sum(0), 2e-3 * EPS_SCALE)
    assert_close("dwconv stats sumsq", tot[1], (yf * yf).sum(0), 2e-3 * EPS_SCALE)


def check_dwconv_pair(device, B, heads, Cw, thw, stride, cls, seed=0):
    """PAIR forms (tokens.dwconv_*_pair: the k and v slices of a [B, N, 3C] tensor in one launch per direction) against two single
    calls: the same arithmetic, so every result must be EQUAL bit for bit (outputs, the two input-gradient slices, both weight
    gradients with and without accumulation)."""
    g = torch.Generator().manual_seed(seed)
    T, H, W = thw
    C = heads * Cw
    N = cls + T * H * W
    kernel, pad = (3, 3, 3), (1, 1, 1)
    geom = tokens.DwGeom(B, C, Cw, thw, kernel, stride, pad, cls)
    big = _h(torch.randn((B, N, 3 * C), generator=g).to(ACT).float(), device)
    k_in, v_in = big[..., C:2 * C], big[..., 2 * C:3 * C]
    wk = (torch.randn((Cw, 1) + kernel, generator=g) * 0.3).to(device)
    wv = (torch.randn((Cw, 1) + kernel, generator=g) * 0.3).to(device)
    assert tokens.dwconv_pair_ok(geom, 3 * C, C)
    yk, yv = tokens.dwconv_fwd_pair(k_in, v_in, wk, wv, geom)
    assert torch.equal(yk, tokens.dwconv_fwd(k_in, wk, geom)) and torch.equal(yv, tokens.dwconv_fwd(v_in, wv, geom))
    dk = _h(torch.randn((geom.rows_out, C), generator=g).to(ACT).float(), device)
    dv = _h(torch.randn((geom.rows_out, C), generator=g).to(ACT).float(), device)
    d1 = torch.zeros((B, N, 3 * C), dtype=ACT, device=device)
    d2 = torch.zeros((B, N, 3 * C), dtype=ACT, device=device)
    tokens.dwconv_dgrad_pair(dk, dv, wk, wv, geom, out=d1[..., C:2 * C], out2=d1[..., 2 * C:3 * C])
    tokens.dwconv_dgrad(dk, wk, geom, out=d2[..., C:2 * C])
    tokens.dwconv_dgrad(dv, wv, geom, out=d2[..., 2 * C:3 * C])
    assert torch.equal(d1, d2)
    for zero_first in (True, False):
        gk1, gv1 = torch.full(wk.shape, 2.0, device=device), torch.full(wv.shape, -3.0, device=device)
        gk2, gv2 = gk1.clone(), gv1.clone()
        tokens.dwconv_wgrad_pair(k_in, v_in, dk, dv, geom, gk1, gv1, zero_first=zero_first, zero_first2=not zero_first)
        tokens.dwconv_wgrad(k_in, dk, geom, gk2, zero_first=zero_first)
        tokens.dwconv_wgrad(v_in, dv, geom, gv2, zero_first=not zero_first)
        assert torch.equal(gk1, gk2) and torch.equal(gv1, gv2)


def check_token_pool(device, B, C, thw, stride, seed=0):
    g = torch.Generator().manual_seed(seed)
    T, H, W = thw
    x = torch.randn((B, 1 + T * H * W, C), generator=g).to(ACT).float()
    xr = x.clone().requires_grad_(True)
    y, thw_o = mvit_ref.attention_pool(xr, None, stride, thw, True, pool_mode="max")
    dy = torch.randn(y.shape, generator=g).to(ACT).float()
    y.backward(dy)
    k = [s + 1 if s > 1 else s for s in stride]
    p = [v // 2 for v in k]
    out, arg, thw_k = tokens.token_pool_fwd(_h(x, device), B, thw, k, stride, p, cls=True)
    assert list(thw_k) == list(thw_o)
    assert_close("token pool fwd", out.float().cpu(), y.detach(), 1e-6)
    gk = tokens.token_pool_bwd(_h(dy, device), out, arg, B, thw, k, stride, p, C, cls=True)
    assert_close("token pool bwd", gk.float().cpu(), xr.grad, 2 * F16_EPS)


def check_attention_core(device, B, heads, D, q_thw, k_thw, seed=0):
    """scores + rel-pos bias + softmax + P.V + residual pooling, forward and backward, against the reference math."""
    from slowfast_amd.mvit_engine import _rel_index
    g = torch.Generator().manual_seed(seed)
    C = heads * D
    Nq, Nk = 1 + q_thw[0] * q_thw[1] * q_thw[2], 1 + k_thw[0] * k_thw[1] * k_thw[2]
    q = torch.randn((B, Nq, C), generator=g).to(ACT).float()
    k = torch.randn((B, Nk, C), generator=g).to(ACT).float()
    v = torch.randn((B, Nk, C), generator=g).to(ACT).float()
    rows = (2 * max(q_thw[1], k_thw[1]) - 1, 2 * max(q_thw[2], k_thw[2]) - 1, 2 * max(q_thw[0], k_thw[0]) - 1)
    tabs = [torch.randn((r, D), generator=g) * 0.3 for r in rows]
    do = torch.randn((B, Nq, C), generator=g).to(ACT).float()
    scale = D ** -0.5
    # reference
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    tr = [t.clone().requires_grad_(True) for t in tabs]
    qh, kh, vh = (t.reshape(B, -1, heads, D).permute(0, 2, 1, 3) for t in (qr, kr, vr))
    attn = (qh * scale) @ kh.transpose(-2, -1)
    attn = mvit_ref._rel_pos_bias(attn, qh, True, q_thw, k_thw, tr[0], tr[1], tr[2]).softmax(-1)
    o = attn @ vh
    o = torch.cat([o[:, :, :1], o[:, :, 1:] + qh[:, :, 1:]], 2).transpose(1, 2).reshape(B, Nq, C)
    o.backward(do)
    # kernels
    d = tokens.attn_desc(B, heads, D, True, q_thw, k_thw, *rows)
    idx = (_rel_index(q_thw[1], k_thw[1], device), _rel_index(q_thw[2], k_thw[2], device), _rel_index(q_thw[0], k_thw[0], device))
    qd, kd, vd, tabd = _h(q, device), _h(k, device), _h(v, device), [t.to(device) for t in tabs]
    lds = (Nk + 7) // 8 * 8
    rq = tokens.relpos_fwd(d, qd, tabd, idx)
    S = torch.empty((B, heads, Nq, lds), dtype=ACT, device=device)
    tokens.bgemm_heads(qd, (Nq * C, D), Nq, D, C, kd, (Nk * C, D), Nk, C, S, (heads * Nq * lds, Nq * lds), lds, B, heads)
    P = tokens.softmax_fwd(d, S, scale, rq)
    assert_close("softmax probabilities", P[..., :Nk].float().cpu(), attn.detach(), 4 * F16_EPS)
    assert float(P[..., Nk:].abs().max()) == 0.0 if lds > Nk else True
    vt = tokens.transpose_heads(vd, B, Nk, heads, D, lds)
    ok = torch.empty((B, Nq, C), dtype=ACT, device=device)
    tokens.bgemm_heads(P, (heads * Nq * lds, Nq * lds), Nq, lds, lds, vt, (heads * D * lds, D * lds), D, lds,
                       ok, (Nq * C, D), C, B, heads, resid=qd, r_strides=(Nq * C, D), ldr=C, resid_row0=1)
    assert_close("attention out", ok.float().cpu(), o.detach(), 3 * F16_EPS)
    dod = _h(do, device)
    dP = torch.empty((B, heads, Nq, lds), dtype=ACT, device=device)
    tokens.bgemm_heads(dod, (Nq * C, D), Nq, D, C, vd, (Nk * C, D), Nk, C, dP, (heads * Nq * lds, Nq * lds), lds, B, heads)
    dv = torch.empty((B, Nk, C), dtype=ACT, device=device)
    tokens.bgemm_tn_heads(P, (heads * Nq * lds, Nq * lds), lds, dod, (Nq * C, D), C, Nq, Nk, D, dv, (Nk * C, D), C, B, heads)
    assert_close("dV", dv.float().cpu(), vr.grad, 3 * F16_EPS)
    dS, drq = tokens.softmax_bwd(d, dP, P, scale, want_drq=True)
    kt = tokens.transpose_heads(kd, B, Nk, heads, D, lds)
    dq = torch.empty((B, Nq, C), dtype=ACT, device=device)
    tokens.bgemm_heads(dS, (heads * Nq * lds, Nq * lds), Nq, lds, lds, kt, (heads * D * lds, D * lds), D, lds,
                       dq, (Nq * C, D), C, B, heads, resid=dod, r_strides=(Nq * C, D), ldr=C, resid_row0=1)
    dk = torch.empty((B, Nk, C), dtype=ACT, device=device)
    tokens.bgemm_tn_heads(dS, (heads * Nq * lds, Nq * lds), lds, qd, (Nq * C, D), C, Nq, Nk, D, dk, (Nk * C, D), C, B, heads)
    dts = [torch.full(t.shape, 2.0, device=device) for t in tabs]
    tokens.relpos_bwd(d, qd, tabd, idx, drq, dq, dts, [False, False, False])
    assert_close("dK", dk.float().cpu(), kr.grad, 6 * F16_EPS)
    assert_close("dQ", dq.float().cpu(), qr.grad, 6 * F16_EPS)
    for name, got, ref in zip(("d rel_pos_h", "d rel_pos_w", "d rel_pos_t"), dts, tr):
        assert_close(name, got.cpu(), ref.grad, 5e-3 * EPS_SCALE)


def check_attention_fused(device, B, heads, D, q_thw, k_thw, cls=True, rel=True, residual=True, seed=0):
    """sf_attn_fwd / sf_attn_bwd (flash-style, no score tensor) against the reference attention math: output,
    dQ / dK / dV and the rel-pos table gradients."""
    from slowfast_amd.mvit_engine import _rel_index
    g = torch.Generator().manual_seed(seed)
    C = heads * D
    c = int(cls)
    Nq, Nk = c + q_thw[0] * q_thw[1] * q_thw[2], c + k_thw[0] * k_thw[1] * k_thw[2]
    q = torch.randn((B, Nq, C), generator=g).to(ACT).float()
    k = torch.randn((B, Nk, C), generator=g).to(ACT).float()
    v = torch.randn((B, Nk, C), generator=g).to(ACT).float()
    rows = (2 * max(q_thw[1], k_thw[1]) - 1, 2 * max(q_thw[2], k_thw[2]) - 1, 2 * max(q_thw[0], k_thw[0]) - 1)
    tabs = [torch.randn((r, D), generator=g) * 0.3 for r in rows]
    do = torch.randn((B, Nq, C), generator=g).to(ACT).float()
    scale = D ** -0.5
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    tr = [t.clone().requires_grad_(True) for t in tabs]
    qh, kh, vh = (t.reshape(B, -1, heads, D).permute(0, 2, 1, 3) for t in (qr, kr, vr))
    attn = (qh * scale) @ kh.transpose(-2, -1)
    if rel:
        attn = mvit_ref._rel_pos_bias(attn, qh, cls, q_thw, k_thw, tr[0], tr[1], tr[2])
    attn = attn.softmax(-1)
    o = attn @ vh
    if residual:
        o = torch.cat([o[:, :, :c], o[:, :, c:] + qh[:, :, c:]], 2) if cls else o + qh
    o = o.transpose(1, 2).reshape(B, Nq, C)
    o.backward(do)
    d = tokens.attn_desc(B, heads, D, cls, q_thw, k_thw, *(rows if rel else (0, 0, 0)))
    qd, kd, vd, tabd, dod = _h(q, device), _h(k, device), _h(v, device), [t.to(device) for t in tabs], _h(do, device)
    rq = None
    idx = None
    if rel:
        idx = (_rel_index(q_thw[1], k_thw[1], device), _rel_index(q_thw[2], k_thw[2], device),
               _rel_index(q_thw[0], k_thw[0], device))
        rq = tokens.relpos_fwd(d, qd, tabd, idx)
    oh = tokens.attn_onehot(d, device) if rel else None
    of, lse = tokens.attn_fwd(d, qd, kd, vd, scale, rq, residual, onehot=oh)
    assert_close("fused attention out", of.float().cpu(), o.detach(), 3 * F16_EPS)
    dq, dk, dv, drq = tokens.attn_bwd(d, qd, kd, vd, scale, rq, residual, of, dod, lse, onehot=oh)
    assert_close("fused dV", dv.float().cpu(), vr.grad, 4 * F16_EPS)
    assert_close("fused dK", dk.float().cpu(), kr.grad, 6 * F16_EPS)
    if rel:
        dts = [torch.full(t.shape, 2.0, device=device) for t in tabs]
        tokens.relpos_bwd(d, qd, tabd, idx, drq, dq, dts, [False, False, False])
        for name, got, ref in zip(("d rel_pos_h", "d rel_pos_w", "d rel_pos_t"), dts, tr):
            assert_close("fused " + name, got.cpu(), ref.grad, 5e-3 * EPS_SCALE)
    assert_close("fused dQ", dq.float().cpu(), qr.grad, 6 * F16_EPS)


def check_gemm_gelu(device, M, K, N, seed=0):
    """sf_gemm_act: fc1 with the GELU in the epilogue (both outputs) and the fc2 data gradient times gelu'(h)."""
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn((M, K), generator=g) * 0.7).to(ACT).float()
    w = (torch.randn((N, K), generator=g) * (1.0 / K ** 0.5)).to(ACT).float()
    b = torch.randn(N, generator=g) * 0.2
    h_ref = a @ w.t() + b
    h, act = tokens.gemm_gelu(_h(a, device), _h(w, device), bias=b.to(device))
    assert_close("fc1 pre-activation", h.float().cpu(), h_ref, 2 * F16_EPS)
    assert_close("gelu(fc1)", act.float().cpu(), torch.nn.functional.gelu(h.float().cpu()), 2 * F16_EPS)
    dy = torch.randn((M, K), generator=g).to(ACT).float()       # gradient w.r.t. an [M, K] output of a Linear(N -> K)
    w2 = (torch.randn((K, N), generator=g) * (1.0 / N ** 0.5)).to(ACT).float()        # that Linear's weight [K, N]
    hh = h.float().cpu().requires_grad_(True)
    (torch.nn.functional.gelu(hh) @ w2.t()).backward(dy)
    dh = tokens.gemm_gelu_grad(_h(dy, device), _h(w2.t().contiguous(), device), h)
    assert_close("d(fc1 output)", dh.float().cpu(), hh.grad, 3 * F16_EPS)


def check_rows32(device, B, Ntok, K, N, period_full=False, bias=True, src32=True, seed=0, igemm2=False, monkeypatch=None):
    """fp32 side rows of a residual sum (sf_gemm_rows32 / sf_layernorm_fwd_rows32 / sf_row_scale_add_rows32): the side rows are
    the fp32 sum acc + bias + residual (fp32 residual rows when given) -- NOT rounded to the storage type anywhere -- the 16-bit
    rows are their rounding, every other row is the plain kernel's."""
    g = torch.Generator().manual_seed(seed)
    M = B * Ntok
    period = 1 if period_full else Ntok
    a = torch.randn((M, K), generator=g).to(ACT).float()
    w = (torch.randn((N, K), generator=g) / K ** 0.5).to(ACT).float()
    b = torch.randn(N, generator=g) if bias else None
    r16 = (torch.randn((M, N), generator=g) * 3).to(ACT).float()
    S = M // period
    r32 = (r16[::period] + torch.randn((S, N), generator=g) * 1e-3) if src32 else None      # NOT 16-bit representable
    lin = F.linear(a.double(), w.double(), b.double() if bias else None)
    ref16 = (lin + r16.double()).float()
    ref_side = (lin[::period] + (r32.double() if src32 else r16[::period].double())).float()
    dst = torch.full((S, N), float("nan"), device=device)
    if igemm2:
        monkeypatch.setenv("SF_IGEMM2_MINK", "32")
        monkeypatch.setenv("SF_IGEMM2_MINROWS", "1")
    out = tokens.gemm(_h(a, device), _h(w, device), bias=b.to(device) if bias else None, resid=_h(r16, device),
                      side=(period, r32.to(device) if src32 else None, dst))
    # side rows: fp32 accumulate of 16-bit products, no 16-bit rounding -> 1e-5 of the row scale
    assert_close("rows32 gemm side", dst.cpu(), ref_side, 2e-5)
    o = out.float().cpu()
    assert torch.equal(o[::period], dst.cpu().to(ACT).float()), "16-bit side rows must be the rounded fp32 rows"
    if not period_full:
        mask = torch.ones(M, dtype=torch.bool)
        mask[::period] = False
        assert_close("rows32 gemm other rows", o[mask], ref16[mask], 3 * F16_EPS)
    # LayerNorm reading the side rows: statistics and output of those rows come from the fp32 copy
    gamma, beta = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.2
    x16 = out                                                           # the stream tensor as the next kernel sees it
    yk, mean, rstd = tokens.layernorm_fwd(x16, gamma.to(device), beta.to(device), 1e-6, side=(period, dst))
    xmix = x16.float().cpu().clone()
    xmix[::period] = dst.cpu()
    yref = F.layer_norm(xmix, (N,), gamma, beta, 1e-6)
    assert_close("rows32 ln", yk.float().cpu(), yref, 2 * F16_EPS)
    assert_close("rows32 ln mean", mean.cpu(), xmix.mean(1), 1e-5)
    # stochastic-depth form of the same sum
    sc = torch.tensor([1.25, 0.0] * (B // 2) + [1.25] * (B % 2))
    y16 = (torch.randn((M, N), generator=g)).to(ACT).float()
    dst2 = torch.full((S, N), float("nan"), device=device)
    o2 = tokens.row_scale_add(_h(y16, device), sc.to(device), Ntok, resid=_h(r16, device),
                              side=(period, r32.to(device) if src32 else None, dst2))
    scale_rows = sc.repeat_interleave(Ntok).view(-1, 1)
    ref2 = r16 + scale_rows * y16
    ref2_side = (r32 if src32 else r16[::period]) + scale_rows[::period] * y16[::period]
    assert_close("rows32 row_scale_add side", dst2.cpu(), ref2_side, 1e-6)
    o2c = o2.float().cpu()
    assert torch.equal(o2c[::period], dst2.cpu().to(ACT).float())
    if not period_full:
        assert_close("rows32 row_scale_add rows", o2c[mask], ref2[mask], 2 * F16_EPS)
