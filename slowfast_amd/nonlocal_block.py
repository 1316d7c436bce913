"""Nonlocal block drop-in (slowfast/models/nonlocal_helper.py:10-144) on libsfamd kernels: theta / phi / g / out 1x1x1
convolutions (+bias) on the MFMA implicit GEMM, MaxPool3d of the phi/g input, the two affinity contractions as
per-sample batched GEMMs, softmax or 1/N normalisation, BatchNorm statistics in the out-conv epilogue, residual add.
Same constructor, children (conv_theta, conv_phi, conv_g, conv_out, bn, pool) and state_dict names as the reference.
"""
import torch
import torch.nn as nn

from . import lib as _sflib

from . import engine, ops, tokens
from .engine import ConvUnit, _grad_dest, _notify, as_cl, param_grads
from .lib import get_lib
from .x3d import cl5d, rows2d

_f16 = _sflib.act_dtype()        # fp16, or bf16 under SF_ACT_DTYPE=bf16 (lib.ACT_MODE)


def pool3d_fwd(x, k):
    N, C, T, H, W = x.shape
    To, Ho, Wo = T // k[0], H // k[1], W // k[2]
    out = ops.cl_empty((N, C, To, Ho, Wo), x.device)
    arg = torch.empty((N, To, Ho, Wo, C), dtype=torch.uint8, device=x.device)
    get_lib().call("sf_pool3d_fwd", N, T, H, W, C, k[0], k[1], k[2], x.data_ptr(), ops.cl_ld(x), out.data_ptr(),
                   ops.cl_ld(out), arg.data_ptr(), ops._stream(x), work=dict(bytes=2.0 * (x.numel() + out.numel())))
    return out, arg


def pool3d_bwd(dout, arg, in_shape, k):
    N, C, T, H, W = in_shape
    dx = ops.cl_empty(in_shape, dout.device)
    get_lib().call("sf_pool3d_bwd", N, T, H, W, C, k[0], k[1], k[2], arg.data_ptr(), dout.data_ptr(), ops.cl_ld(dout),
                   dx.data_ptr(), ops.cl_ld(dx), ops._stream(dout), work=dict(bytes=2.0 * (dx.numel() + 1.5 * dout.numel())))
    return dx


class BiasConvUnit(ConvUnit):
    """1x1x1 nn.Conv3d with bias (optionally followed by BatchNorm): ConvUnit + the bias gradient."""

    def backward(self, x, in_affine, dy, need_dx, resid=None):
        b = self.conv.bias
        if b is not None and b.requires_grad:
            db, zero_first = _grad_dest(b)
            tokens.bias_grad(rows2d(dy), db, accumulate=not zero_first)
        return super().backward(x, in_affine, dy, need_dx, resid=resid)


def _affinity(mod, theta, phi, g, N, S, T, H, W, device):
    """y = softmax(theta^T phi / sqrt(Ci)) g^T  or  (theta^T phi / P) g^T (nonlocal_helper.py:117-137) as per-sample batched
    GEMMs; returns y (N, Ci, T, H, W) and what the backward needs."""
    Ci = mod.dim_inner
    P = ops.rows(phi) // N
    ldp = (P + 7) // 8 * 8
    th2, ph2, g2 = rows2d(theta), rows2d(phi), rows2d(g)                # [N*S, Ci], [N*P, Ci]
    A = torch.empty((N, 1, S, ldp), dtype=_f16, device=device)
    softmax = mod.instantiation == "softmax"
    alpha = 0.0 if softmax else 1.0 / P
    tokens.bgemm_heads(th2, (S * Ci, 0), S, Ci, Ci, ph2, (P * Ci, 0), P, Ci, A, (S * ldp, 0), ldp, N, 1, alpha=alpha)
    desc = tokens.attn_desc(N, 1, Ci, False, (1, 1, S), (1, 1, P))
    if softmax:
        tokens.softmax_fwd(desc, A, Ci ** -0.5, None)
    elif ldp > P:
        A[..., P:].zero_()                                              # pad columns feed the next contraction
    gt = tokens.transpose_heads(g2, N, P, 1, Ci, ldp)                   # [N, 1, Ci, ldp]
    y2 = torch.empty((N * S, Ci), dtype=_f16, device=device)
    tokens.bgemm_heads(A, (S * ldp, 0), S, ldp, ldp, gt, (Ci * ldp, 0), Ci, ldp, y2, (S * Ci, 0), Ci, N, 1)
    y = cl5d(y2, N, Ci, (T, H, W))
    return y, A, desc, P, ldp


class NonlocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, *params):
        engine.record_params(ctx, params)
        x = as_cl(x)
        N, C, T, H, W = x.shape
        tr = mod.training
        S = T * H * W
        Ci = mod.dim_inner
        theta, _ = mod._theta.forward(x, None, tr)                         # (N, Ci, T, H, W)
        if mod.use_pool:
            xp, arg = pool3d_fwd(x, tuple(mod.pool_size))
            if engine.CAPTURE is not None:
                engine.CAPTURE.append({"kind": "nonlocal", "mod": mod, "argmax": arg, "kernel": tuple(mod.pool_size),
                                       "in_shape": tuple(x.shape)})
        else:
            xp, arg = x, None
        phi, _ = mod._phi.forward(xp, None, tr)
        g, _ = mod._g.forward(xp, None, tr)
        y, A, desc, P, ldp = _affinity(mod, theta, phi, g, N, S, T, H, W, x.device)
        praw, st = mod._out.forward(y, None, tr)
        out = ops.bn_act(praw, st.scale, st.shift, relu=False, resid=x)
        ctx.mod = mod
        ctx.sv = dict(theta=theta, phi=phi, g=g, A=A, y=y, praw=praw, st=st, xp=xp, arg=arg, desc=desc, P=P, ldp=ldp)
        ctx.save_for_backward(x)
        return out

    @staticmethod
    @engine.delivers_grads
    def backward(ctx, dout):
        mod, sv = ctx.mod, ctx.sv
        (x,) = ctx.saved_tensors
        dout = as_cl(dout)
        N, C, T, H, W = x.shape
        S, Ci, P, ldp = T * H * W, mod.dim_inner, sv["P"], sv["ldp"]
        softmax = mod.instantiation == "softmax"
        dpraw = mod._out.bn_backward(dout, sv["praw"], sv["st"])
        dy = mod._out.backward(sv["y"], None, dpraw, need_dx=True)
        dy2, th2, ph2, g2, A = rows2d(dy), rows2d(sv["theta"]), rows2d(sv["phi"]), rows2d(sv["g"]), sv["A"]
        # dA = dy g^T ; dg = A^T dy
        dA = torch.empty((N, 1, S, ldp), dtype=_f16, device=x.device)
        tokens.bgemm_heads(dy2, (S * Ci, 0), S, Ci, Ci, g2, (P * Ci, 0), P, Ci, dA, (S * ldp, 0), ldp, N, 1,
                           alpha=0.0 if softmax else 1.0 / P)
        dg2 = torch.empty((N * P, Ci), dtype=_f16, device=x.device)
        tokens.bgemm_tn_heads(A, (S * ldp, 0), ldp, dy2, (S * Ci, 0), Ci, S, P, Ci, dg2, (P * Ci, 0), Ci, N, 1)
        if softmax:
            dS, _ = tokens.softmax_bwd(sv["desc"], dA, A, Ci ** -0.5, want_drq=False)
        else:
            dS = dA                                                          # 1/P already applied through alpha
            if ldp > P:
                dS[..., P:].zero_()
        pt = tokens.transpose_heads(ph2, N, P, 1, Ci, ldp)
        dth2 = torch.empty((N * S, Ci), dtype=_f16, device=x.device)
        tokens.bgemm_heads(dS, (S * ldp, 0), S, ldp, ldp, pt, (Ci * ldp, 0), Ci, ldp, dth2, (S * Ci, 0), Ci, N, 1)
        dph2 = torch.empty((N * P, Ci), dtype=_f16, device=x.device)
        tokens.bgemm_tn_heads(dS, (S * ldp, 0), ldp, th2, (S * Ci, 0), Ci, S, P, Ci, dph2, (P * Ci, 0), Ci, N, 1)
        xp = sv["xp"]
        dphi, dg = cl5d(dph2, N, Ci, xp.shape[2:]), cl5d(dg2, N, Ci, xp.shape[2:])
        dxp = mod._phi.backward(xp, None, dphi, need_dx=True)
        dxp = mod._g.backward(xp, None, dg, need_dx=True, resid=dxp)
        if mod.use_pool:
            dxskip = pool3d_bwd(dxp, sv["arg"], tuple(x.shape), tuple(mod.pool_size))
            dxskip = ops.bn_act(dxskip, resid=dout)                           # + identity path
        else:
            dxskip = ops.bn_act(dxp, resid=dout)
        dx = mod._theta.backward(x, None, cl5d(dth2, N, Ci, (T, H, W)), need_dx=True, resid=dxskip)
        _notify(list(mod.parameters()))
        ctx.sv = None
        return (dx, None) + param_grads(ctx, 2)


class Nonlocal(nn.Module):
    def __init__(self, dim, dim_inner, pool_size=None, instantiation="softmax", zero_init_final_conv=False,
                 zero_init_final_norm=True, norm_eps=1e-5, norm_momentum=0.1, norm_module=nn.BatchNorm3d):
        super().__init__()
        if instantiation not in ("softmax", "dot_product"):
            raise NotImplementedError(f"Unknown norm type {instantiation}")
        self.dim, self.dim_inner, self.pool_size, self.instantiation = dim, dim_inner, pool_size, instantiation
        self.use_pool = False if pool_size is None else any(size > 1 for size in pool_size)
        self.norm_eps, self.norm_momentum = norm_eps, norm_momentum
        self.conv_theta = nn.Conv3d(dim, dim_inner, kernel_size=1, stride=1, padding=0)
        self.conv_phi = nn.Conv3d(dim, dim_inner, kernel_size=1, stride=1, padding=0)
        self.conv_g = nn.Conv3d(dim, dim_inner, kernel_size=1, stride=1, padding=0)
        self.conv_out = nn.Conv3d(dim_inner, dim, kernel_size=1, stride=1, padding=0)
        self.conv_out.zero_init = zero_init_final_conv
        self.bn = norm_module(num_features=dim, eps=norm_eps, momentum=norm_momentum)
        self.bn.transform_final_bn = zero_init_final_norm
        if self.use_pool:
            self.pool = nn.MaxPool3d(kernel_size=pool_size, stride=pool_size, padding=[0, 0, 0])
        self._theta, self._phi, self._g = BiasConvUnit(self.conv_theta), BiasConvUnit(self.conv_phi), BiasConvUnit(self.conv_g)
        self._out = BiasConvUnit(self.conv_out, self.bn)

    def forward(self, x):
        if not self.training and self.__dict__.get("_sf_infer"):
            return self._infer(x)
        return NonlocalFn.apply(x, self, *self.parameters())

    # inference fusion (slowfast_amd.inference): x + bn(conv_out(y)) is one launch (BatchNorm folded, residual epilogue)
    def _sf_fold(self):
        for u in (self._theta, self._phi, self._g, self._out):
            u.fold()

    def _infer(self, x):
        x = as_cl(x)
        N, C, T, H, W = x.shape
        theta = self._theta.infer(x)
        xp = pool3d_fwd(x, tuple(self.pool_size))[0] if self.use_pool else x
        phi, g = self._phi.infer(xp), self._g.infer(xp)
        y = _affinity(self, theta, phi, g, N, T * H * W, T, H, W, x.device)[0]
        return self._out.infer(y, relu=False, resid=x)
