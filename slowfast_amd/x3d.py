"""X3D drop-ins: ``X3DStem``, ``SE``, ``X3DTransform``, ``X3DHead`` and the ``X3D`` model builder with the reference's
constructor signatures, cfg keys and state_dict names (slowfast/models/stem_helper.py:204-285, operators.py:15-59,
resnet_helper.py:118-256, head_helper.py:353-488, video_model_builder.py:663-802), executed by hand-written
schedules of libsfamd kernels: 1x1x1 convs on the MFMA implicit GEMM, the depthwise 3x3x3 / (5,1,1) stencils,
BatchNorm statistics in the producers' epilogues, the SE squeeze/gate and the gate*BN->Swish pass.

X3D-M widths 54 and 108 are not multiples of 8: activation buffers are padded to 56 / 112 channels whose pad
lanes are exact zeros end to end (zero weight rows, zero BN scale/shift); parameters keep the reference shapes.
"""
import math
import os
from ctypes import byref

import torch
import torch.nn as nn

from . import lib as _sflib

from . import engine, ops, tokens
from .engine import BNState, ConvUnit, StemConvUnit, _grad_dest, _notify, _sync_of, as_cl, bn_statistics, param_grads
from .lib import get_lib
from .registry import MODEL_REGISTRY
from .resblocks import ResStage, _TRANS
from .video_models import get_norm, init_weights

_f16 = _sflib.act_dtype()        # fp16, or bf16 under SF_ACT_DTYPE=bf16 (lib.ACT_MODE)

# SF_GATE_BN_FUSE=0: the separate sf_bn_bwd_reduce pass behind sf_gate_act_bwd (A/B runs; profiles/r6_v22_gate_bn_fuse_ab.txt)
GATE_BN_FUSE = os.environ.get("SF_GATE_BN_FUSE", "1") != "0"
# SF_SE_FROM_STATS=0: the SE squeeze as a pass of its own over the activation (sf_sample_mean) instead of from the depthwise
# convolution's statistics table (A/B runs; profiles/r6_v24_se_from_stats_ab.txt)
SE_FROM_STATS = os.environ.get("SF_SE_FROM_STATS", "1") != "0"
# SF_GATE_ONE_PASS=0: SE blocks run sf_gate_grad + sf_gate_act_bwd (two passes over y and dz) instead of sf_gate_bwd_sums (A/B runs)
GATE_ONE_PASS = os.environ.get("SF_GATE_ONE_PASS", "1") != "0"


def _pad8(c):
    return (c + 7) // 8 * 8


def rows2d(x):
    """Channels-last (N,C,T,H,W) activation -> its [N*T*H*W, C] row view (pitch preserved)."""
    ld = ops.cl_ld(x)
    return torch.as_strided(x, (ops.rows(x), x.shape[1]), (ld, 1), x.storage_offset())


def cl5d(y2d, N, C, thw):
    """[N*T*H*W, C] contiguous rows -> channels-last (N,C,T,H,W) view."""
    T, H, W = thw
    return y2d.view(N, T, H, W, C).permute(0, 4, 1, 2, 3)


# ------------------------------------------------------------------------------------------------
# wrappers over the X3D entry points of the C ABI
def sample_mean(y, scale, shift, relu):
    """mean over (T,H,W) of relu?(y*scale+shift) per (n, c) -> fp32 [N, C]."""
    N, C = y.shape[:2]
    S = ops.rows(y) // N
    lib = get_lib()
    chunks = lib.call("sf_sample_chunks", S, C)
    part = torch.empty((N * chunks, 2, C), dtype=torch.float32, device=y.device)
    out = torch.empty((N, C), dtype=torch.float32, device=y.device)
    lib.call("sf_sample_mean", N, S, C, y.data_ptr(), ops.cl_ld(y), ops._ptr(scale), ops._ptr(shift), int(bool(relu)),
             part.data_ptr(), out.data_ptr(), ops._stream(y), work=dict(bytes=2.0 * y.numel()))
    return out


def gate_grad(y, scale, shift, dz, gate, swish):
    N, C = y.shape[:2]
    S = ops.rows(y) // N
    lib = get_lib()
    chunks = lib.call("sf_sample_chunks", S, C)
    part = torch.empty((N * chunks, 2, C), dtype=torch.float32, device=y.device)
    out = torch.empty((N, C), dtype=torch.float32, device=y.device)
    lib.call("sf_gate_grad", N, S, C, y.data_ptr(), ops.cl_ld(y), scale.data_ptr(), shift.data_ptr(), dz.data_ptr(),
             ops.cl_ld(dz), ops._ptr(gate), int(bool(swish)), part.data_ptr(), out.data_ptr(), ops._stream(y),
             work=dict(bytes=4.0 * y.numel()))
    return out


def gate_act_fwd(y, scale, shift, gate, swish):
    N, C = y.shape[:2]
    S = ops.rows(y) // N
    z = ops.cl_empty(y.shape, y.device)
    get_lib().call("sf_gate_act_fwd", N, S, C, y.data_ptr(), ops.cl_ld(y), scale.data_ptr(), shift.data_ptr(),
                   ops._ptr(gate), int(bool(swish)), z.data_ptr(), ops.cl_ld(z), ops._stream(y),
                   work=dict(bytes=4.0 * y.numel()))
    return z


def gate_act_bwd(y, scale, shift, gate, swish, dz, dmean, bn_part=False):
    """``bn_part``: also return the [rows, 2, C] column sums of du and du * y -- the reduction of the BatchNorm backward that
    follows, taken in this pass instead of one of its own (ops.bn_bwd(..., part=...))."""
    N, C = y.shape[:2]
    S = ops.rows(y) // N
    du = ops.cl_empty(y.shape, y.device)
    lib = get_lib()
    if bn_part:
        part = torch.empty((lib.call("sf_gate_act_bwd_bn_rows", N, S, C), 2, C), dtype=torch.float32, device=y.device)
        lib.call("sf_gate_act_bwd_bn", N, S, C, y.data_ptr(), ops.cl_ld(y), scale.data_ptr(), shift.data_ptr(),
                 ops._ptr(gate), int(bool(swish)), dz.data_ptr(), ops.cl_ld(dz), ops._ptr(dmean), du.data_ptr(),
                 ops.cl_ld(du), part.data_ptr(), ops._stream(y), work=dict(bytes=6.0 * y.numel()))
        return du, part
    lib.call("sf_gate_act_bwd", N, S, C, y.data_ptr(), ops.cl_ld(y), scale.data_ptr(), shift.data_ptr(),
             ops._ptr(gate), int(bool(swish)), dz.data_ptr(), ops.cl_ld(dz), ops._ptr(dmean), du.data_ptr(),
             ops.cl_ld(du), ops._stream(y), work=dict(bytes=6.0 * y.numel()))
    return du


def gate_bwd_sums(y, scale, shift, gate, swish, dz):
    """ONE pass over y and dz (round 6): returns du0 = dz * act'(gate * u) * gate (the gate / Swish backward without the SE squeeze
    term) and sums [N, 3, C] = per-sample sums of dz * act'(gate * u) * u (the gate's gradient), du0, du0 * y."""
    N, C = y.shape[:2]
    S = ops.rows(y) // N
    lib = get_lib()
    chunks = lib.call("sf_sample_chunks", S, C)
    part = torch.empty((N * chunks, 4, C), dtype=torch.float32, device=y.device)
    sums = torch.empty((N, 3, C), dtype=torch.float32, device=y.device)
    du0 = ops.cl_empty(y.shape, y.device)
    lib.call("sf_gate_bwd_sums", N, S, C, y.data_ptr(), ops.cl_ld(y), scale.data_ptr(), shift.data_ptr(), dz.data_ptr(),
             ops.cl_ld(dz), ops._ptr(gate), int(bool(swish)), du0.data_ptr(), ops.cl_ld(du0), part.data_ptr(), sums.data_ptr(),
             ops._stream(y), work=dict(bytes=6.0 * y.numel()))
    return du0, sums


class BNUnit:
    """Stand-alone nn.BatchNorm3d container fed by per-block partial sums (dwconv epilogue)."""

    def __init__(self, bn):
        assert bn.momentum is not None
        self.bn = bn

    def finalize(self, part, count, C, training):
        return bn_statistics(self.bn, part, count, C, training)

    def backward(self, dz, y, st, relu_self=False, part=None, sample_add=None):
        bn = self.bn
        dgamma, zg = _grad_dest(bn.weight)
        dbeta, zb = _grad_dest(bn.bias)
        assert zg == zb
        return ops.bn_bwd(dz, y, bn.weight, st.mean, st.rstd, dgamma, dbeta,
                          relu_affine=(st.scale, st.shift) if relu_self else None, accumulate=not zg, sync=_sync_of(bn),
                          part=part, sample_add=sample_add)


class DwUnit:
    """Depthwise nn.Conv3d(C, C, k, groups=C) container bound to the stencil kernels (channels-last 5-D tensors)."""

    def __init__(self, conv):
        assert conv.groups == conv.in_channels == conv.out_channels and conv.bias is None
        assert conv.dilation == (1, 1, 1)
        self.conv = conv
        self._geoms = {}

    def geom(self, shape):
        key = tuple(shape)
        g = self._geoms.get(key)
        if g is None:
            N, Cp, T, H, W = key
            c = self.conv
            g = self._geoms[key] = tokens.DwGeom(N, Cp, Cp, (T, H, W), c.kernel_size, c.stride, c.padding, cls=0,
                                                 Cw_real=c.out_channels)
        return g

    def forward(self, x, stats=True):
        g = self.geom(x.shape)
        res = tokens.dwconv_fwd(rows2d(x), self.conv.weight, g, stats=stats)
        y2d, part = res if stats else (res, None)
        return cl5d(y2d, x.shape[0], x.shape[1], g.out_thw), part, g

    def backward(self, x, dy, need_dx=True):
        g = self.geom(x.shape)
        w = self.conv.weight
        if w.requires_grad:
            dw, zero_first = _grad_dest(w)
            tokens.dwconv_wgrad(rows2d(x), rows2d(dy), g, dw, zero_first=zero_first)
        if not need_dx:
            return None
        dx2d = tokens.dwconv_dgrad(rows2d(dy), w, g)
        return cl5d(dx2d, x.shape[0], x.shape[1], g.thw)


# ------------------------------------------------------------------------------------------------
class X3DStemFn(torch.autograd.Function):
    """conv_xy (1,3,3)/(1,2,2) -> depthwise (5,1,1) -> BN -> ReLU (stem_helper.py:279-285)."""

    @staticmethod
    def forward(ctx, x, mod, *params):
        engine.record_params(ctx, params)
        xy, dw, bn = mod._xy, mod._dw, mod._bn
        xcl = xy.prepare_input(x)
        y1, _ = xy.forward(xcl, None, mod.training)
        y2, part, g = dw.forward(y1, stats=True)
        st = bn.finalize(part, g.rows_out, y2.shape[1], mod.training)
        out = ops.bn_act(y2, st.scale, st.shift, relu=True)
        if engine.CAPTURE is not None:
            engine.CAPTURE.append({"kind": "x3d_stem", "mod": mod, "raw": [y2], "bn": [(st.scale, st.shift)]})
        ctx.mod, ctx.sv = mod, (xcl, y1, y2, st)
        return out

    @staticmethod
    @engine.delivers_grads
    def backward(ctx, dout):
        mod = ctx.mod
        xcl, y1, y2, st = ctx.sv
        dy2 = mod._bn.backward(as_cl(dout), y2, st, relu_self=True)
        dy1 = mod._dw.backward(y1, dy2, need_dx=True)
        mod._xy.backward(xcl, None, dy1, need_dx=False)
        _notify(list(mod.parameters()))
        ctx.sv = None
        return (None, None) + param_grads(ctx, 2)


class X3DBlockFn(torch.autograd.Function):
    """relu(shortcut(x) + X3DTransform(x)) (resnet_helper.py:253-256, 512-521)."""

    @staticmethod
    def forward(ctx, x, mod, *params):
        engine.record_params(ctx, params)
        ctx.prev_bn = getattr(x, "_sf_block_bn", None) if engine.BN_FUSE_REDUCE else None      # see engine.ResBlockFn
        x = as_cl(x)
        t = mod.branch2
        tr = mod.training
        A, C, P = t._a, t._c, mod._proj
        ya, sa = A.forward(x, None, tr)
        za = ops.bn_act(ya, sa.scale, sa.shift, relu=True)
        yb, part, g = t._b.forward(za, stats=True)
        # SE squeeze (operators.py:38-45) = per-sample mean of bn(y) = an affine map of the per-sample channel sums, which the
        # plane sweep's statistics table already holds (one row per (sample, tile)): no pass over yb.  Taken BEFORE the
        # finalize (which may fold the table in place).
        nrow = getattr(g, "stat_sample_rows", 0) if (t._se is not None and SE_FROM_STATS) else 0
        ysum = part.view(yb.shape[0], nrow, 2, -1)[:, :, 0].sum(1) if nrow > 0 else None
        sb = t._b_bn.finalize(part, g.rows_out, yb.shape[1], tr)
        gate = se = None
        if t._se is not None:
            if ysum is not None:
                m = torch.addcmul(sb.shift, ysum, sb.scale, value=float(yb.shape[0]) / float(g.rows_out))
            else:
                m = sample_mean(yb, sb.scale, sb.shift, relu=False)
            h, gate = t._se.gate_fwd(m)
            se = (m, h)
        zb = gate_act_fwd(yb, sb.scale, sb.shift, gate, t._swish_inner)
        yc, sc = C.forward(zb, None, tr)
        # the backward pass needs only the sign of the block output: a 1-bit mask stands in for it (engine.ResBlockFn)
        if P is not None:
            y1, s1 = P.forward(x, None, tr)
            out, bits = ops.bn_act(yc, sc.scale, sc.shift, relu=True, resid=y1, rscale=s1.scale, rshift=s1.shift,
                                   want_mask=True)
        else:
            y1, s1 = None, None
            out, bits = ops.bn_act(yc, sc.scale, sc.shift, relu=True, resid=x, want_mask=True)
        if engine.CAPTURE is not None:
            engine.CAPTURE.append({"kind": "x3d_block", "mod": mod, "raw": [ya], "bn": [(sa.scale, sa.shift)], "out": out,
                                   "se_h": None if se is None else se[1]})
        ctx.mod = mod
        ctx.sv = dict(ya=ya, sa=sa, za=za, yb=yb, sb=sb, gate=gate, se=se, zb=zb, yc=yc, sc=sc, y1=y1, s1=s1, bits=bits,
                      ysum=ysum)
        ctx.save_for_backward(x)
        if engine.BN_FUSE_REDUCE and tr:
            out._sf_block_bn = {"bits": bits, "y0": yc, "sync": _sync_of(C.bn) is not None}
        return out

    @staticmethod
    @engine.delivers_grads
    def backward(ctx, dout):
        mod, sv = ctx.mod, ctx.sv
        t = mod.branch2
        A, C, P = t._a, t._c, mod._proj
        (x,) = ctx.saved_tensors
        part_c = engine.tagged_bn_part(dout, sv["yc"])       # still describing dout? (engine.tag_bn_part)
        dout = as_cl(dout)
        need_dx = ctx.needs_input_grad[0]
        bits = sv["bits"]
        dyc = C.bn_backward(dout, sv["yc"], sv["sc"], zmask=bits, part=part_c)
        if P is not None:
            dy1 = P.bn_backward(dout, sv["y1"], sv["s1"], zmask=bits)
        dzb = C.backward(sv["zb"], None, dyc, need_dx=True)
        yb, sb, gate = sv["yb"], sv["sb"], sv["gate"]
        dmean = None
        if t._se is not None and sv["ysum"] is not None and engine.BN_FUSE_REDUCE and GATE_ONE_PASS:
            # SE block, ONE pass over (yb, dzb) instead of three (sf_gate_grad, sf_gate_act_bwd, sf_bn_bwd_reduce): the squeeze
            # term dmean[n][c] / S of du is a per-sample constant, so it is never stored -- the pass leaves du0 and the
            # per-sample sums; b_bn's reduction follows from them and from the per-sample sums of yb the forward kept
            # (sum du = sum_n (sum du0_n + dmean_n), sum du * y = sum_n (sum (du0 y)_n + dmean_n / S * sum y_n)), and the apply
            # pass adds the constant back: dyb = k1 * (du0 + dmean_n / S) + k2 + k3 * yb.
            du, sums = gate_bwd_sums(yb, sb.scale, sb.shift, gate, t._swish_inner, dzb)
            dmean = t._se.gate_bwd(sv["se"][0], sv["se"][1], gate, sums[:, 0].contiguous())
            add = dmean * (float(yb.shape[0]) / float(ops.rows(yb)))                    # dmean / S, [N, C]
            part_b = torch.stack(((sums[:, 1] + dmean).sum(0), (sums[:, 2] + add * sv["ysum"]).sum(0))).unsqueeze(0).contiguous()
            dyb = t._b_bn.backward(du, yb, sb, part=part_b, sample_add=add.contiguous())
        else:
            if t._se is not None:
                dgate = gate_grad(yb, sb.scale, sb.shift, dzb, gate, t._swish_inner)
                dmean = t._se.gate_bwd(sv["se"][0], sv["se"][1], gate, dgate)
            if engine.BN_FUSE_REDUCE and GATE_BN_FUSE:  # the reduction of b_bn's backward rides on the gate / Swish backward pass
                du, part_b = gate_act_bwd(yb, sb.scale, sb.shift, gate, t._swish_inner, dzb, dmean, bn_part=True)
            else:
                du, part_b = gate_act_bwd(yb, sb.scale, sb.shift, gate, t._swish_inner, dzb, dmean), None
            dyb = t._b_bn.backward(du, yb, sb, part=part_b)
        dza = t._b.backward(sv["za"], dyb, need_dx=True)
        dya = A.bn_backward(dza, sv["ya"], sv["sa"], relu_self=True)
        prev = ctx.prev_bn if need_dx else None
        if prev is not None and prev["sync"]:      # the PRODUCER's BatchNorm reduces its sums across ranks: not fused
            prev = None
        if P is not None:
            dx1 = P.backward(x, None, dy1, need_dx=need_dx)
            dx = A.backward(x, None, dya, need_dx=need_dx, resid=dx1, bn_fuse=prev)
        else:       # identity shortcut: the masked block-output gradient is added in the dgrad epilogue
            dx = A.backward(x, None, dya, need_dx=need_dx, resid=dout, resid_bits=bits, bn_fuse=prev)
        if prev is not None:
            dx, pc = dx
            if pc is not None:
                engine.tag_bn_part(dx, prev["y0"], pc)
        _notify(mod._param_list)
        ctx.sv = ctx.prev_bn = None
        return (dx, None) + param_grads(ctx, 2)


class X3DHeadPoolFn(torch.autograd.Function):
    """conv_5 -> BN -> ReLU -> average pool over the whole (T,H,W) extent -> fp32 (N, dim_inner)
    (head_helper.py:461-465); the tiny lin_5 / projection layers stay torch fp32 ops."""

    @staticmethod
    def forward(ctx, x, mod, *params):
        engine.record_params(ctx, params)
        x = as_cl(x)
        unit = mod._conv5
        y, st = unit.forward(x, None, mod.training)
        m = sample_mean(y, st.scale, st.shift, relu=True)
        if engine.CAPTURE is not None:
            engine.CAPTURE.append({"kind": "x3d_head", "mod": mod, "raw": [y], "bn": [(st.scale, st.shift)]})
        ctx.mod, ctx.y, ctx.st = mod, y, st
        ctx.save_for_backward(x)
        return m[:, :unit.conv.out_channels].contiguous()

    @staticmethod
    @engine.delivers_grads
    def backward(ctx, dm):
        mod, y, st = ctx.mod, ctx.y, ctx.st
        unit = mod._conv5
        (x,) = ctx.saved_tensors
        N, Cp = y.shape[:2]
        S = ops.rows(y) // N
        dmp = torch.zeros((N, Cp), dtype=torch.float32, device=dm.device)
        dmp[:, :dm.shape[1]] = dm / S
        dz2d = dmp.to(_f16)[:, None, :].expand(N, S, Cp).contiguous().view(N * S, Cp)
        dz = cl5d(dz2d, N, Cp, y.shape[2:])
        dy = unit.bn_backward(dz, y, st, relu_self=True)
        dx = unit.backward(x, None, dy, need_dx=ctx.needs_input_grad[0])
        _notify(unit.params())
        ctx.y = None
        return (dx, None) + param_grads(ctx, 2)


# ------------------------------------------------------------------------------------------------
class X3DStem(nn.Module):
    def __init__(self, dim_in, dim_out, kernel, stride, padding, inplace_relu=True, eps=1e-5, bn_mmt=0.1,
                 norm_module=nn.BatchNorm3d):
        super().__init__()
        self.kernel, self.stride, self.padding = kernel, stride, padding
        self.inplace_relu, self.eps, self.bn_mmt = inplace_relu, eps, bn_mmt
        self.conv_xy = nn.Conv3d(dim_in, dim_out, kernel_size=(1, kernel[1], kernel[2]), stride=(1, stride[1], stride[2]),
                                 padding=(0, padding[1], padding[2]), bias=False)
        self.conv = nn.Conv3d(dim_out, dim_out, kernel_size=(kernel[0], 1, 1), stride=(stride[0], 1, 1),
                              padding=(padding[0], 0, 0), bias=False, groups=dim_out)
        self.bn = norm_module(num_features=dim_out, eps=eps, momentum=bn_mmt)
        self.relu = nn.ReLU(inplace_relu)
        self._xy, self._dw, self._bn = StemConvUnit(self.conv_xy, None), DwUnit(self.conv), BNUnit(self.bn)

    def forward(self, x):
        return X3DStemFn.apply(x, self, *self.parameters())


class SE(nn.Module):
    """Squeeze-and-Excitation: AvgPool, FC, ReLU (or Swish), FC, Sigmoid (operators.py:15-59)."""

    @staticmethod
    def _round_width(width, multiplier, min_width=8, divisor=8):
        if not multiplier:
            return width
        width *= multiplier
        min_width = min_width or divisor
        width_out = max(min_width, int(width + divisor / 2) // divisor * divisor)
        if width_out < 0.9 * width:
            width_out += divisor
        return int(width_out)

    def __init__(self, dim_in, ratio, relu_act=True):
        super().__init__()
        if not relu_act:
            raise NotImplementedError("SE with a Swish squeeze activation is not used by any X3D config")
        self.avg_pool = nn.AdaptiveAvgPool3d((1, 1, 1))
        dim_fc = self._round_width(dim_in, ratio)
        self.fc1 = nn.Conv3d(dim_in, dim_fc, 1, bias=True)
        self.fc1_act = nn.ReLU()
        self.fc2 = nn.Conv3d(dim_fc, dim_in, 1, bias=True)
        self.fc2_sig = nn.Sigmoid()
        self.dim_in, self.dim_fc = dim_in, dim_fc

    def gate_fwd(self, m):
        N, Cp = m.shape
        h = torch.empty((N, self.dim_fc), dtype=torch.float32, device=m.device)
        gate = torch.empty((N, Cp), dtype=torch.float32, device=m.device)
        get_lib().call("sf_se_gate_fwd", N, self.dim_in, Cp, self.dim_fc, m.data_ptr(), self.fc1.weight.data_ptr(),
                       self.fc1.bias.data_ptr(), self.fc2.weight.data_ptr(), self.fc2.bias.data_ptr(), h.data_ptr(),
                       gate.data_ptr(), ops._stream(m))
        return h, gate

    def gate_bwd(self, m, h, gate, dgate):
        """Writes the fc1/fc2 weight and bias gradients; returns d(loss)/d(squeezed means) [N, Cp]."""
        N, Cp = m.shape
        C, F = self.dim_in, self.dim_fc
        dev = m.device
        dpre2 = torch.empty((N, Cp), dtype=torch.float32, device=dev)
        dpre1 = torch.empty((N, F), dtype=torch.float32, device=dev)
        dm = torch.empty((N, Cp), dtype=torch.float32, device=dev)
        lib, s = get_lib(), ops._stream(m)
        lib.call("sf_se_gate_bwd", N, C, Cp, F, gate.data_ptr(), h.data_ptr(), self.fc1.weight.data_ptr(),
                 self.fc2.weight.data_ptr(), dgate.data_ptr(), dpre2.data_ptr(), dpre1.data_ptr(), dm.data_ptr(), s)
        for a, lda, b, ldb, I, J, prm in ((dpre2, Cp, h, F, C, F, self.fc2.weight), (dpre2, Cp, None, 0, C, 1, self.fc2.bias),
                                          (dpre1, F, m, Cp, F, C, self.fc1.weight), (dpre1, F, None, 0, F, 1, self.fc1.bias)):
            dst, zero_first = _grad_dest(prm)
            lib.call("sf_outer_sum", a.data_ptr(), lda, ops._ptr(b), ldb, N, I, J, dst.data_ptr(), 1.0, int(not zero_first), s)
        return dm


class X3DTransform(nn.Module):
    """1x1x1 -> BN -> ReLU -> depthwise Tx3x3 -> BN -> [SE] -> Swish -> 1x1x1 -> BN; children a, a_bn, a_relu, b, b_bn,
    [se], b_relu, c, c_bn (resnet_helper.py:118-256)."""

    def __init__(self, dim_in, dim_out, temp_kernel_size, stride, dim_inner, num_groups, stride_1x1=False,
                 inplace_relu=True, eps=1e-5, bn_mmt=0.1, dilation=1, norm_module=nn.BatchNorm3d, se_ratio=0.0625,
                 swish_inner=True, block_idx=0):
        super().__init__()
        if num_groups != dim_inner or stride_1x1 or dilation != 1:
            raise NotImplementedError("X3DTransform: channelwise 3x3x3 (X3D.CHANNELWISE_3x3x3), stride on the 3x3x3")
        self.temp_kernel_size = temp_kernel_size
        self._inplace_relu, self._eps, self._bn_mmt = inplace_relu, eps, bn_mmt
        self._se_ratio, self._swish_inner, self._stride_1x1, self._block_idx = se_ratio, swish_inner, stride_1x1, block_idx
        bn = dict(eps=eps, momentum=bn_mmt)
        self.a = nn.Conv3d(dim_in, dim_inner, kernel_size=[1, 1, 1], stride=[1, 1, 1], padding=[0, 0, 0], bias=False)
        self.a_bn = norm_module(num_features=dim_inner, **bn)
        self.a_relu = nn.ReLU(inplace=inplace_relu)
        self.b = nn.Conv3d(dim_inner, dim_inner, [temp_kernel_size, 3, 3], stride=[1, stride, stride],
                           padding=[int(temp_kernel_size // 2), 1, 1], groups=num_groups, bias=False, dilation=[1, 1, 1])
        self.b_bn = norm_module(num_features=dim_inner, **bn)
        self.__dict__["_se"] = None                     # plain attribute: the SE module is registered once, as `se`
        if se_ratio > 0.0 and (block_idx + 1) % 2:
            self.se = SE(dim_inner, se_ratio)
            self.__dict__["_se"] = self.se
        if swish_inner:
            self.b_relu = nn.SiLU()             # parameter-free stand-in for pytorchvideo's Swish (x * sigmoid(x))
        else:
            self.b_relu = nn.ReLU(inplace=inplace_relu)
        self.c = nn.Conv3d(dim_inner, dim_out, kernel_size=[1, 1, 1], stride=[1, 1, 1], padding=[0, 0, 0], bias=False)
        self.c_bn = norm_module(num_features=dim_out, **bn)
        self.c_bn.transform_final_bn = True
        self._a, self._c = ConvUnit(self.a, self.a_bn), ConvUnit(self.c, self.c_bn)
        self._b, self._b_bn = DwUnit(self.b), BNUnit(self.b_bn)

    def forward(self, x):
        raise NotImplementedError("X3DTransform runs fused inside ResBlock (engine X3DBlockFn)")

    # inference fusion (slowfast_amd.inference): the two 1x1x1 convolutions and the projection shortcut take the folded
    # single-launch form (BatchNorm in the weights, ReLU / residual in the epilogue) -- two full elementwise passes
    # fewer per block on a bandwidth-bound network; depthwise 3x3x3 -> BN -> SE -> Swish keeps running statistics
    def _sf_fold_in_block(self, block):
        for u in (self._a, self._c, block._proj):
            if u is not None:
                u.fold()
        return True

    def _infer_in_block(self, block, x):
        za = self._a.infer(x, relu=True)
        yb, _, g = self._b.forward(za, stats=False)
        sb = self._b_bn.finalize(None, g.rows_out, yb.shape[1], False)
        gate = None
        if self._se is not None:
            gate = self._se.gate_fwd(sample_mean(yb, sb.scale, sb.shift, relu=False))[1]
        zb = gate_act_fwd(yb, sb.scale, sb.shift, gate, self._swish_inner)
        sc = x if block._proj is None else block._proj.infer(x)
        return self._c.infer(zb, relu=True, resid=sc)


_TRANS["x3d_transform"] = X3DTransform


class X3DHead(nn.Module):
    def __init__(self, dim_in, dim_inner, dim_out, num_classes, pool_size, dropout_rate=0.0, act_func="softmax",
                 inplace_relu=True, eps=1e-5, bn_mmt=0.1, norm_module=nn.BatchNorm3d, bn_lin5_on=False):
        super().__init__()
        self.pool_size, self.dropout_rate, self.num_classes, self.act_func = pool_size, dropout_rate, num_classes, act_func
        self.eps, self.bn_mmt, self.inplace_relu, self.bn_lin5_on = eps, bn_mmt, inplace_relu, bn_lin5_on
        self.conv_5 = nn.Conv3d(dim_in, dim_inner, kernel_size=(1, 1, 1), stride=(1, 1, 1), padding=(0, 0, 0), bias=False)
        self.conv_5_bn = norm_module(num_features=dim_inner, eps=eps, momentum=bn_mmt)
        self.conv_5_relu = nn.ReLU(inplace_relu)
        self.avg_pool = nn.AdaptiveAvgPool3d((1, 1, 1)) if pool_size is None else nn.AvgPool3d(pool_size, stride=1)
        self.lin_5 = nn.Conv3d(dim_inner, dim_out, kernel_size=(1, 1, 1), stride=(1, 1, 1), padding=(0, 0, 0), bias=False)
        if bn_lin5_on:                      # X3D.BN_LIN5 (head_helper.py:440-443): on the pooled (B, dim_out, 1, 1, 1) features,
            self.lin_5_bn = norm_module(num_features=dim_out, eps=eps, momentum=bn_mmt)   # fp32 torch like the FC layers
        self.lin_5_relu = nn.ReLU(inplace_relu)
        if dropout_rate > 0.0:
            self.dropout = nn.Dropout(dropout_rate)
        self.projection = nn.Linear(dim_out, num_classes, bias=True)
        if act_func == "softmax":
            self.act = nn.Softmax(dim=4)
        elif act_func == "sigmoid":
            self.act = nn.Sigmoid()
        else:
            raise NotImplementedError(f"{act_func} is not supported as an activationfunction.")
        self._conv5 = ConvUnit(self.conv_5, self.conv_5_bn)

    def forward(self, inputs):
        assert len(inputs) == 1, "Input tensor does not contain 1 pathway"
        x = inputs[0]
        if self.pool_size is not None and tuple(self.pool_size) != tuple(x.shape[2:]):
            return self._forward_sliding(x)
        m = X3DHeadPoolFn.apply(x, self, self.conv_5.weight, self.conv_5_bn.weight, self.conv_5_bn.bias)
        z = torch.nn.functional.linear(m, self.lin_5.weight.view(self.lin_5.out_channels, -1))
        if self.bn_lin5_on:
            z = self.lin_5_bn(z.view(z.shape[0], -1, 1, 1, 1)).view(z.shape[0], -1)
        if engine.CAPTURE is not None:
            engine.CAPTURE.append({"kind": "x3d_lin5", "mod": self, "pre": z.detach()})
        z = torch.relu(z)
        if hasattr(self, "dropout"):
            z = self.dropout(z)
        z = self.projection(z)
        if not self.training:
            z = torch.softmax(z, 1) if self.act_func == "softmax" else torch.sigmoid(z)
        return z.view(z.shape[0], -1)

    def _forward_sliding(self, x):
        """Fully-convolutional inference (head_helper.py:461-488 on DATA.TEST_CROP_SIZE > TRAIN_CROP_SIZE clips, e.g.
        X3D-M 224 -> 256): conv_5 + BN + ReLU on the kernels, then the AvgPool3d window slides over the (tiny)
        feature map and lin_5 / projection / activation / spatial mean run per window position in fp32."""
        from .engine import ConvBNActFn
        unit = self._conv5
        y = ConvBNActFn.apply(x, unit, True, self.training, *unit.params())
        y = y[:, :self.conv_5.out_channels].float()
        z = self.avg_pool(y.contiguous())
        z = torch.nn.functional.conv3d(z, self.lin_5.weight)
        if self.bn_lin5_on:
            z = self.lin_5_bn(z)
        z = torch.relu(z).permute(0, 2, 3, 4, 1)
        if hasattr(self, "dropout"):
            z = self.dropout(z)
        z = self.projection(z)
        if not self.training:
            z = self.act(z).mean([1, 2, 3])
        return z.reshape(z.shape[0], -1)


def round_width(width, multiplier, min_width=8, divisor=8):
    """slowfast/models/utils.py:10-23 with the X3D defaults used at video_model_builder.py:690-705."""
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    width_out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if width_out < 0.9 * width:
        width_out += divisor
    return int(width_out)


@MODEL_REGISTRY.register()
class X3D(nn.Module):
    """X3D backbone; forward(x=[clip NCTHW]) -> logits (B, num_classes)."""

    def __init__(self, cfg):
        super().__init__()
        self.norm_module = get_norm(cfg)
        assert not cfg.DETECTION.ENABLE
        self.enable_detection, self.num_pathways, self.cfg = False, 1, cfg
        x = cfg.X3D
        self.dim_c1 = x.DIM_C1
        self.dim_res2 = round_width(self.dim_c1, 2.0, divisor=8) if x.SCALE_RES2 else self.dim_c1
        self.dim_res3 = round_width(self.dim_res2, 2.0, divisor=8)
        self.dim_res4 = round_width(self.dim_res3, 2.0, divisor=8)
        self.dim_res5 = round_width(self.dim_res4, 2.0, divisor=8)
        self.block_basis = [[1, self.dim_res2, 2], [2, self.dim_res3, 2], [5, self.dim_res4, 2], [3, self.dim_res5, 2]]
        w_mul, d_mul = x.WIDTH_FACTOR, x.DEPTH_FACTOR
        dim_res1 = round_width(self.dim_c1, w_mul)
        from .stems import VideoModelStem
        self.s1 = VideoModelStem(dim_in=cfg.DATA.INPUT_CHANNEL_NUM, dim_out=[dim_res1], kernel=[[5, 3, 3]],
                                 stride=[[1, 2, 2]], padding=[[2, 1, 1]], norm_module=self.norm_module,
                                 stem_func_name="x3d_stem")
        dim_in = dim_res1
        for stage, block in enumerate(self.block_basis):
            dim_out = round_width(block[1], w_mul)
            dim_inner = int(x.BOTTLENECK_FACTOR * dim_out)
            n_rep = int(math.ceil(d_mul * block[0])) if d_mul else block[0]
            s = ResStage(dim_in=[dim_in], dim_out=[dim_out], dim_inner=[dim_inner], temp_kernel_sizes=[[3]],
                         stride=[block[2]], num_blocks=[n_rep],
                         num_groups=[dim_inner] if x.CHANNELWISE_3x3x3 else [cfg.RESNET.NUM_GROUPS],
                         num_block_temp_kernel=[n_rep], nonlocal_inds=cfg.NONLOCAL.LOCATION[0],
                         nonlocal_group=cfg.NONLOCAL.GROUP[0], nonlocal_pool=cfg.NONLOCAL.POOL[0],
                         instantiation=cfg.NONLOCAL.INSTANTIATION, trans_func_name=cfg.RESNET.TRANS_FUNC,
                         stride_1x1=cfg.RESNET.STRIDE_1X1, norm_module=self.norm_module,
                         dilation=cfg.RESNET.SPATIAL_DILATIONS[stage])
            dim_in = dim_out
            self.add_module(f"s{stage + 2}", s)
        spat = int(math.ceil(cfg.DATA.TRAIN_CROP_SIZE / 32.0))
        self.head = X3DHead(dim_in=dim_out, dim_inner=dim_inner, dim_out=x.DIM_C5, num_classes=cfg.MODEL.NUM_CLASSES,
                            pool_size=[cfg.DATA.NUM_FRAMES, spat, spat], dropout_rate=cfg.MODEL.DROPOUT_RATE,
                            act_func=cfg.MODEL.HEAD_ACT, bn_lin5_on=x.BN_LIN5)
        init_weights(self, cfg.MODEL.FC_INIT_STD, cfg.RESNET.ZERO_INIT_FINAL_BN)

    def forward(self, x, bboxes=None):
        from .video_models import _bump_batches_tracked, hold_notifications, num_splits_of, run_in_splits
        if self.training:
            _bump_batches_tracked(self)
            S = num_splits_of(self, full_batch=(self.head,))
            # only the backbone's parameters receive S contributions; the head (plain BatchNorm3d in the reference too)
            # runs once on the re-interleaved features of the whole batch
            hold_notifications(1)
            if S > 1:                    # SubBatchNorm3d: S sub-batch passes over the backbone (batchnorm.run_in_splits)
                backbone = [p for m in (self.s1, self.s2, self.s3, self.s4, self.s5) for p in m.parameters()]
                feats = run_in_splits(self, lambda xs: self._backbone(xs)[0], list(x), S, params=backbone)
                return self.head([feats])
        return self.head(self._backbone(x))

    def _backbone(self, x):
        x = self.s1(list(x))
        for i, s in enumerate((self.s2, self.s3, self.s4, self.s5)):
            x = s(x)
            if i < 3:                        # stage boundaries: backward segments of step.TrainStep (identity otherwise)
                x = engine.cut(x)
        return x


from . import stems as _stems  # noqa: E402
_stems._STEMS["x3d_stem"] = X3DStem
X3DTransform._block_fn = X3DBlockFn
