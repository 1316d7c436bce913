"""Forward/backward schedules of the MViT path on the token-space kernels (tokens.py).

One ``torch.autograd.Function`` per reference block: PatchEmbed (+ cls token), MultiScaleBlock (LayerNorm ->
pooled attention with relative positions and residual pooling -> skip pool / projection -> LayerNorm -> Mlp) and
the final LayerNorm on the cls rows.  Inside a Function nothing goes through ATen math or the autograd tape except
tiny parameter reshuffles; parameter gradients are written straight into ``param.grad`` (fp32) and announced to
``grad_ready`` listeners like the ResNet engine does.

Reference graph: slowfast/models/attention.py:293-392 (MultiScaleAttention.forward), :491-514
(MultiScaleBlock.forward), :13-45 (attention_pool), :64-147 (relative positions), common.py:25-34 (Mlp.forward),
stem_helper.py:315-320 (PatchEmbed.forward), video_model_builder.py:1166-1244 (MViT.forward).
"""
import math
import os

import torch

from . import engine, tokens
from .engine import _grad_dest, _notify

_f16 = torch.float16


class LinearUnit:
    """nn.Linear parameter container bound to the GEMM kernels: fp16 operand caches + gradient writes."""

    def __init__(self, lin):
        self.lin = lin
        self._key, self._w, self._wt = None, None, None

    def _ops(self, fresh=False):
        w = self.lin.weight
        key = (w.data_ptr(), w._version, w.device)
        if (fresh and engine.FORCE_WEIGHT_PREP) or self._key != key:
            wd = w.detach()
            self._w = wd.to(_f16)                      # [N, K]  forward operand
            self._wt = wd.t().contiguous().to(_f16)    # [K, N]  data-gradient operand
            self._key = key
        return self._w, self._wt

    def forward(self, x, resid=None, out=None):
        w, _ = self._ops(fresh=True)
        return tokens.gemm(x, w, bias=self.lin.bias, resid=resid, out=out)

    def forward_gelu(self, x):
        """(pre-activation, gelu(pre-activation)) with the GELU in the GEMM epilogue."""
        w, _ = self._ops(fresh=True)
        return tokens.gemm_gelu(x, w, bias=self.lin.bias)

    def backward_through_gelu(self, x, dy, h):
        """Like backward(), but returns d(loss)/d(h) for x = gelu(h): the data gradient times gelu'(h) in one GEMM."""
        lin = self.lin
        if lin.weight.requires_grad:
            dw, zero_first = _grad_dest(lin.weight)
            tokens.linear_wgrad(x, dy, dw, zero_first=zero_first)
        if lin.bias is not None and lin.bias.requires_grad:
            db, zero_first = _grad_dest(lin.bias)
            tokens.bias_grad(dy, db, accumulate=not zero_first)
        _, wt = self._ops()
        return tokens.gemm_gelu_grad(dy, wt, h)

    def backward(self, x, dy, need_dx=True, resid=None, out=None):
        """weight/bias gradients into .grad; returns dx (+ resid)."""
        lin = self.lin
        if lin.weight.requires_grad:
            dw, zero_first = _grad_dest(lin.weight)
            tokens.linear_wgrad(x, dy, dw, zero_first=zero_first)
        if lin.bias is not None and lin.bias.requires_grad:
            db, zero_first = _grad_dest(lin.bias)
            tokens.bias_grad(dy, db, accumulate=not zero_first)
        if not need_dx:
            return None
        _, wt = self._ops()
        return tokens.gemm(dy, wt, resid=resid, out=out)

    def params(self):
        return [p for p in (self.lin.weight, self.lin.bias) if p is not None]


class NormUnit:
    """nn.LayerNorm parameter container bound to the LayerNorm kernels."""

    def __init__(self, ln):
        self.ln = ln

    def forward(self, x):
        return tokens.layernorm_fwd(x, self.ln.weight, self.ln.bias, self.ln.eps)

    def backward(self, dy, x, mean, rstd, resid=None):
        ln = self.ln
        dg, zg = _grad_dest(ln.weight)
        db, zb = _grad_dest(ln.bias)
        assert zg == zb
        return tokens.layernorm_bwd(dy, x, ln.weight, mean, rstd, dg, db, resid=resid, accumulate=not zg)

    def params(self):
        return [self.ln.weight, self.ln.bias]


def _rel_index(q_size, k_size, device):
    """dist table of cal_rel_pos_spatial / cal_rel_pos_temporal (attention.py:76-88, 123-130) as int32."""
    q_ratio = max(k_size / q_size, 1.0)
    k_ratio = max(q_size / k_size, 1.0)
    dist = torch.arange(q_size)[:, None] * q_ratio - torch.arange(k_size)[None, :] * k_ratio
    dist += (k_size - 1) * k_ratio
    return dist.long().to(torch.int32).contiguous().to(device)


class AttentionPlan:
    """Shapes and cached index tables of one MultiScaleAttention call at a given (B, thw)."""

    def __init__(self, att, B, thw, device):
        self.B, self.thw = B, tuple(thw)
        self.heads, self.att = att.num_heads, att.dim_out
        self.D = self.att // self.heads
        cls = att.has_cls_embed
        self.cls = int(cls)
        C = self.att
        # blocks without a q (k / v) pooling conv leave that tensor as the qkv projection produced it
        # (attention.py:199-203, 236-262: MViTv1 blocks outside POOL_Q_STRIDE, plain ViT blocks)
        self.gq = self.gk = None
        if att.pool_q is not None:
            kq, sq, pq = tuple(att.pool_q.kernel_size), tuple(att.pool_q.stride), tuple(att.pool_q.padding)
            self.gq = tokens.DwGeom(B, C, self.D, thw, kq, sq, pq, cls)
        if att.pool_k is not None:
            kk, sk, pk = tuple(att.pool_k.kernel_size), tuple(att.pool_k.stride), tuple(att.pool_k.padding)
            self.gk = tokens.DwGeom(B, C, self.D, thw, kk, sk, pk, cls)
        self.q_thw = self.gq.out_thw if self.gq is not None else tuple(thw)
        self.k_thw = self.gk.out_thw if self.gk is not None else tuple(thw)
        self.Nq = self.cls + math.prod(self.q_thw)
        self.Nk = self.cls + math.prod(self.k_thw)
        self.lds = (self.Nk + 31) // 32 * 32     # score-row pitch: zero pad columns, a whole number of 32-wide GEMM K steps
        self.rel = att.rel_pos_spatial and att.rel_pos_temporal
        if att.rel_pos_spatial != att.rel_pos_temporal:
            raise NotImplementedError("spatial and temporal relative positions are used together on this path (MViTv2)")
        rows = (att.rel_pos_h.shape[0], att.rel_pos_w.shape[0], att.rel_pos_t.shape[0]) if self.rel else (0, 0, 0)
        self.desc = tokens.attn_desc(B, self.heads, self.D, cls, self.q_thw, self.k_thw, *rows)
        self.idx = None
        self.onehot = None          # fused attention: bias-bucket indicator matrix (tokens.attn_onehot), built on first use
        if self.rel:
            (qt, qh, qw), (kt, kh, kw) = self.q_thw, self.k_thw
            assert rows == (2 * max(qh, kh) - 1, 2 * max(qw, kw) - 1, 2 * max(qt, kt) - 1), \
                "rel-pos tables are used at their constructed size (no interpolation on this path)"
            self.idx = (_rel_index(qh, kh, device), _rel_index(qw, kw, device), _rel_index(qt, kt, device))


# Measured on MI355X: the first version of the fused kernels (bias lookups, expf, per-chunk rescale, no query split in
# the dK/dV kernel) lost to the unfused chain, 414 vs 445 clips/s (profiles/r1_visit12_bench_mvit_*.json); with the
# bias on the matrix cores, exp2, lazy rescale and the query split it wins, 488 vs 439 (profiles/r1_visit13_*).
_FUSED_DEFAULT = "1"
# GELU in the fc1 GEMM epilogue / gelu' in the fc2 data-gradient epilogue (SF_GELU_FUSED=0: separate elementwise passes)
_FUSED_GELU = os.environ.get("SF_GELU_FUSED", "1") != "0"


def _fused_attention(plan):
    """The flash-style kernels (sf_attn_*) cover head dims 32/64/96/128 and key grids with kH + kW + kT <= 48;
    SF_ATTN_FUSED=0 selects the unfused GEMM / softmax / GEMM chain (kept for A/B runs and other shapes)."""
    if os.environ.get("SF_ATTN_FUSED", _FUSED_DEFAULT) == "0":
        return False
    kt, kh, kw = plan.k_thw
    return plan.D % 32 == 0 and plan.D <= 128 and (not plan.rel or kt + kh + kw <= 48)


def attention_forward(att, plan, qkv):
    """qkv [B, N, 3*att] -> (o [B, Nq, att], saved tensors).  attention.py:318-385."""
    B, heads, D, C = plan.B, plan.heads, plan.D, plan.att
    Nq, Nk, lds = plan.Nq, plan.Nk, plan.lds
    q_in, k_in, v_in = qkv[..., 0:C], qkv[..., C:2 * C], qkv[..., 2 * C:3 * C]
    qp = kp = vp = None
    mq = rq_ = mk = rk_ = mv = rv_ = None
    if plan.gq is not None:
        qp = tokens.dwconv_fwd(q_in, att.pool_q.weight, plan.gq).view(B, Nq, C)
        qn, mq, rq_ = att._norm_q.forward(qp.view(B * Nq * heads, D))
        qn = qn.view(B, Nq, C)
    else:
        qn = q_in                                   # channel slice of qkv (row pitch 3C): used in place
    if plan.gk is not None:
        kp = tokens.dwconv_fwd(k_in, att.pool_k.weight, plan.gk).view(B, Nk, C)
        vp = tokens.dwconv_fwd(v_in, att.pool_v.weight, plan.gk).view(B, Nk, C)
        kn, mk, rk_ = att._norm_k.forward(kp.view(B * Nk * heads, D))
        vn, mv, rv_ = att._norm_v.forward(vp.view(B * Nk * heads, D))
        kn, vn = kn.view(B, Nk, C), vn.view(B, Nk, C)
    else:
        kn, vn = k_in, v_in
    tables = (att.rel_pos_h, att.rel_pos_w, att.rel_pos_t) if plan.rel else None
    t16 = t16t = None
    if plan.rel:
        t16, t16t = tokens.relpos_tables16(tables)
    rq = tokens.relpos_fwd(plan.desc, qn, tables, plan.idx, t16=t16) if plan.rel else None
    if _fused_attention(plan):
        if plan.rel and plan.onehot is None:
            plan.onehot = tokens.attn_onehot(plan.desc, qkv.device)
        o, lse = tokens.attn_fwd(plan.desc, qn, kn, vn, att.scale, rq, att.residual_pooling, onehot=plan.onehot)
        saved = dict(qp=qp, kp=kp, vp=vp, qn=qn, kn=kn, vn=vn, sq=(mq, rq_), sk=(mk, rk_), sv=(mv, rv_), t16t=t16t,
                     fused=(o, lse, rq))
        return o, saved
    if plan.gq is None or plan.gk is None:
        # the unfused chain (A/B runs, head dims the fused kernels do not cover) addresses contiguous operands
        qn, kn, vn = qn.contiguous(), kn.contiguous(), vn.contiguous()
    S = torch.empty((B, heads, Nq, lds), dtype=_f16, device=qkv.device)
    tokens.bgemm_heads(qn, (Nq * C, D), Nq, D, C, kn, (Nk * C, D), Nk, C, S, (heads * Nq * lds, Nq * lds), lds, B, heads)
    P = tokens.softmax_fwd(plan.desc, S, att.scale, rq)
    vt = tokens.transpose_heads(vn, B, Nk, heads, D, lds)
    o = torch.empty((B, Nq, C), dtype=_f16, device=qkv.device)
    resid = qn if att.residual_pooling else None
    tokens.bgemm_heads(P, (heads * Nq * lds, Nq * lds), Nq, lds, lds, vt, (heads * D * lds, D * lds), D, lds,
                       o, (Nq * C, D), C, B, heads, resid=resid, r_strides=(Nq * C, D), ldr=C, resid_row0=plan.cls)
    saved = dict(qp=qp, kp=kp, vp=vp, qn=qn, kn=kn, vn=vn, sq=(mq, rq_), sk=(mk, rk_), sv=(mv, rv_), P=P, t16t=t16t)
    return o, saved


def attention_backward(att, plan, qkv, sv, do):
    """d(o) -> d(qkv) [B, N, 3*att]; writes the gradients of pool_{q,k,v}, norm_{q,k,v}, rel_pos_{h,w,t}."""
    B, heads, D, C = plan.B, plan.heads, plan.D, plan.att
    Nq, Nk, lds = plan.Nq, plan.Nk, plan.lds
    qn, kn, vn = sv["qn"], sv["kn"], sv["vn"]
    dev = do.device
    if "fused" in sv:
        o, lse, rq = sv["fused"]
        dqkv = torch.empty(qkv.shape, dtype=_f16, device=dev)
        # gradients of un-pooled q / k / v ARE slices of d(qkv): the kernels write them in place (row pitch 3C)
        dq_out = dqkv[..., 0:C] if plan.gq is None and not plan.rel else None
        dkv_out = (dqkv[..., C:2 * C], dqkv[..., 2 * C:3 * C]) if plan.gk is None else None
        dqn, dkn, dvn, drq = tokens.attn_bwd(plan.desc, qn, kn, vn, att.scale, rq, att.residual_pooling, o, do, lse,
                                             onehot=plan.onehot, dq_out=dq_out, dkv_out=dkv_out)
        return _attention_backward_tail(att, plan, qkv, sv, qn, dqn, dkn, dvn, drq, dqkv)
    P = sv["P"]
    if plan.gq is None or plan.gk is None:
        qn, kn, vn = qn.contiguous(), kn.contiguous(), vn.contiguous()
    # dP = dO V^T ; dV = P^T dO
    dP = torch.empty((B, heads, Nq, lds), dtype=_f16, device=dev)
    tokens.bgemm_heads(do, (Nq * C, D), Nq, D, C, vn, (Nk * C, D), Nk, C, dP, (heads * Nq * lds, Nq * lds), lds, B, heads)
    dvn = torch.empty((B, Nk, C), dtype=_f16, device=dev)
    tokens.bgemm_tn_heads(P, (heads * Nq * lds, Nq * lds), lds, do, (Nq * C, D), C, Nq, Nk, D, dvn, (Nk * C, D), C,
                          B, heads)
    dS, drq = tokens.softmax_bwd(plan.desc, dP, P, att.scale, want_drq=plan.rel)      # dS already carries `scale`
    # dQ = dS K (+ dO on the non-cls rows: residual pooling) ; dK = dS^T Q
    kt = tokens.transpose_heads(kn, B, Nk, heads, D, lds)
    dqn = torch.empty((B, Nq, C), dtype=_f16, device=dev)
    resid = do if att.residual_pooling else None
    tokens.bgemm_heads(dS, (heads * Nq * lds, Nq * lds), Nq, lds, lds, kt, (heads * D * lds, D * lds), D, lds,
                       dqn, (Nq * C, D), C, B, heads, resid=resid, r_strides=(Nq * C, D), ldr=C, resid_row0=plan.cls)
    dkn = torch.empty((B, Nk, C), dtype=_f16, device=dev)
    tokens.bgemm_tn_heads(dS, (heads * Nq * lds, Nq * lds), lds, qn, (Nq * C, D), C, Nq, Nk, D, dkn, (Nk * C, D), C,
                          B, heads)
    return _attention_backward_tail(att, plan, qkv, sv, qn, dqn, dkn, dvn, drq)


def _attention_backward_tail(att, plan, qkv, sv, qn, dqn, dkn, dvn, drq, dqkv=None):
    """rel-pos table gradients (+ their dq term), LayerNorm(head_dim) and depthwise pooling backward.  Tensors that were
    not pooled skip both: their gradient is (or is copied into) the matching slice of d(qkv)."""
    B, heads, D, C = plan.B, plan.heads, plan.D, plan.att
    Nq, Nk = plan.Nq, plan.Nk
    dev = dqn.device
    if plan.rel:
        tabs = (att.rel_pos_h, att.rel_pos_w, att.rel_pos_t)
        dests = [_grad_dest(t) for t in tabs]
        tokens.relpos_bwd(plan.desc, qn, tabs, plan.idx, drq, dqn, [d[0] for d in dests], [not d[1] for d in dests],
                          t16t=sv["t16t"])
    if dqkv is None:
        dqkv = torch.empty(qkv.shape, dtype=_f16, device=dev)
    work = []
    if plan.gq is not None:                                  # LayerNorm(head_dim) backward, then the pooling conv
        dqp = att._norm_q.backward(dqn.view(-1, D), sv["qp"].view(-1, D), *sv["sq"]).view(B, Nq, C)
        work.append((0, dqp, att.pool_q, plan.gq))
    elif dqn.data_ptr() != dqkv.data_ptr():
        dqkv[..., 0:C].copy_(dqn)
    if plan.gk is not None:
        dkp = att._norm_k.backward(dkn.view(-1, D), sv["kp"].view(-1, D), *sv["sk"]).view(B, Nk, C)
        dvp = att._norm_v.backward(dvn.view(-1, D), sv["vp"].view(-1, D), *sv["sv"]).view(B, Nk, C)
        work += [(1, dkp, att.pool_k, plan.gk), (2, dvp, att.pool_v, plan.gk)]
    elif dkn.data_ptr() != dqkv[..., C:2 * C].data_ptr():
        dqkv[..., C:2 * C].copy_(dkn)
        dqkv[..., 2 * C:3 * C].copy_(dvn)
    # depthwise pooling backward into the slices of d(qkv)
    for i, dy, pool, geom in work:
        x_in = qkv[..., i * C:(i + 1) * C]
        tokens.dwconv_dgrad(dy.view(-1, C), pool.weight, geom, out=dqkv[..., i * C:(i + 1) * C])
        dw, zero_first = _grad_dest(pool.weight)
        tokens.dwconv_wgrad(x_in, dy.view(-1, C), geom, dw, zero_first=zero_first)
    return dqkv


class MultiScaleBlockFn(torch.autograd.Function):
    """MultiScaleBlock.forward (attention.py:491-514): conv pooling (or none), cls token, dimension change before the
    attention residual (DIM_MUL_IN_ATT, MViTv2) or after the Mlp (MViTv1)."""

    @staticmethod
    def forward(ctx, x, mod, thw, drop, *params):
        att = mod.attn
        B, N, dim = x.shape
        plan = mod._plan(B, thw, x.device)
        xn, m1, r1 = mod._norm1.forward(x)
        qkv = att._qkv.forward(xn)
        o, sv = attention_forward(att, plan, qkv)
        proj_first = mod._proj is not None and mod.dim_mul_in_att
        proj_last = mod._proj is not None and not mod.dim_mul_in_att
        if proj_first:
            xs = mod._proj.forward(xn)                     # dim change on the normed input (attention.py:494-495)
        else:
            xs = x
        pool = None
        if mod.pool_skip is not None:
            k, s, p = mod.pool_skip.kernel_size, mod.pool_skip.stride, mod.pool_skip.padding
            xres, arg, _ = tokens.token_pool_fwd(xs, B, thw, k, s, p, cls=mod.has_cls_embed)
            pool = (k, s, p, arg, xres)
        else:
            xres = xs
        if drop is None:
            x1 = att._proj.forward(o, resid=xres)          # x_res + attention output
        else:                                              # x_res + drop_path(attention output), attention.py:500-502
            x1 = tokens.row_scale_add(att._proj.forward(o), drop[0], xres.shape[1], resid=xres)
        xn2, m2, r2 = mod._norm2.forward(x1)
        if _FUSED_GELU:
            h, a = mod.mlp._fc1.forward_gelu(xn2)
        else:
            h = mod.mlp._fc1.forward(xn2)
            a = tokens.gelu_fwd(h)
        # MViTv1 (DIM_MUL_IN_ATT False): the dimension change acts on the normed Mlp input (attention.py:507-508)
        xb = mod._proj.forward(xn2) if proj_last else x1
        if drop is None:
            out = mod.mlp._fc2.forward(a, resid=xb)
        else:                                              # x + drop_path(mlp), attention.py:508-510
            out = tokens.row_scale_add(mod.mlp._fc2.forward(a), drop[1], xb.shape[1], resid=xb)
        ctx.drop = drop
        ctx.mod, ctx.plan, ctx.thw = mod, plan, tuple(thw)
        ctx.sv = dict(x=x, xn=xn, s1=(m1, r1), qkv=qkv, att=sv, o=o, pool=pool, x1=x1, xn2=xn2, s2=(m2, r2), h=h, a=a)
        ctx.out_thw = plan.q_thw
        return out

    @staticmethod
    def backward(ctx, dout):
        mod, plan, sv = ctx.mod, ctx.plan, ctx.sv
        att = mod.attn
        B = plan.B
        dout = dout.contiguous() if dout.dtype == _f16 else dout.to(_f16).contiguous()
        drop = ctx.drop
        # Mlp
        dbr = dout if drop is None else tokens.row_scale_add(dout, drop[1], dout.shape[1])
        if _FUSED_GELU:
            dh = mod.mlp._fc2.backward_through_gelu(sv["a"], dbr, sv["h"])
        else:
            da = mod.mlp._fc2.backward(sv["a"], dbr)
            dh = tokens.gelu_bwd(sv["h"], da)
        dxn2 = mod.mlp._fc1.backward(sv["xn2"], dh)
        proj_first = mod._proj is not None and mod.dim_mul_in_att
        if mod._proj is not None and not proj_first:       # the skip path went through proj(norm2(x1))
            dxn2 = mod._proj.backward(sv["xn2"], dout, resid=dxn2)
            dx1 = mod._norm2.backward(dxn2, sv["x1"], *sv["s2"])
        else:
            dx1 = mod._norm2.backward(dxn2, sv["x1"], *sv["s2"], resid=dout)
        # attention output projection, attention core, qkv projection
        do = att._proj.backward(sv["o"], dx1 if drop is None else tokens.row_scale_add(dx1, drop[0], dx1.shape[1]))
        dqkv = attention_backward(att, plan, sv["qkv"], sv["att"], do)
        dxn = att._qkv.backward(sv["xn"], dqkv)
        # skip path
        dxs = dx1
        if sv["pool"] is not None:
            k, s, p, arg, xres = sv["pool"]
            dxs = tokens.token_pool_bwd(dx1, xres, arg, B, ctx.thw, k, s, p, xres.shape[-1], cls=mod.has_cls_embed)
        if proj_first:
            dxn = mod._proj.backward(sv["xn"], dxs, resid=dxn)
            dx_skip = None
        else:
            dx_skip = dxs
        dx = mod._norm1.backward(dxn, sv["x"], *sv["s1"], resid=dx_skip)
        _notify(mod._param_list)
        ctx.sv = None
        return (dx, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)


class PatchEmbedFn(torch.autograd.Function):
    """PatchEmbed conv (+bias) -> tokens, with the cls token prepended (stem_helper.py:315-320,
    video_model_builder.py:1180-1186)."""

    @staticmethod
    def forward(ctx, x, mod, cls_token, pos, *params):
        unit = mod._unit
        xcl = unit.prepare_input(x)
        y, _ = unit.forward(xcl, None, mod.training)            # (B, C, T, H, W) channels-last == (B, THW, C) rows
        B, C, T, H, W = y.shape
        tok = y.permute(0, 2, 3, 4, 1).reshape(B, T * H * W, C)
        p16 = pos.detach().to(_f16) if pos is not None else None   # [1, cls + THW, C]: x += pos_embed (fp16 storage)
        if cls_token is not None:
            out = torch.empty((B, 1 + T * H * W, C), dtype=_f16, device=y.device)
            c16 = cls_token.detach().view(1, C).to(_f16)
            if p16 is None:
                out[:, 0] = c16
                out[:, 1:] = tok
            else:                                               # the copy that prepends the cls row becomes an add
                out[:, 0] = c16 + p16[0, :1]
                torch.add(tok, p16[:, 1:], out=out[:, 1:])
        else:
            out = tok if p16 is None else tok + p16
        ctx.mod, ctx.xcl, ctx.cls, ctx.yshape, ctx.has_pos = mod, xcl, cls_token, tuple(y.shape), pos is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        mod, cls_token = ctx.mod, ctx.cls
        unit = mod._unit
        B, C, T, H, W = ctx.yshape
        dout = dout.to(_f16)
        if cls_token is not None:
            if cls_token.requires_grad:
                g, zero_first = _grad_dest(cls_token)
                s = dout[:, 0].float().sum(0).view_as(g)
                g.copy_(s) if zero_first else g.add_(s)
            dtok = dout[:, 1:].contiguous()
        else:
            dtok = dout.contiguous()
        dy = dtok.view(B, T, H, W, C).permute(0, 4, 1, 2, 3)
        unit.backward(ctx.xcl, None, dy, need_dx=False)
        if unit.conv.bias is not None and unit.conv.bias.requires_grad:
            db, zero_first = _grad_dest(unit.conv.bias)
            tokens.bias_grad(dtok.view(-1, C), db, accumulate=not zero_first)
        _notify(unit.params() + ([cls_token] if cls_token is not None else []))
        ctx.xcl = None
        # d(pos_embed) = sum over the batch of d(tokens), fp32; autograd routes it to the (separate) embedding tables
        dpos = dout.float().sum(0, keepdim=True) if ctx.has_pos and ctx.needs_input_grad[3] else None
        return (None, None, None, dpos) + (None,) * (len(ctx.needs_input_grad) - 4)


class ClsNormFn(torch.autograd.Function):
    """Final LayerNorm on what the head consumes (video_model_builder.py:1226-1238): the cls rows
    (norm(x)[:, 0] == norm(x[:, 0])), or with USE_MEAN_POOLING the mean of the patch tokens (mean first, then norm)."""

    @staticmethod
    def forward(ctx, x, mod, mean_pool, *params):
        unit = mod._norm_unit
        if mean_pool:
            xc = x[:, 1:].float().mean(1).to(_f16)
        else:
            xc = x[:, 0].contiguous()
        y, m, r = unit.forward(xc)
        ctx.unit, ctx.xc, ctx.st, ctx.shape, ctx.mean_pool = unit, xc, (m, r), tuple(x.shape), mean_pool
        return y

    @staticmethod
    def backward(ctx, dy):
        dxc = ctx.unit.backward(dy.to(_f16).contiguous(), ctx.xc, *ctx.st)
        if ctx.mean_pool:
            B, N, C = ctx.shape
            dx = torch.empty(ctx.shape, dtype=_f16, device=dy.device)
            dx[:, 0] = 0
            dx[:, 1:] = (dxc.float() / (N - 1)).to(_f16)[:, None, :]
        else:
            dx = torch.zeros(ctx.shape, dtype=_f16, device=dy.device)
            dx[:, 0] = dxc
        _notify(ctx.unit.params())
        return (dx, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)
