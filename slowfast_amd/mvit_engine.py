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

from . import lib as _sflib

from . import engine, tokens
from . import engine as _engine
from .engine import _grad_dest, _notify, param_grads

_f16 = _sflib.act_dtype()        # fp16, or bf16 under SF_ACT_DTYPE=bf16 (lib.ACT_MODE)

# pool_k and pool_v of a block in ONE launch per direction (tokens.dwconv_*_pair).  Measured in the step
# (profiles/r6_v27_pair_kv_ab.txt): +1.2 % on MViT-B-16x4, whose blocks mostly pool k / v only -- the two chains then run back to
# back on one stream -- and -0.5 % on MViTv2-S, where every block also pools q: the k / v chains already hide behind the longer q
# chain on a second stream (engine.run_branches) and one 768-workgroup launch disturbs it more than two of 384.  So: "auto" =
# paired only in blocks without a q pooling branch; SF_PAIR_KV=1 always, 0 never (A/B runs).
PAIR_KV = os.environ.get("SF_PAIR_KV", "auto")
# SF_BIAS_FROM_POOL=0: the qkv bias gradient from a pass over d(qkv) (sf_colsum) instead of from the column sums the pooling data
# gradients leave (A/B runs; profiles/r6_v31_bias_from_pool_ab.txt)
BIAS_FROM_POOL = os.environ.get("SF_BIAS_FROM_POOL", "1") != "0"


def _pair_kv(plan):
    return PAIR_KV == "1" or (PAIR_KV not in ("0", "1") and plan.gq is None)
# fc1's bias gradient from the epilogue of the GEMM that produces d(loss)/d(fc1 output) (tokens.gemm_gelu_grad(dbias=...))
# instead of a separate column-sum pass over it; SF_FUSE_COLSUM=0 keeps the pass (A/B)
FUSE_COLSUM = os.environ.get("SF_FUSE_COLSUM", "1") != "0"


# fp32 side copy of the residual stream: the reference adds the two branch outputs of a MultiScaleBlock to an fp32 stream
# (attention.py:500-510; autocast does not touch additions).  Here the stream is 16-bit storage with the class-token row of every
# residual sum -- the only row the classifier reads -- ALSO kept in fp32: the GEMM epilogue that adds the residual sums those rows
# from its fp32 accumulators (sf_gemm_rows32), the LayerNorms that read the stream normalise them from the fp32 copy.  The 16-bit
# rows are the rounded fp32 rows, so every other consumer (skip pooling, backward) is unchanged.
# Measured on MI355X (profiles/r4/r4_v1_mvit_resid32_ab.txt), MViTv2-S 16x224^2 full-size logits against the fp32 oracle:
#   SF_MVIT_RESID32=0     16-bit stream only              1.192e-3   549.0 clips/s
#   (default)             class-token rows                9.05e-4    547.9
#   SF_MVIT_RESID32=full  + every row of the last stage   9.09e-4    545-546   (no measurable gain for 77 MB of fp32 rows: opt-in)
RESID32 = os.environ.get("SF_MVIT_RESID32", "1") != "0"
RESID32_FULL = os.environ.get("SF_MVIT_RESID32", "1") == "full"


class ResidSide:
    """The fp32 side rows that travel with the token tensor between MultiScaleBlocks: ``cls32`` [B, C] (the class-token rows)
    or ``full32`` [B, N, C] (every row; last stage).  Forward-only data: gradients flow through the 16-bit tensor."""

    def __init__(self, cls32=None, full32=None):
        self.cls32, self.full32 = cls32, full32

    def reader(self, N):
        """``side`` argument of a kernel that READS the stream tensor [B, N, C]: (period, fp32 rows)."""
        return (1, self.full32) if self.full32 is not None else (N, self.cls32)

    def cls_rows(self):
        return self.cls32 if self.full32 is None else self.full32[:, 0].contiguous()


def _sum_side(src, B, N, C, full, device):
    """``side`` argument of a kernel that WRITES a residual sum [B, N, C] of the stream: (period, fp32 residual rows or None,
    fp32 result rows) and the ResidSide the sum travels on with.  src: the ResidSide of the residual operand (None: the operand
    has no fp32 copy, e.g. it is a Linear output)."""
    if full:
        dst = torch.empty((B, N, C), dtype=torch.float32, device=device)
        return (1, src.full32 if src is not None else None, dst), ResidSide(full32=dst)
    dst = torch.empty((B, C), dtype=torch.float32, device=device)
    return (N, src.cls_rows() if src is not None else None, dst), ResidSide(cls32=dst)


def _bias_from_parts(dy, db, accumulate, parts):
    """db (+)= column sums of dy [.., n * C]: slice i from ``parts[i]`` ([rows, 2, C] tables whose slot 0 holds partial column sums,
    tokens.dwconv_dgrad(sums=True)) where given, from a pass over that slice (sf_colsum) otherwise."""
    if not parts or all(p is None for p in parts):
        tokens.bias_grad(dy, db, accumulate=accumulate)
        return
    C = dy.shape[-1] // len(parts)
    for i, part in enumerate(parts):
        dst = db[i * C:(i + 1) * C]
        if part is None:
            tokens.bias_grad(dy[..., i * C:(i + 1) * C], dst, accumulate=accumulate)
        else:
            tokens.colsum_finalize(part, C, C, dst, None, 1.0, accumulate)


class LinearUnit:
    """nn.Linear parameter container bound to the GEMM kernels: fp16 operand caches + gradient writes."""

    def __init__(self, lin):
        self.lin = lin
        self._key, self._w, self._wt, self._planned = None, None, None, False

    def _ops(self, fresh=False):
        w = self.lin.weight
        key = (w.data_ptr(), w._version, w.device, _engine.PARAM_EPOCH)
        if (fresh and engine.FORCE_WEIGHT_PREP and not self._planned) or self._key != key:
            wd = w.detach()
            self._w = wd.to(_f16)                      # [N, K]  forward operand
            self._wt = wd.t().contiguous().to(_f16)    # [K, N]  data-gradient operand
            self._key = key
            N, K = w.shape
            if _engine.PACK_RECORD is not None and N % 32 == 0 and K % 32 == 0 and w.is_contiguous():
                _engine.PACK_RECORD.append((self, None, True))
        return self._w, self._wt

    # -- engine.WeightPackPlan protocol: the Linear weight is the 1x1x1 convolution case of the packing kernel --------------
    def pack_item(self, geom, need_dgrad):
        from . import ops
        w = self.lin.weight
        N, K = w.shape
        g = ops.ConvGeom((1, K, 1, 1, 1), N, 1)
        assert g.ldf == K and g.ldd == N
        return (w, g.desc(K, N), torch.empty((N, K), dtype=_f16, device=w.device),
                torch.empty((K, N), dtype=_f16, device=w.device))

    def pack_assign(self, geom, wf, wd):
        w = self.lin.weight
        self._w, self._wt, self._planned = wf, wd, True
        self._key = (w.data_ptr(), w._version, w.device, _engine.PARAM_EPOCH)

    def forward(self, x, resid=None, out=None, side=None):
        w, _ = self._ops(fresh=True)
        return tokens.gemm(x, w, bias=self.lin.bias, resid=resid, out=out, side=side)

    def forward_gelu(self, x):
        """(pre-activation, gelu(pre-activation)) with the GELU in the GEMM epilogue."""
        w, _ = self._ops(fresh=True)
        return tokens.gemm_gelu(x, w, bias=self.lin.bias)

    def backward_through_gelu(self, x, dy, h, consumer_bias=None, bias_done=False):
        """Like backward(), but returns (d(loss)/d(h), bias_done) for x = gelu(h): the data gradient times gelu'(h) in one
        GEMM.  ``consumer_bias``: the bias parameter of the Linear that produced h (fc1); the returned flag = its gradient was
        taken in this GEMM's epilogue.  ``bias_done``: this layer's own bias gradient comes from elsewhere (bias_sum_dest)."""
        lin = self.lin
        if lin.weight.requires_grad:
            dw, zero_first = _grad_dest(lin.weight)
            tokens.linear_wgrad(x, dy, dw, zero_first=zero_first)
        if lin.bias is not None and lin.bias.requires_grad and not bias_done:
            db, zero_first = _grad_dest(lin.bias)
            tokens.bias_grad(dy, db, accumulate=not zero_first)
        _, wt = self._ops()
        if consumer_bias is not None and consumer_bias.requires_grad and FUSE_COLSUM:
            # the result IS d(loss)/d(output of the Linear in front of the GELU): its bias gradient = the column sums of the tile
            # the epilogue stores (one launch instead of a pass over the widest gradient tensor of the block)
            dcb, zf = _grad_dest(consumer_bias)
            return tokens.gemm_gelu_grad(dy, wt, h, dbias=dcb, accumulate=not zf), True
        return tokens.gemm_gelu_grad(dy, wt, h), False

    def backward(self, x, dy, need_dx=True, resid=None, out=None, bias_done=False):
        """weight/bias gradients into .grad; returns dx (+ resid).  ``bias_done``: the producer of dy already wrote the bias
        gradient (backward_through_gelu(consumer_bias=...))."""
        lin = self.lin
        if lin.weight.requires_grad:
            dw, zero_first = _grad_dest(lin.weight)
            tokens.linear_wgrad(x, dy, dw, zero_first=zero_first)
        if lin.bias is not None and lin.bias.requires_grad and not bias_done:
            # (the stand-alone bias pass -- the qkv Linear's: d(qkv) is written by three pooling kernels -- on a second stream beside
            # these GEMMs costs more in fork / join than the 40 us it hides: 729.9 -> 725.3 clips/s, profiles/r6_v30_bias_side_ab.txt)
            db, zero_first = _grad_dest(lin.bias)
            _bias_from_parts(dy, db, not zero_first, getattr(dy, "_sf_bias_parts", None))
        if not need_dx:
            return None
        _, wt = self._ops()
        return tokens.gemm(dy, wt, resid=resid, out=out)

    def bias_sum_dest(self):
        """(fp32 [N] destination, accumulate) for a kernel that produces the column sums of this layer's output gradient as a
        by-product (tokens.layernorm_bwd(sums=...)), or None when the bias takes no gradient."""
        b = self.lin.bias
        if b is None or not b.requires_grad:
            return None
        dest, zero_first = _grad_dest(b)
        return dest, not zero_first

    def params(self):
        return [p for p in (self.lin.weight, self.lin.bias) if p is not None]


class QKVUnit:
    """SEPARATE_QKV (attention.py:188-191, 310-314): three nn.Linear layers of the same input, run as ONE GEMM against
    the row-concatenated fp16 operand [3*att, dim] (the same arithmetic as the fused ``qkv`` Linear); the weight / bias
    gradients go to the three parameters from channel slices of d(qkv), the data gradient is again one GEMM."""

    def __init__(self, q, k, v):
        self.lins = (q, k, v)
        self._key, self._w, self._wt = None, None, None

    def _ops(self, fresh=False):
        key = tuple((lin.weight.data_ptr(), lin.weight._version, lin.weight.device) for lin in self.lins) + (_engine.PARAM_EPOCH,)
        if (fresh and engine.FORCE_WEIGHT_PREP) or self._key != key:
            w = torch.cat([lin.weight.detach() for lin in self.lins], 0)
            self._w = w.to(_f16)                       # [3*att, dim]  forward operand
            self._wt = w.t().contiguous().to(_f16)     # [dim, 3*att]  data-gradient operand
            self._key = key
        return self._w, self._wt

    def forward(self, x):
        w, _ = self._ops(fresh=True)
        bias = None
        if self.lins[0].bias is not None:
            bias = torch.cat([lin.bias.detach() for lin in self.lins])
        return tokens.gemm(x, w, bias=bias)

    def backward(self, x, dqkv):
        C = self.lins[0].out_features
        for i, lin in enumerate(self.lins):
            dy = dqkv[..., i * C:(i + 1) * C]
            if lin.weight.requires_grad:
                dw, zero_first = _grad_dest(lin.weight)
                tokens.linear_wgrad(x, dy, dw, zero_first=zero_first)
            if lin.bias is not None and lin.bias.requires_grad:
                db, zero_first = _grad_dest(lin.bias)
                parts = getattr(dqkv, "_sf_bias_parts", None)
                _bias_from_parts(dy, db, not zero_first, None if parts is None else [parts[i]])
        _, wt = self._ops()
        return tokens.gemm(dqkv, wt)

    def params(self):
        return [p for lin in self.lins for p in (lin.weight, lin.bias) if p is not None]


class NormUnit:
    """nn.LayerNorm parameter container bound to the LayerNorm kernels."""

    def __init__(self, ln):
        self.ln = ln

    def forward(self, x, side=None):
        return tokens.layernorm_fwd(x, self.ln.weight, self.ln.bias, self.ln.eps, side=side)

    def backward(self, dy, x, mean, rstd, resid=None, sums=None):
        ln = self.ln
        dg, zg = _grad_dest(ln.weight)
        db, zb = _grad_dest(ln.bias)
        assert zg == zb
        return tokens.layernorm_bwd(dy, x, ln.weight, mean, rstd, dg, db, resid=resid, accumulate=not zg, sums=sums)

    def params(self):
        return [self.ln.weight, self.ln.bias]


def _rel_index(q_size, k_size, device):
    """dist table of cal_rel_pos_spatial / cal_rel_pos_temporal (attention.py:76-88, 123-130) as int32."""
    q_ratio = max(k_size / q_size, 1.0)
    k_ratio = max(q_size / k_size, 1.0)
    dist = torch.arange(q_size)[:, None] * q_ratio - torch.arange(k_size)[None, :] * k_ratio
    dist += (k_size - 1) * k_ratio
    return dist.long().to(torch.int32).contiguous().to(device)


def _interp_matrix(stored, used, device):
    """W [used, stored] with get_rel_pos(table, used) == W @ table; None when the table is used as stored."""
    if stored == used:
        return None
    eye = torch.eye(stored, dtype=torch.float32)
    w = torch.nn.functional.interpolate(eye.reshape(1, stored, stored).permute(0, 2, 1), size=used, mode="linear")
    return w.reshape(stored, used).permute(1, 0).contiguous().to(device)


class AttentionPlan:
    """Shapes and cached index tables of one MultiScaleAttention call at a given (B, thw)."""

    def __init__(self, att, B, thw, device):
        self.B, self.thw = B, tuple(thw)
        self.heads, self.att = att.num_heads, att.dim_out
        self.D = self.att // self.heads
        cls = att.has_cls_embed
        self.cls = int(cls)
        # POOL_FIRST pools the block input (dim channels = heads x dim / heads) before the q / k / v Linears
        # (attention.py:236-241); otherwise the pooling acts on the projected q / k / v (att channels)
        self.dim_in = att.dim_in
        C = self.dim_in if att.pool_first else self.att
        Cw = C // self.heads
        # blocks without a q (k / v) pooling conv leave that tensor as the qkv projection produced it
        # (attention.py:199-203, 236-262: MViTv1 blocks outside POOL_Q_STRIDE, plain ViT blocks)
        self.gq = self.gk = None
        if att.pool_q is not None:
            kq, sq, pq = tuple(att.pool_q.kernel_size), tuple(att.pool_q.stride), tuple(att.pool_q.padding)
            self.gq = tokens.DwGeom(B, C, Cw, thw, kq, sq, pq, cls)
        if att.pool_k is not None:
            kk, sk, pk = tuple(att.pool_k.kernel_size), tuple(att.pool_k.stride), tuple(att.pool_k.padding)
            self.gk = tokens.DwGeom(B, C, Cw, thw, kk, sk, pk, cls)
        self.q_thw = self.gq.out_thw if self.gq is not None else tuple(thw)
        self.k_thw = self.gk.out_thw if self.gk is not None else tuple(thw)
        self.Nq = self.cls + math.prod(self.q_thw)
        self.Nk = self.cls + math.prod(self.k_thw)
        self.lds = (self.Nk + 31) // 32 * 32     # score-row pitch: zero pad columns, a whole number of 32-wide GEMM K steps
        self.rel = att.rel_pos_spatial and att.rel_pos_temporal
        if att.rel_pos_spatial != att.rel_pos_temporal:
            raise NotImplementedError("spatial and temporal relative positions are used together on this path (MViTv2)")
        (qt, qh, qw), (kt, kh, kw) = self.q_thw, self.k_thw
        rows = (2 * max(qh, kh) - 1, 2 * max(qw, kw) - 1, 2 * max(qt, kt) - 1) if self.rel else (0, 0, 0)
        self.desc = tokens.attn_desc(B, self.heads, self.D, cls, self.q_thw, self.k_thw, *rows)
        self.idx = None
        self.onehot = None          # fused attention: bias-bucket indicator matrix (tokens.attn_onehot), built on first use
        self.interp = (None, None, None)
        if self.rel:
            # get_rel_pos (attention.py:48-61): a table constructed with another row count (size // stride at construction
            # vs the pooling conv's ceil(size / stride): odd extents, e.g. MViTv2-L at 312^2) is resampled along its rows by
            # F.interpolate(mode="linear") -- a fixed linear map, kept as a [rows used, rows stored] fp32 matrix
            self.interp = tuple(_interp_matrix(t.shape[0], r, device)
                                for t, r in zip((att.rel_pos_h, att.rel_pos_w, att.rel_pos_t), rows))
            self.idx = (_rel_index(qh, kh, device), _rel_index(qw, kw, device), _rel_index(qt, kt, device))

    def tables(self, att):
        """rel_pos_{h,w,t} at the row counts this call uses (resampled where they were constructed differently)."""
        return tuple(t if w is None else w @ t.detach()
                     for t, w in zip((att.rel_pos_h, att.rel_pos_w, att.rel_pos_t), self.interp))


# Measured on MI355X: the first version of the fused kernels (bias lookups, expf, per-chunk rescale, no query split in
# the dK/dV kernel) lost to the unfused chain, 414 vs 445 clips/s (profiles/r1/r1_visit12_bench_mvit_*.json); with the
# bias on the matrix cores, exp2, lazy rescale and the query split it wins, 488 vs 439 (profiles/r1/r1_visit13_*).
_FUSED_DEFAULT = "1"
# GELU in the fc1 GEMM epilogue / gelu' in the fc2 data-gradient epilogue (SF_GELU_FUSED=0: separate elementwise passes)
_FUSED_GELU = os.environ.get("SF_GELU_FUSED", "1") != "0"
# bias gradients of mlp.fc2 / attn.proj from the column sums norm2's backward takes in passing (SF_LN_BIAS_SUMS=0: separate passes)
_LN_BIAS_SUMS = os.environ.get("SF_LN_BIAS_SUMS", "1") != "0"


def _fused_attention(plan):
    """The flash-style kernels (sf_attn_*) cover head dims 32/64/96/128 and key grids with kH + kW + kT <= 48;
    SF_ATTN_FUSED=0 selects the unfused GEMM / softmax / GEMM chain (kept for A/B runs and other shapes)."""
    if os.environ.get("SF_ATTN_FUSED", _FUSED_DEFAULT) == "0":
        return False
    kt, kh, kw = plan.k_thw
    return plan.D % 32 == 0 and plan.D <= 128 and (not plan.rel or kt + kh + kw <= 48)


def _relpos_forward(att, plan, qn):
    """The q . rel_pos_{h,w,t} contractions (attention.py:48-147): a function of q alone -> (rq, packed tables for the backward);
    (None, None) without relative positions.  attention_forward runs it inside the q branch (engine.run_branches)."""
    if not plan.rel:
        return None, None
    tables = plan.tables(att)
    t16, t16t = tokens.relpos_tables16(tables)
    return tokens.relpos_fwd(plan.desc, qn, tables, plan.idx, t16=t16), t16t


def _core_forward(att, plan, qn, kn, vn, rel=None):
    """softmax(q*scale k^T + rel-pos bias) v (+ q on the non-cls rows: residual pooling), attention.py:355-385.
    qn / kn / vn: [B, N*, att] token tensors of any row pitch -> (o [B, Nq, att], saved tensors of the core).
    ``rel``: what _relpos_forward(att, plan, qn) returned, when the caller already ran it."""
    B, heads, D, C = plan.B, plan.heads, plan.D, plan.att
    Nq, Nk, lds = plan.Nq, plan.Nk, plan.lds
    rq, t16t = rel if rel is not None else _relpos_forward(att, plan, qn)
    if _fused_attention(plan):
        if plan.rel and plan.onehot is None:
            plan.onehot = tokens.attn_onehot(plan.desc, qn.device)
        o, lse = tokens.attn_fwd(plan.desc, qn, kn, vn, att.scale, rq, att.residual_pooling, onehot=plan.onehot)
        return o, dict(qn=qn, kn=kn, vn=vn, t16t=t16t, fused=(o, lse, rq))
    if rows_not_dense(qn) or rows_not_dense(kn) or rows_not_dense(vn):
        # the unfused chain (A/B runs, head dims the fused kernels do not cover) addresses contiguous operands
        qn, kn, vn = qn.contiguous(), kn.contiguous(), vn.contiguous()
    S = torch.empty((B, heads, Nq, lds), dtype=_f16, device=qn.device)
    tokens.bgemm_heads(qn, (Nq * C, D), Nq, D, C, kn, (Nk * C, D), Nk, C, S, (heads * Nq * lds, Nq * lds), lds, B, heads)
    P = tokens.softmax_fwd(plan.desc, S, att.scale, rq)
    vt = tokens.transpose_heads(vn, B, Nk, heads, D, lds)
    o = torch.empty((B, Nq, C), dtype=_f16, device=qn.device)
    resid = qn if att.residual_pooling else None
    tokens.bgemm_heads(P, (heads * Nq * lds, Nq * lds), Nq, lds, lds, vt, (heads * D * lds, D * lds), D, lds,
                       o, (Nq * C, D), C, B, heads, resid=resid, r_strides=(Nq * C, D), ldr=C, resid_row0=plan.cls)
    return o, dict(qn=qn, kn=kn, vn=vn, P=P, t16t=t16t)


def rows_not_dense(x):
    return x.stride(-2) != x.shape[-1]


def _core_backward(att, plan, core, do, dq_out=None, dkv_out=None, defer_rel=False):
    """d(o) -> (dq, dk, dv) of the attention core, each [B, N*, att]; writes the gradients of rel_pos_{h,w,t} and adds
    their term to dq.  dq_out / dkv_out: destinations the fused kernels may write in place (slices of d(qkv)).
    ``defer_rel``: returns (dq, dk, dv, drq) WITHOUT the rel-pos part; the caller runs _relpos_backward(att, plan, core, drq, dq)
    before it uses dq (attention_backward does so inside its q branch)."""
    B, heads, D, C = plan.B, plan.heads, plan.D, plan.att
    Nq, Nk, lds = plan.Nq, plan.Nk, plan.lds
    qn, kn, vn = core["qn"], core["kn"], core["vn"]
    dev = do.device
    if "fused" in core:
        o, lse, rq = core["fused"]
        dqn, dkn, dvn, drq = tokens.attn_bwd(plan.desc, qn, kn, vn, att.scale, rq, att.residual_pooling, o, do, lse,
                                             onehot=plan.onehot, dq_out=dq_out, dkv_out=dkv_out)
    else:
        P = core["P"]
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
    if defer_rel:
        return dqn, dkn, dvn, drq
    _relpos_backward(att, plan, core, drq, dqn)
    return dqn, dkn, dvn


def _relpos_backward(att, plan, core, drq, dqn):
    """Gradient of the q . rel_pos_{h,w,t} contractions: the tables' gradients, and their term added to dq in place."""
    qn = core["qn"]
    if plan.rel:
        params = (att.rel_pos_h, att.rel_pos_w, att.rel_pos_t)
        dests = [_grad_dest(t) for t in params]
        # a resampled table receives its gradient through the transposed interpolation map
        tmp = [None if w is None else torch.empty((w.shape[0], plan.D), dtype=torch.float32, device=dqn.device)
               for w in plan.interp]
        tokens.relpos_bwd(plan.desc, qn, [d[0] if t is None else t for d, t in zip(dests, tmp)], plan.idx, drq, dqn,
                          [d[0] if t is None else t for d, t in zip(dests, tmp)],
                          [(not d[1]) and t is None for d, t in zip(dests, tmp)], t16t=core["t16t"])
        for (g, zero_first), t, w in zip(dests, tmp, plan.interp):
            if w is not None:
                g.copy_(w.t() @ t) if zero_first else g.add_(w.t() @ t)


def _pool_norm(x_in, pool, norm_unit, geom, B, N, C, D):
    """attention_pool (attention.py:13-45): depthwise conv over the non-cls tokens, LayerNorm over the head channels."""
    xp = tokens.dwconv_fwd(x_in, pool.weight, geom).view(B, N, C)
    xn, m, r = norm_unit.forward(xp.view(-1, D))
    return xp, xn.view(B, N, C), (m, r)


def attention_forward(att, plan, qkv):
    """qkv [B, N, 3*att] -> (o [B, Nq, att], saved tensors).  attention.py:318-385."""
    B, D, C = plan.B, plan.D, plan.att
    Nq, Nk = plan.Nq, plan.Nk
    q_in, k_in, v_in = qkv[..., 0:C], qkv[..., C:2 * C], qkv[..., 2 * C:3 * C]
    qp = kp = vp = sq = sk = sv_ = None

    rel = None

    def q_chain():
        qp_, qn_, sq_ = _pool_norm(q_in, att.pool_q, att._norm_q, plan.gq, B, Nq, C, D)
        return qp_, qn_, sq_, _relpos_forward(att, plan, qn_)      # the rel-pos contractions need q only: same branch

    def k_chain():
        return _pool_norm(k_in, att.pool_k, att._norm_k, plan.gk, B, Nk, C, D)

    def v_chain():
        return _pool_norm(v_in, att.pool_v, att._norm_v, plan.gk, B, Nk, C, D)

    def kv_chain():
        if _pair_kv(plan) and tokens.dwconv_pair_ok(plan.gk, tokens.rows_pitch(k_in)[2], C):
            # pool_k and pool_v: one geometry, adjacent channel slices of qkv -> ONE launch (tokens.dwconv_fwd_pair)
            kp_, vp_ = tokens.dwconv_fwd_pair(k_in, v_in, att.pool_k.weight, att.pool_v.weight, plan.gk)
            kn_, mk, rk = att._norm_k.forward(kp_.view(-1, D))
            vn_, mv, rv = att._norm_v.forward(vp_.view(-1, D))
            return (kp_.view(B, Nk, C), kn_.view(B, Nk, C), (mk, rk)), (vp_.view(B, Nk, C), vn_.view(B, Nk, C), (mv, rv))
        return k_chain(), v_chain()

    if plan.gq is not None and plan.gk is not None:
        # the q chain and the k / v chains read disjoint channel slices of qkv: two streams (engine.run_branches)
        (qp, qn, sq, rel), ((kp, kn, sk), (vp, vn, sv_)) = engine.run_branches([q_chain, kv_chain], qkv)
    else:
        if plan.gq is not None:
            qp, qn, sq, rel = q_chain()
        else:
            qn = q_in                               # channel slice of qkv (row pitch 3C): used in place
        if plan.gk is not None:
            (kp, kn, sk), (vp, vn, sv_) = kv_chain()
        else:
            kn, vn = k_in, v_in
    o, core = _core_forward(att, plan, qn, kn, vn, rel=rel)
    return o, dict(qp=qp, kp=kp, vp=vp, sq=sq, sk=sk, sv=sv_, core=core)


def attention_backward(att, plan, qkv, sv, do):
    """d(o) -> d(qkv) [B, N, 3*att]; writes the gradients of pool_{q,k,v}, norm_{q,k,v}, rel_pos_{h,w,t}."""
    B, D, C = plan.B, plan.D, plan.att
    Nq, Nk = plan.Nq, plan.Nk
    core = sv["core"]
    dqkv = torch.empty(qkv.shape, dtype=_f16, device=do.device)
    dq_out = dkv_out = None
    if "fused" in core:
        # gradients of un-pooled q / k / v ARE slices of d(qkv): the kernels write them in place (row pitch 3C)
        dq_out = dqkv[..., 0:C] if plan.gq is None and not plan.rel else None
        dkv_out = (dqkv[..., C:2 * C], dqkv[..., 2 * C:3 * C]) if plan.gk is None else None
    dqn, dkn, dvn, drq = _core_backward(att, plan, core, do, dq_out=dq_out, dkv_out=dkv_out, defer_rel=True)
    # LayerNorm(head_dim) and depthwise pooling backward.  Tensors that were not pooled skip both: their gradient is
    # (or is copied into) the matching slice of d(qkv).
    # column sums of the slices of d(qkv) (= the qkv bias gradient) that the pooling data gradients leave as a by-product: taken by
    # the qkv Linear's backward instead of a pass over d(qkv) (LinearUnit / QKVUnit.backward: dy._sf_bias_parts)
    bias_parts = [None, None, None]

    def pool_back(i, dy, pool, geom):                        # depthwise pooling backward into slice i of d(qkv)
        x_in = qkv[..., i * C:(i + 1) * C]
        res = tokens.dwconv_dgrad(dy.view(-1, C), pool.weight, geom, out=dqkv[..., i * C:(i + 1) * C], sums=BIAS_FROM_POOL)
        if BIAS_FROM_POOL:
            bias_parts[i] = res[1]
        dw, zero_first = _grad_dest(pool.weight)
        tokens.dwconv_wgrad(x_in, dy.view(-1, C), geom, dw, zero_first=zero_first)

    def q_chain():                                           # rel-pos, LayerNorm(head_dim) backward, then the pooling conv
        _relpos_backward(att, plan, core, drq, dqn)
        dqp = att._norm_q.backward(dqn.view(-1, D), sv["qp"].view(-1, D), *sv["sq"]).view(B, Nq, C)
        pool_back(0, dqp, att.pool_q, plan.gq)

    def k_chain():
        dkp = att._norm_k.backward(dkn.view(-1, D), sv["kp"].view(-1, D), *sv["sk"]).view(B, Nk, C)
        pool_back(1, dkp, att.pool_k, plan.gk)

    def v_chain():
        dvp = att._norm_v.backward(dvn.view(-1, D), sv["vp"].view(-1, D), *sv["sv"]).view(B, Nk, C)
        pool_back(2, dvp, att.pool_v, plan.gk)

    def kv_chain():
        k_in, v_in = qkv[..., C:2 * C], qkv[..., 2 * C:3 * C]
        if _pair_kv(plan) and tokens.dwconv_pair_ok(plan.gk, tokens.rows_pitch(k_in)[2], C):
            dkp = att._norm_k.backward(dkn.view(-1, D), sv["kp"].view(-1, D), *sv["sk"]).view(-1, C)
            dvp = att._norm_v.backward(dvn.view(-1, D), sv["vp"].view(-1, D), *sv["sv"]).view(-1, C)
            tokens.dwconv_dgrad_pair(dkp, dvp, att.pool_k.weight, att.pool_v.weight, plan.gk,
                                     out=dqkv[..., C:2 * C], out2=dqkv[..., 2 * C:3 * C])
            (dwk, zk), (dwv, zv) = _grad_dest(att.pool_k.weight), _grad_dest(att.pool_v.weight)
            tokens.dwconv_wgrad_pair(k_in, v_in, dkp, dvp, plan.gk, dwk, dwv, zero_first=zk, zero_first2=zv)
            return
        k_chain()
        v_chain()

    if plan.gq is not None and plan.gk is not None and not engine.GRADS_VIA_AUTOGRAD:
        engine.run_branches([q_chain, kv_chain], do)         # disjoint slices of d(qkv), disjoint parameters: two streams
    else:
        if plan.gq is not None:
            q_chain()
        else:
            _relpos_backward(att, plan, core, drq, dqn)
            if dqn.data_ptr() != dqkv.data_ptr():
                dqkv[..., 0:C].copy_(dqn)
        if plan.gk is not None:
            kv_chain()
        elif dkn.data_ptr() != dqkv[..., C:2 * C].data_ptr():
            dqkv[..., C:2 * C].copy_(dkn)
            dqkv[..., 2 * C:3 * C].copy_(dvn)
    if any(bp is not None for bp in bias_parts):
        dqkv._sf_bias_parts = bias_parts
    return dqkv


def attention_forward_pool_first(att, plan, xn):
    """POOL_FIRST (attention.py:296-301, 318-351): the normed block input xn [B, N, dim], read as heads x (dim / heads)
    channels, is pooled (+ LayerNorm over dim / heads) three times, and the q / k / v Linears run on the POOLED tokens
    (fewer GEMM rows for k / v).  -> (o [B, Nq, att], saved tensors)."""
    B, Ci = plan.B, plan.dim_in
    Dc = Ci // plan.heads
    Nq, Nk = plan.Nq, plan.Nk
    qp = kp = vp = sq = sk = sv_ = None
    xq = xk = xv = xn
    if plan.gq is not None:
        qp, xq, sq = _pool_norm(xn, att.pool_q, att._norm_q, plan.gq, B, Nq, Ci, Dc)
    if plan.gk is not None:
        kp, xk, sk = _pool_norm(xn, att.pool_k, att._norm_k, plan.gk, B, Nk, Ci, Dc)
        vp, xv, sv_ = _pool_norm(xn, att.pool_v, att._norm_v, plan.gk, B, Nk, Ci, Dc)
    qn, kn, vn = att._q.forward(xq), att._k.forward(xk), att._v.forward(xv)
    o, core = _core_forward(att, plan, qn, kn, vn)
    return o, dict(qp=qp, kp=kp, vp=vp, sq=sq, sk=sk, sv=sv_, xq=xq, xk=xk, xv=xv, core=core)


def attention_backward_pool_first(att, plan, xn, sv, do):
    """d(o) -> d(xn) [B, N, dim] of the POOL_FIRST attention: core, q / k / v Linears, LayerNorm(dim / heads), pooling
    convs; the three branches' input gradients are summed."""
    B, Ci = plan.B, plan.dim_in
    Dc = Ci // plan.heads
    dqn, dkn, dvn = _core_backward(att, plan, sv["core"], do)
    dxn = None
    for name, dy, unit, pool, norm, geom, N in (
            ("q", dqn, att._q, att.pool_q, att._norm_q, plan.gq, plan.Nq),
            ("k", dkn, att._k, att.pool_k, att._norm_k, plan.gk, plan.Nk),
            ("v", dvn, att._v, att.pool_v, att._norm_v, plan.gk, plan.Nk)):
        if geom is None:                                     # Linear on xn itself: chain the sum through the GEMM residual
            dxn = unit.backward(xn, dy, resid=dxn)
            continue
        dxp = unit.backward(sv["x" + name], dy)
        dpool = norm.backward(dxp.view(-1, Dc), sv[name + "p"].view(-1, Dc), *sv["s" + name]).view(B, N, Ci)
        g = tokens.dwconv_dgrad(dpool.view(-1, Ci), pool.weight, geom).view(xn.shape)
        dw, zero_first = _grad_dest(pool.weight)
        tokens.dwconv_wgrad(xn, dpool.view(-1, Ci), geom, dw, zero_first=zero_first)
        dxn = g if dxn is None else dxn.add_(g)
    return dxn


class MultiScaleBlockFn(torch.autograd.Function):
    """MultiScaleBlock.forward (attention.py:491-514): conv pooling (or none), cls token, dimension change before the
    attention residual (DIM_MUL_IN_ATT, MViTv2) or after the Mlp (MViTv1)."""

    @staticmethod
    def forward(ctx, x, mod, thw, drop, side, *params):
        """``side``: the ResidSide travelling with x (None: 16-bit stream only); it is UPDATED in place to the one of the
        block output."""
        engine.record_params(ctx, params)
        att = mod.attn
        B, N, dim = x.shape
        plan = mod._plan(B, thw, x.device)
        xn, m1, r1 = mod._norm1.forward(x, side=side.reader(N) if side is not None else None)
        if att.pool_first:
            qkv = None
            o, sv = attention_forward_pool_first(att, plan, xn)
        else:
            qkv = att._qkv.forward(xn)
            o, sv = attention_forward(att, plan, qkv)
        proj_first = mod._proj is not None and mod.dim_mul_in_att
        proj_last = mod._proj is not None and not mod.dim_mul_in_att
        rside = side                                       # fp32 rows of the residual operand of the first sum
        if proj_first:
            xs = mod._proj.forward(xn)                     # dim change on the normed input (attention.py:494-495)
            rside = None                                   # a Linear output: 16-bit only
        else:
            xs = x
        pool = None
        if mod.pool_skip is not None:
            k, s, p = mod.pool_skip.kernel_size, mod.pool_skip.stride, mod.pool_skip.padding
            xres, arg, _ = tokens.token_pool_fwd(xs, B, thw, k, s, p, cls=mod.has_cls_embed)
            pool = (k, s, p, arg, xres)
            if rside is not None:                          # the pooling passes the class-token row through: its fp32 copy stays valid
                rside = ResidSide(cls32=rside.cls_rows())
        else:
            xres = xs
        Nq, Ca = plan.Nq, xres.shape[-1]
        full = bool(getattr(mod, "_resid32_full", False)) and RESID32_FULL
        s1 = side1 = None
        if side is not None:
            s1, side1 = _sum_side(rside if (rside is None or not full or rside.full32 is not None) else None,
                                  B, Nq, Ca, full, x.device)
        if drop is None:
            x1 = att._proj.forward(o, resid=xres, side=s1)     # x_res + attention output
        else:                                              # x_res + drop_path(attention output), attention.py:500-502
            x1 = tokens.row_scale_add(att._proj.forward(o), drop[0], xres.shape[1], resid=xres, side=s1)
        xn2, m2, r2 = mod._norm2.forward(x1, side=side1.reader(Nq) if side1 is not None else None)
        if _FUSED_GELU:
            h, a = mod.mlp._fc1.forward_gelu(xn2)
        else:
            h = mod.mlp._fc1.forward(xn2)
            a = tokens.gelu_fwd(h)
        # MViTv1 (DIM_MUL_IN_ATT False): the dimension change acts on the normed Mlp input (attention.py:507-508)
        xb = mod._proj.forward(xn2) if proj_last else x1
        s2 = side2 = None
        if side is not None:
            s2, side2 = _sum_side(None if proj_last else side1, B, Nq, xb.shape[-1], full, x.device)
        if drop is None:
            out = mod.mlp._fc2.forward(a, resid=xb, side=s2)
        else:                                              # x + drop_path(mlp), attention.py:508-510
            out = tokens.row_scale_add(mod.mlp._fc2.forward(a), drop[1], xb.shape[1], resid=xb, side=s2)
        if side is not None:
            side.cls32, side.full32 = side2.cls32, side2.full32
        ctx.drop = drop
        ctx.mod, ctx.plan, ctx.thw = mod, plan, tuple(thw)
        ctx.sv = dict(x=x, xn=xn, s1=(m1, r1), qkv=qkv, att=sv, o=o, pool=pool, x1=x1, xn2=xn2, s2=(m2, r2), h=h, a=a)
        ctx.out_thw = plan.q_thw
        return out

    @staticmethod
    @engine.delivers_grads
    def backward(ctx, dout):
        # the ~9 column-sum finalizes of the block (LayerNorm affine and bias gradients) leave as one launch at the end
        with tokens.deferred_finalizes():
            dx = MultiScaleBlockFn._backward(ctx, dout)
        _notify(ctx.mod._param_list)
        return (dx, None, None, None, None) + param_grads(ctx, 5)

    @staticmethod
    def _backward(ctx, dout):
        mod, plan, sv = ctx.mod, ctx.plan, ctx.sv
        att = mod.attn
        B = plan.B
        dout = dout.contiguous() if dout.dtype == _f16 else dout.to(_f16).contiguous()
        drop = ctx.drop
        # Mlp
        dbr = dout if drop is None else tokens.row_scale_add(dout, drop[1], dout.shape[1])
        proj_first = mod._proj is not None and mod.dim_mul_in_att
        proj_last = mod._proj is not None and not proj_first
        # Without stochastic depth, norm2's backward reads d(block output) as its residual operand and writes d(x1): their column
        # sums ARE the bias gradients of mlp.fc2 and attn.proj -- taken from that pass instead of two column-sum passes (round 4)
        ln_sums = drop is None and not proj_last and _LN_BIAS_SUMS
        if _FUSED_GELU:
            dh, b1_done = mod.mlp._fc2.backward_through_gelu(sv["a"], dbr, sv["h"], consumer_bias=mod.mlp.fc1.bias,
                                                            bias_done=ln_sums)
        else:
            da = mod.mlp._fc2.backward(sv["a"], dbr, bias_done=ln_sums)
            dh, b1_done = tokens.gelu_bwd(sv["h"], da), False
        dxn2 = mod.mlp._fc1.backward(sv["xn2"], dh, bias_done=b1_done)
        if proj_last:                                      # the skip path went through proj(norm2(x1))
            dxn2 = mod._proj.backward(sv["xn2"], dout, resid=dxn2)
            dx1 = mod._norm2.backward(dxn2, sv["x1"], *sv["s2"])
        elif ln_sums:
            dx1 = mod._norm2.backward(dxn2, sv["x1"], *sv["s2"], resid=dout,
                                      sums=(mod.mlp._fc2.bias_sum_dest(), att._proj.bias_sum_dest()))
        else:
            dx1 = mod._norm2.backward(dxn2, sv["x1"], *sv["s2"], resid=dout)
        # attention output projection, attention core, qkv projection
        do = att._proj.backward(sv["o"], dx1 if drop is None else tokens.row_scale_add(dx1, drop[0], dx1.shape[1]),
                                bias_done=ln_sums)
        if att.pool_first:
            dxn = attention_backward_pool_first(att, plan, sv["xn"], sv["att"], do)
        else:
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
        ctx.sv = None
        return dx


def _attention_sub_forward(sub, plan, x):
    """AttentionSubBlock.forward (reversible_mvit.py:670-672): attention(norm(x)) up to (not including) the output
    projection -> (o, saved)."""
    att = sub.attn
    xn, m, r = sub._norm.forward(x)
    if att.pool_first:
        qkv = None
        o, sv = attention_forward_pool_first(att, plan, xn)
    else:
        qkv = att._qkv.forward(xn)
        o, sv = attention_forward(att, plan, qkv)
    return o, dict(x=x, xn=xn, st=(m, r), qkv=qkv, att=sv, o=o)


def _attention_sub_backward(sub, plan, sv, d_out, resid=None):
    """d(proj output) -> d(x) (+ resid): output projection, attention core, pooling, qkv projection, LayerNorm."""
    att = sub.attn
    do = att._proj.backward(sv["o"], d_out)
    if att.pool_first:
        dxn = attention_backward_pool_first(att, plan, sv["xn"], sv["att"], do)
    else:
        dxn = att._qkv.backward(sv["xn"], attention_backward(att, plan, sv["qkv"], sv["att"], do))
    return sub._norm.backward(dxn, sv["x"], *sv["st"], resid=resid)


def _mlp_sub_forward(sub, x):
    """MLPSubblock.forward (reversible_mvit.py:616-617) up to (not including) fc2 -> (gelu(fc1(norm(x))), saved)."""
    xn, m, r = sub._norm.forward(x)
    if _FUSED_GELU:
        h, a = sub.mlp._fc1.forward_gelu(xn)
    else:
        h = sub.mlp._fc1.forward(xn)
        a = tokens.gelu_fwd(h)
    return a, dict(x=x, xn=xn, st=(m, r), h=h, a=a)


def _mlp_sub_backward(sub, sv, d_out, resid=None):
    if _FUSED_GELU:
        dh, b1_done = sub.mlp._fc2.backward_through_gelu(sv["a"], d_out, sv["h"], consumer_bias=sub.mlp.fc1.bias)
    else:
        dh, b1_done = tokens.gelu_bwd(sv["h"], sub.mlp._fc2.backward(sv["a"], d_out)), False
    dxn = sub.mlp._fc1.backward(sv["xn"], dh, bias_done=b1_done)
    return sub._norm.backward(dxn, sv["x"], *sv["st"], resid=resid)


def _f16c(t):
    return t.contiguous() if t.dtype == _f16 else t.to(_f16).contiguous()


class RevBlockFn(torch.autograd.Function):
    """ReversibleBlock.forward (reversible_mvit.py:491-526): Y1 = X1 + drop(F(X2)), Y2 = X2 + drop(G(Y1)), the two
    streams kept as two token tensors (the reference concatenates them only to pass them through RevBackProp).  The
    reference re-derives X1 / X2 from Y1 / Y2 in backward to avoid storing them (:528-590); here the block inputs and
    the usual GEMM / pooling outputs are kept (288 GB of HBM; identical values, no recomputation pass)."""

    @staticmethod
    def forward(ctx, x1, x2, mod, thw, drop, *params):
        engine.record_params(ctx, params)
        B, N, _ = x2.shape
        plan = mod._plan(B, thw, x2.device)
        att = mod.F.attn
        o, sf = _attention_sub_forward(mod.F, plan, x2)
        if drop is None:
            y1 = att._proj.forward(o, resid=x1)
        else:
            y1 = tokens.row_scale_add(att._proj.forward(o), drop, N, resid=x1)
        a, sg = _mlp_sub_forward(mod.G, y1)
        if drop is None:
            y2 = mod.G.mlp._fc2.forward(a, resid=x2)
        else:
            y2 = tokens.row_scale_add(mod.G.mlp._fc2.forward(a), drop, N, resid=x2)
        ctx.mod, ctx.plan, ctx.drop, ctx.sf, ctx.sg = mod, plan, drop, sf, sg
        return y1, y2

    @staticmethod
    @engine.delivers_grads
    def backward(ctx, dy1, dy2):
        mod, plan, drop = ctx.mod, ctx.plan, ctx.drop
        dy1, dy2 = _f16c(dy1), _f16c(dy2)
        N = dy1.shape[1]
        dg = dy2 if drop is None else tokens.row_scale_add(dy2, drop, N)
        dy1t = _mlp_sub_backward(mod.G, ctx.sg, dg, resid=dy1)              # d(Y1) = dY1 + G'(Y1)^T d(G)
        df = dy1t if drop is None else tokens.row_scale_add(dy1t, drop, N)
        dx2 = _attention_sub_backward(mod.F, plan, ctx.sf, df, resid=dy2)  # d(X2) = dY2 + F'(X2)^T d(F)
        _notify(mod._param_list)
        ctx.sf = ctx.sg = None
        return (dy1t, dx2) + param_grads(ctx, 2)


class StageTransitionFn(torch.autograd.Function):
    """StageTransitionBlock.forward (reversible_mvit.py:350-409), PRE_Q_FUSION "avg", RES_PATH "conv": the mean of the
    two streams goes through attention with q pooling; the residual takes the attention's OWN pool_q conv + norm_q (and
    res_proj when the width changes, before the pooling or -- POOL_FIRST -- after it); then x + G(x), drop_path on the sum."""

    @staticmethod
    def forward(ctx, x1, x2, mod, thw, drop, *params):
        engine.record_params(ctx, params)
        B, N, C = x1.shape
        plan = mod._plan(B, thw, x1.device)
        att = mod.F.attn
        half = mod._half(B, x1.device)
        x = x1 if x2 is None else tokens.row_scale_add(x1, half, N, resid=tokens.row_scale_add(x2, half, N))
        o, sf = _attention_sub_forward(mod.F, plan, x)
        proj_before = mod._res_proj is not None and not att.pool_first
        xr = mod._res_proj.forward(x) if proj_before else x
        # the residual has the channel count the q pooling acts on in either order (dim_out after res_proj, dim under
        # POOL_FIRST), so it shares the attention's pooling geometry plan.gq
        rp, rn, rst = _pool_norm(xr, att.pool_q, att._norm_q, plan.gq, B, plan.Nq, xr.shape[-1], plan.gq.Cw)
        xres = mod._res_proj.forward(rn) if mod._res_proj is not None and not proj_before else rn
        xa = att._proj.forward(o, resid=xres)
        a, sg = _mlp_sub_forward(mod.G, xa)
        out = mod.G.mlp._fc2.forward(a, resid=xa)
        if drop is not None:                                   # drop_path on the block output (:407)
            out = tokens.row_scale_add(out, drop, out.shape[1])
        ctx.mod, ctx.plan, ctx.drop, ctx.sf, ctx.sg = mod, plan, drop, sf, sg
        ctx.res = dict(x=x, xr=xr, rp=rp, rn=rn, rst=rst, proj_before=proj_before, two=x2 is not None, N=N)
        return out

    @staticmethod
    @engine.delivers_grads
    def backward(ctx, dout):
        mod, plan, drop, res = ctx.mod, ctx.plan, ctx.drop, ctx.res
        att = mod.F.attn
        B = plan.B
        dout = _f16c(dout)
        if drop is not None:
            dout = tokens.row_scale_add(dout, drop, dout.shape[1])
        dxa = _mlp_sub_backward(mod.G, ctx.sg, dout, resid=dout)             # x + G(x)
        # residual path: [res_proj] <- norm_q <- pool_q <- [res_proj]; parameters shared with the attention accumulate
        dres = dxa
        if mod._res_proj is not None and not res["proj_before"]:
            dres = mod._res_proj.backward(res["rn"], dres)
        Cr, Dc = res["xr"].shape[-1], plan.gq.Cw
        drp = att._norm_q.backward(dres.view(-1, Dc), res["rp"].view(-1, Dc), *res["rst"]).view(B, plan.Nq, Cr)
        dxr = tokens.dwconv_dgrad(drp.view(-1, Cr), att.pool_q.weight, plan.gq).view(res["xr"].shape)
        dw, zero_first = _grad_dest(att.pool_q.weight)
        tokens.dwconv_wgrad(res["xr"], drp.view(-1, Cr), plan.gq, dw, zero_first=zero_first)
        dx_res = mod._res_proj.backward(res["x"], dxr) if res["proj_before"] else dxr
        dx = _attention_sub_backward(mod.F, plan, ctx.sf, dxa, resid=dx_res)
        _notify(mod._param_list)
        ctx.sf = ctx.sg = ctx.res = None
        if res["two"]:
            half = mod._half(B, dx.device)
            g1 = tokens.row_scale_add(dx, half, res["N"])
            return (g1, g1.clone()) + param_grads(ctx, 2)
        return (dx, None) + param_grads(ctx, 2)


class PatchEmbedFn(torch.autograd.Function):
    """PatchEmbed conv (+bias) -> tokens, with the cls token prepended (stem_helper.py:315-320,
    video_model_builder.py:1180-1186)."""

    @staticmethod
    def forward(ctx, x, mod, cls_token, pos, *params):
        engine.record_params(ctx, params)
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
    @engine.delivers_grads
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
        pg, (dcls,) = param_grads(ctx, 4, extra=(cls_token,)) if cls_token is not None else (param_grads(ctx, 4), (None,))
        return (None, None, dcls, dpos) + pg


class ClsNormFn(torch.autograd.Function):
    """Final LayerNorm on what the head consumes (video_model_builder.py:1226-1238).  mode "cls": the cls rows
    (norm(x)[:, 0] == norm(x[:, 0])); "mean_norm" (USE_MEAN_POOLING): the mean of the patch tokens, then norm;
    "norm_mean" (no cls token, the reference's default there): norm of every token, then the mean over tokens."""

    @staticmethod
    def forward(ctx, x, mod, mode, has_cls, side, *params):
        engine.record_params(ctx, params)
        unit = mod._norm_unit
        s = int(bool(has_cls))
        if mode == "norm_mean":
            B, N, C = x.shape
            xc = x
            yt, m, r = unit.forward(x.view(B * N, C))
            y = yt.view(B, N, C).float().mean(1).to(_f16)
        else:
            xc = x[:, s:].float().mean(1).to(_f16) if mode == "mean_norm" else x[:, 0].contiguous()
            # the class-token rows of the stream in fp32 (ResidSide) when the blocks kept them, else the 16-bit rows
            x32 = side.cls_rows() if (side is not None and mode == "cls") else xc.float()
            y, m, r = unit.forward(xc, side=(1, x32) if (side is not None and mode == "cls") else None)
            # what the classifier consumes is a [B, C] tensor: hand it over in fp32 (the row statistics come from the
            # kernel, the affine map of 32 x 768 values is free) instead of rounding the normalised features to fp16 right
            # before a cancelling sum over C -- the dominant term of the logits' deviation from the fp32 reference
            ln = unit.ln
            y = (x32 - m.view(-1, 1)) * r.view(-1, 1) * ln.weight.detach().float() + ln.bias.detach().float()
        ctx.unit, ctx.xc, ctx.st, ctx.shape, ctx.mode, ctx.s = unit, xc, (m, r), tuple(x.shape), mode, s
        return y

    @staticmethod
    @engine.delivers_grads
    def backward(ctx, dy):
        B, N, C = ctx.shape
        s = ctx.s
        if ctx.mode == "norm_mean":
            dyt = (dy.float() / N).to(_f16)[:, None, :].expand(B, N, C).contiguous()
            dx = ctx.unit.backward(dyt.view(B * N, C), ctx.xc.view(B * N, C), *ctx.st).view(B, N, C)
        else:
            dxc = ctx.unit.backward(dy.to(_f16).contiguous(), ctx.xc, *ctx.st)
            if ctx.mode == "mean_norm":
                dx = torch.empty(ctx.shape, dtype=_f16, device=dy.device)
                dx[:, :s] = 0
                dx[:, s:] = (dxc.float() / (N - s)).to(_f16)[:, None, :]
            else:
                dx = torch.zeros(ctx.shape, dtype=_f16, device=dy.device)
                dx[:, 0] = dxc
        _notify(ctx.unit.params())
        return (dx, None, None, None, None) + param_grads(ctx, 5)


class TokenNormFn(torch.autograd.Function):
    """The detection path's final LayerNorm (video_model_builder.py:1218-1224): norm of every token, cls row dropped;
    the (B, T*H*W, C) result IS the channels-last memory of the (B, C, T, H, W) feature map the RoI head consumes."""

    @staticmethod
    def forward(ctx, x, mod, has_cls, *params):
        engine.record_params(ctx, params)
        unit = mod._norm_unit
        s = int(bool(has_cls))
        B, N, C = x.shape
        xt = x[:, s:].contiguous() if s else x
        y, m, r = unit.forward(xt.view(B * (N - s), C))
        ctx.unit, ctx.xt, ctx.st, ctx.shape, ctx.s = unit, xt, (m, r), tuple(x.shape), s
        return y.view(B, N - s, C)

    @staticmethod
    @engine.delivers_grads
    def backward(ctx, dy):
        B, N, C = ctx.shape
        s = ctx.s
        dxt = ctx.unit.backward(_f16c(dy).view(-1, C), ctx.xt.view(-1, C), *ctx.st).view(B, N - s, C)
        if s:
            dx = torch.empty(ctx.shape, dtype=_f16, device=dy.device)
            dx[:, :1] = 0
            dx[:, 1:] = dxt
        else:
            dx = dxt
        _notify(ctx.unit.params())
        return (dx, None, None) + param_grads(ctx, 3)
