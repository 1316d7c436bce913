"""Tensor-level wrappers over the token-space C ABI (include/sfamd.h, second half): batched GEMMs, LayerNorm,
GELU, column sums, depthwise convolution, relative-position terms, softmax, transposes, token max-pool.

Token tensors are fp16 with unit stride in the last (channel) dimension and a uniform row pitch: [B, N, C]
contiguous, a channel slice of one, or a 2-D [M, C] view.  PyTorch is used for memory and streams only.
"""
from ctypes import byref, c_int32

import torch

from . import lib as _sflib

from .lib import AttnDesc, ColFinItem, DwDesc, Rows32, SfError, get_lib
from .ops import _ptr, _stream, _workspace

_f16 = _sflib.act_dtype()        # fp16, or bf16 under SF_ACT_DTYPE=bf16 (lib.ACT_MODE)


def rows_pitch(x):
    """(rows, C, pitch) of a token tensor; raises unless rows are uniformly spaced with unit channel stride."""
    if x.dtype != _f16 or x.dim() < 2:
        raise SfError(f"expected an fp16 token tensor, got {tuple(x.shape)} {x.dtype}")
    C = x.shape[-1]
    if C > 1 and x.stride(-1) != 1:
        raise SfError("token tensor must have unit channel stride")
    ld = x.stride(-2) if x.shape[-2] > 1 else max(C, x.stride(-2))
    rows = x.shape[-2]
    for d in range(x.dim() - 3, -1, -1):
        if x.shape[d] > 1 and x.stride(d) != rows * ld:
            raise SfError(f"token tensor rows are not uniformly spaced: shape {tuple(x.shape)} strides {x.stride()}")
        rows *= x.shape[d]
    if ld % 8 or C % 8 or x.data_ptr() % 16:
        raise SfError(f"token tensor needs C % 8 == 0, pitch % 8 == 0 and a 16-byte base (C={C}, pitch={ld})")
    return rows, C, ld


def _lib_call(name, *args, **kw):
    return get_lib().call(name, *args, **kw)


# ------------------------------------------------------------------------------------------------
# fp32 side rows of a token residual stream (include/sfamd.h: sf_rows32)
def _rows32(M, C, period, src=None, dst=None):
    """ctypes descriptor: rows m % period == 0 of an [M, C] token tensor carry an fp32 copy at row m // period of
    ``src`` (residual operand rows, optional) / ``dst`` (result rows), fp32 [M // period, C] each."""
    for t in (src, dst):
        if t is not None:
            assert t.dtype == torch.float32 and t.is_contiguous() and t.shape[-1] == C and t.numel() * period == M * C, \
                (tuple(t.shape), M, C, period)
    return Rows32(_ptr(src), _ptr(dst), C, period)


# ------------------------------------------------------------------------------------------------
# GEMMs
def gemm(a, w16, bias=None, resid=None, out=None, alpha=0.0, side=None):
    """out[m][n] = sum_k a[m][k] * w16[n][k] (+ bias[n]) (+ resid[m][n]).  a: token tensor (..., K); w16: fp16
    [N, K] (pitch = stride(0)); the nn.Linear forward with w16 = weight.half(), or its data gradient with
    w16 = weight.t().half().  ``side = (period, src32 | None, dst32)``: rows m % period == 0 are also summed in fp32 from the
    accumulators (+ src32 row, or the fp16 residual row when src32 is None) into dst32; their fp16 row is round(dst32 row)."""
    M, K, lda = rows_pitch(a)
    N = w16.shape[0]
    assert w16.dtype == _f16 and w16.shape[1] == K and w16.stride(1) == 1
    if out is None:
        out = torch.empty(a.shape[:-1] + (N,), dtype=_f16, device=a.device)
    Mo, No, ldy = rows_pitch(out)
    assert (Mo, No) == (M, N)
    ldr = 0
    if resid is not None:
        Mr, Nr, ldr = rows_pitch(resid)
        assert (Mr, Nr) == (M, N)
    if side is not None:
        assert alpha == 0.0
        d = _rows32(M, N, side[0], side[1], side[2])
        _lib_call("sf_gemm_rows32", M, N, K, a.data_ptr(), lda, w16.data_ptr(), w16.stride(0), _ptr(bias), _ptr(resid), ldr,
                  out.data_ptr(), ldy, byref(d), _stream(a),
                  work=dict(flops=2.0 * M * N * K, bytes=2.0 * (M * K + M * N + N * K)))
        return out
    _lib_call("sf_bgemm", M, N, K, a.data_ptr(), lda, w16.data_ptr(), w16.stride(0), _ptr(bias), _ptr(resid), ldr,
              out.data_ptr(), ldy, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, float(alpha), _stream(a),
              work=dict(flops=2.0 * M * N * K, bytes=2.0 * (M * K + M * N + N * K)))
    return out


def gemm_gelu(a, w16, bias=None):
    """fc1 of the Mlp with the GELU in the GEMM epilogue: returns (h = a w^T + bias, gelu(h))."""
    M, K, lda = rows_pitch(a)
    N = w16.shape[0]
    h = torch.empty(a.shape[:-1] + (N,), dtype=_f16, device=a.device)
    act = torch.empty_like(h)
    _lib_call("sf_gemm_act", M, N, K, a.data_ptr(), lda, w16.data_ptr(), w16.stride(0), _ptr(bias), h.data_ptr(), N, 1,
              act.data_ptr(), N, _stream(a), work=dict(flops=2.0 * M * N * K, bytes=2.0 * (M * K + 2 * M * N + N * K)))
    return h, act


def gemm_gelu_grad(dy, wt16, h, dbias=None, accumulate=False):
    """Data gradient of fc2 times gelu'(h): d(loss)/d(fc1 output) in one GEMM.  ``dbias``: fc1's bias gradient (the column
    sums of the result) is taken from the tiles the epilogue stores and folded into ``dbias`` -- no separate pass over dh."""
    M, K, lda = rows_pitch(dy)
    N = wt16.shape[0]
    assert rows_pitch(h)[:2] == (M, N)
    dh = torch.empty(h.shape, dtype=_f16, device=dy.device)
    work = dict(flops=2.0 * M * N * K, bytes=2.0 * (M * K + 2 * M * N + N * K))
    if dbias is None:
        _lib_call("sf_gemm_act", M, N, K, dy.data_ptr(), lda, wt16.data_ptr(), wt16.stride(0), None, dh.data_ptr(), N, 2,
                  h.data_ptr(), rows_pitch(h)[2], _stream(dy), work=work)
        return dh
    cap = (M + 127) // 128
    part = torch.empty((cap, 2, N), dtype=torch.float32, device=dy.device)
    nrows = c_int32(0)
    _lib_call("sf_gemm_act_colsum", M, N, K, dy.data_ptr(), lda, wt16.data_ptr(), wt16.stride(0), None, dh.data_ptr(), N, 2,
              h.data_ptr(), rows_pitch(h)[2], part.data_ptr(), cap, byref(nrows), _stream(dy), work=work)
    colsum_finalize(part[:nrows.value], N, N, dbias, None, 1.0, accumulate)
    return dh


def bgemm_heads(a, a_strides, M, K, lda, w, w_strides, N, ldw, out, o_strides, ldy, B, heads, resid=None,
                r_strides=(0, 0), ldr=0, resid_row0=0, alpha=0.0):
    """Per-(batch, head) GEMM: out[b,h][m][n] = sum_k a[b,h][m][k] * w[b,h][n][k] (+ resid).  *_strides = (per-batch,
    per-head) element strides of the operand bases."""
    _lib_call("sf_bgemm", M, N, K, a.data_ptr(), lda, w.data_ptr(), ldw, None, _ptr(resid), ldr, out.data_ptr(), ldy,
              B * heads, heads, a_strides[0], a_strides[1], w_strides[0], w_strides[1], o_strides[0], o_strides[1],
              r_strides[0], r_strides[1], resid_row0, float(alpha), _stream(a),
              work=dict(flops=2.0 * B * heads * M * N * K, bytes=2.0 * B * heads * (M * K + N * K + M * N)))
    return out


def bgemm_tn_heads(p, p_strides, ldp, x, x_strides, ldx, M, R, Kc, out, o_strides, ldo, B, heads, scale=1.0):
    """Per-(batch, head): out[b,h][r][c] = scale * sum_m p[b,h][m][r] * x[b,h][m][c]."""
    _lib_call("sf_bgemm_tn", M, R, Kc, p.data_ptr(), ldp, x.data_ptr(), ldx, out.data_ptr(), ldo, float(scale),
              B * heads, heads, p_strides[0], p_strides[1], x_strides[0], x_strides[1], o_strides[0], o_strides[1],
              _stream(p), work=dict(flops=2.0 * B * heads * M * R * Kc, bytes=2.0 * B * heads * (M * R + M * Kc + R * Kc)))
    return out


def linear_wgrad(x, dy, dw, zero_first=True):
    """dw[n][k] (+)= sum_m dy[m][n] * x[m][k] -- the nn.Linear weight gradient, through the conv weight-gradient
    kernels on a 1x1x1 geometry (fp32, deterministic split reduction)."""
    from . import ops
    M, K, ldx = rows_pitch(x)
    Md, N, ldy = rows_pitch(dy)
    assert Md == M and tuple(dw.shape) == (N, K) and dw.dtype == torch.float32 and dw.is_contiguous()
    geom = _linear_geom(M, K, N)
    lib = get_lib()
    d = geom.desc(ldx, ldy)
    if geom.ws_bytes is None:
        geom.ws_bytes = lib.call("sf_conv_wgrad_workspace", byref(d))
    ws = _workspace(x.device, geom.ws_bytes)
    lib.call("sf_conv_wgrad", byref(d), x.data_ptr(), None, None, 0, dy.data_ptr(), dw.data_ptr(), 1.0, int(zero_first),
             ws.data_ptr(), ws.numel(), _ptr(ops._wgrad_rowtab(geom, d, x.device)), _stream(x),
             work=dict(flops=2.0 * M * N * K, bytes=2.0 * M * (N + K)))
    return dw


_geoms = {}


def _linear_geom(M, K, N):
    from . import ops
    key = (M, K, N)
    g = _geoms.get(key)
    if g is None:
        g = _geoms[key] = ops.ConvGeom((1, K, 1, 1, M), N, (1, 1, 1))
    return g


# ------------------------------------------------------------------------------------------------
# LayerNorm / GELU / bias gradients
def layernorm_fwd(x, gamma, beta, eps, out=None, save_stats=True, side=None):
    """``side = (period, src32)``: rows m % period == 0 are normalised from their fp32 copy src32[m // period]."""
    M, C, ldx = rows_pitch(x)
    y = torch.empty(x.shape, dtype=_f16, device=x.device) if out is None else out
    _, _, ldy = rows_pitch(y)
    mean = torch.empty(M, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(M, dtype=torch.float32, device=x.device) if save_stats else None
    if side is not None:
        d = _rows32(M, C, side[0], side[1], None)
        _lib_call("sf_layernorm_fwd_rows32", M, C, x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), float(eps),
                  y.data_ptr(), ldy, _ptr(mean), _ptr(rstd), byref(d), _stream(x), work=dict(bytes=4.0 * M * C))
        return y, mean, rstd
    _lib_call("sf_layernorm_fwd", M, C, x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), float(eps), y.data_ptr(),
              ldy, _ptr(mean), _ptr(rstd), _stream(x), work=dict(bytes=4.0 * M * C))
    return y, mean, rstd


# Deferred finalizes (SF_FIN_BATCH=0: every finalize is its own launch, A/B runs).  Inside ``with deferred_finalizes():`` the
# column-sum finalizes of bias / LayerNorm-affine gradients are collected and issued as ONE sf_colsum_finalize_batch launch when
# the context exits (mvit_engine wraps the backward of a block in it: ~9 launches of 6-7 us become one).  The partial tables
# and outputs are kept alive by the pending list; two finalizes into the same output are never batched together.
import os as _os
_FIN_BATCH = _os.environ.get("SF_FIN_BATCH", "1") != "0"
_pending_fin = None         # list of (ColFinItem fields..., tensors kept alive) while a deferral context is open
_FIN_MAX_ROWS = 2048        # tables up to this many rows are batched (longer ones keep their own launch with its fold stage)


class deferred_finalizes:
    """Collect the column-sum finalizes issued inside the ``with`` block and launch them in batches at its end.  Assumes ONE
    thread per block backward (the autograd worker that runs the Function): the pending list is process-global, not
    thread-safe.  A flush launches on the CURRENT stream; items collected on another stream (engine.run_branches) make it wait
    for that stream first.  Any OTHER kernel that writes one of the
    pending outputs inside the block must call flush_finalizes() first (colsum_finalize does so for its own immediate path)."""

    def __enter__(self):
        global _pending_fin
        self._outer = _pending_fin
        if _FIN_BATCH and _pending_fin is None:
            _pending_fin = []
        return self

    def __exit__(self, *exc):
        global _pending_fin
        if self._outer is None and _pending_fin is not None:
            try:
                if exc[0] is None:
                    flush_finalizes()
            finally:
                _pending_fin = None
        return False


def flush_finalizes():
    """Issue the collected finalizes (one launch per 16)."""
    global _pending_fin
    items = _pending_fin
    if not items:
        return
    _pending_fin = []
    _finalize_batch(items)


def _finalize_batch(items):
    arr = (ColFinItem * len(items))()
    part0 = items[0][0]
    cur = torch.cuda.current_stream(part0.device) if part0.is_cuda else None
    for i, (part, C, fold, out0, out1, scale, accumulate, stride, st) in enumerate(items):
        arr[i] = ColFinItem(part.data_ptr(), part.shape[0], C, fold, _ptr(out0), _ptr(out1), float(scale), int(accumulate),
                            int(stride))
        # an item collected on ANOTHER stream (a branch of engine.run_branches) whose table may still be in production there:
        # the launching stream waits for it (a no-op once the branch has been joined) -- ADVICE r5
        if st is not None and cur is not None and st != cur:
            cur.wait_stream(st)
    _lib_call("sf_colsum_finalize_batch", arr, len(items), _stream(part0))


def colsum_finalize(part, C, fold, out0, out1, scale=1.0, accumulate=False, row_stride=1):
    """``part``: [rows, 2, C] view of the table (``row_stride`` > 1: the pairs of a wider [rows, 2 * row_stride, C] table, the view
    starting at the first pair wanted)."""
    if _pending_fin is not None and part.shape[0] <= _FIN_MAX_ROWS:
        outs = [o.data_ptr() for o in (out0, out1) if o is not None]
        for it in _pending_fin:                         # a second contribution to the same output waits for the first
            if any(o is not None and o.data_ptr() in outs for o in (it[3], it[4])):
                flush_finalizes()
                break
        _pending_fin.append((part, C, fold, out0, out1, scale, accumulate, row_stride,
                             torch.cuda.current_stream(part.device) if part.is_cuda else None))
        return
    if _pending_fin:
        # an immediate finalize (table too long to defer) into an output a deferred one also writes: launch order must stay
        # program order (overwrite-then-accumulate), so the pending items go first
        outs = [o.data_ptr() for o in (out0, out1) if o is not None]
        if any(o is not None and o.data_ptr() in outs for it in _pending_fin for o in (it[3], it[4])):
            flush_finalizes()
    if row_stride != 1:
        _finalize_batch([(part, C, fold, out0, out1, scale, accumulate, row_stride, None)])
        return
    _lib_call("sf_colsum_finalize", part.data_ptr(), part.shape[0], C, fold, _ptr(out0), _ptr(out1), float(scale),
              int(accumulate), _stream(part))


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, resid=None, accumulate=False, out=None, sums=None):
    """dx (+ resid); dgamma / dbeta written (or accumulated) in fp32.  ``sums = ((dest_resid, accumulate) | None,
    (dest_dx, accumulate) | None)``: the column sums of ``resid`` and of the stored result land in those fp32 [C] vectors -- the
    bias gradients of the Linear layers whose output gradient they are -- from the same pass."""
    M, C, ldx = rows_pitch(x)
    _, _, lddy = rows_pitch(dy)
    dx = torch.empty(x.shape, dtype=_f16, device=x.device) if out is None else out
    _, _, lddx = rows_pitch(dx)
    ldr = rows_pitch(resid)[2] if resid is not None else 0
    lib = get_lib()
    nblk = lib.call("sf_layernorm_bwd_blocks", M, C)
    if sums is not None and (sums[0] is not None or sums[1] is not None):
        assert sums[0] is None or resid is not None
        part = torch.empty((nblk, 4, C), dtype=torch.float32, device=x.device)
        lib.call("sf_layernorm_bwd_sums", M, C, dy.data_ptr(), lddy, x.data_ptr(), ldx, gamma.data_ptr(), mean.data_ptr(),
                 rstd.data_ptr(), _ptr(resid), ldr, dx.data_ptr(), lddx, part.data_ptr(), _stream(x),
                 work=dict(bytes=2.0 * M * C * (3 + int(resid is not None))))
        colsum_finalize(part[:, 0:2], C, C, dgamma, dbeta, 1.0, accumulate, row_stride=2)
        (d0, a0), (d1, a1) = (sums[0] or (None, False)), (sums[1] or (None, False))
        if d0 is not None and d1 is not None and a0 == a1:
            colsum_finalize(part[:, 2:4], C, C, d0, d1, 1.0, a0, row_stride=2)
        else:
            if d0 is not None:
                colsum_finalize(part[:, 2:4], C, C, d0, None, 1.0, a0, row_stride=2)
            if d1 is not None:
                colsum_finalize(part[:, 2:4], C, C, None, d1, 1.0, a1, row_stride=2)
        return dx
    part = torch.empty((nblk, 2, C), dtype=torch.float32, device=x.device)
    lib.call("sf_layernorm_bwd", M, C, dy.data_ptr(), lddy, x.data_ptr(), ldx, gamma.data_ptr(), mean.data_ptr(),
             rstd.data_ptr(), _ptr(resid), ldr, dx.data_ptr(), lddx, part.data_ptr(), _stream(x),
             work=dict(bytes=2.0 * M * C * (3 + int(resid is not None))))
    colsum_finalize(part, C, C, dgamma, dbeta, 1.0, accumulate)
    return dx


def bias_grad(dy, dbias, accumulate=False, fold=None):
    """dbias[c % fold] (+)= sum_m dy[m][c]."""
    M, C, ld = rows_pitch(dy)
    lib = get_lib()
    nblk = lib.call("sf_colsum_blocks", M, C)
    part = torch.empty((nblk, 2, C), dtype=torch.float32, device=dy.device)
    lib.call("sf_colsum", M, C, dy.data_ptr(), ld, part.data_ptr(), _stream(dy), work=dict(bytes=2.0 * M * C))
    colsum_finalize(part, C, fold or C, dbias, None, 1.0, accumulate)
    return dbias


def gelu_fwd(h):
    assert h.is_contiguous() and h.dtype == _f16
    a = torch.empty_like(h)
    _lib_call("sf_gelu_fwd", h.numel(), h.data_ptr(), a.data_ptr(), _stream(h), work=dict(bytes=4.0 * h.numel()))
    return a


def gelu_bwd(h, da):
    assert h.is_contiguous() and da.is_contiguous() and da.shape == h.shape
    dh = torch.empty_like(h)
    _lib_call("sf_gelu_bwd", h.numel(), h.data_ptr(), da.data_ptr(), dh.data_ptr(), _stream(h),
              work=dict(bytes=6.0 * h.numel()))
    return dh


# ------------------------------------------------------------------------------------------------
# depthwise convolution on token / channels-last rows
class DwGeom:
    """Depthwise Conv3d geometry over rows (n, [cls], t, h, w) with C channels (C % Cw == 0)."""

    def __init__(self, N, C, Cw, thw, kernel, stride, padding, cls, Cw_real=None):
        self.N, self.C, self.Cw, self.cls = N, C, Cw, int(bool(cls))
        self.Cw_real = Cw if Cw_real is None else Cw_real    # rows of the fp32 weight (X3D widths 54, 108 pad to 56, 112)
        self.thw, self.k, self.s, self.p = tuple(thw), tuple(kernel), tuple(stride), tuple(padding)
        self.out_thw = tuple((i + 2 * p - k) // s + 1 for i, k, s, p in zip(self.thw, self.k, self.s, self.p))
        self.taps = self.k[0] * self.k[1] * self.k[2]
        self.rows_in = N * (self.thw[0] * self.thw[1] * self.thw[2] + self.cls)
        self.rows_out = N * (self.out_thw[0] * self.out_thw[1] * self.out_thw[2] + self.cls)
        self.ws_bytes = None

    def desc(self, ldx, ldy):
        return DwDesc(self.N, self.C, self.Cw, self.cls, *self.thw, *self.out_thw, *self.k, *self.s, *self.p, ldx, ldy,
                      self.Cw_real)


def dwconv_fwd(x, w, geom, out=None, stats=False):
    """x: token tensor with geom.rows_in rows of geom.C channels; w: fp32 nn.Conv3d weight [Cw,1,kT,kH,kW]."""
    M, C, ldx = rows_pitch(x)
    assert M == geom.rows_in and C == geom.C and w.dtype == torch.float32 and w.is_contiguous()
    assert w.numel() == geom.Cw_real * geom.taps
    y = torch.empty((geom.rows_out, C), dtype=_f16, device=x.device) if out is None else out
    Mo, Co, ldy = rows_pitch(y)
    assert Mo == geom.rows_out and Co == C
    lib = get_lib()
    d = geom.desc(ldx, ldy)
    part = None
    if stats:
        part = torch.empty((lib.call("sf_dwconv_fwd_blocks", byref(d)), 2, C), dtype=torch.float32, device=x.device)
        # rows of the table per sample when it is sample-major (0: not), for callers that want per-sample sums (x3d SE squeeze)
        geom.stat_sample_rows = lib.call("sf_dwconv_fwd_sample_rows", byref(d))
    lib.call("sf_dwconv_fwd", byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), _ptr(part), _stream(x),
             work=dict(bytes=2.0 * C * (geom.rows_in + geom.rows_out), flops=2.0 * geom.rows_out * C * geom.taps))
    return (y, part) if stats else y


def dwconv_dgrad(dy, w, geom, out=None, sums=False):
    """``sums``: returns (dx, part) -- part [rows, 2, C] fp32 whose slot 0 holds per-workgroup column sums of dx (cls row included;
    take them with colsum_finalize), or None when the kernel this geometry runs on cannot leave them."""
    M, C, lddy = rows_pitch(dy)
    assert M == geom.rows_out and C == geom.C
    dx = torch.empty((geom.rows_in, C), dtype=_f16, device=dy.device) if out is None else out
    _, _, lddx = rows_pitch(dx)
    d = geom.desc(lddx, lddy)
    work = dict(bytes=2.0 * C * (geom.rows_in + geom.rows_out), flops=2.0 * geom.rows_out * C * geom.taps)
    if sums:
        lib = get_lib()
        rows = lib.call("sf_dwconv_dgrad_sum_rows", byref(d))
        if rows > 0:
            part = torch.empty((rows, 2, C), dtype=torch.float32, device=dy.device)
            lib.call("sf_dwconv_dgrad_sums", byref(d), dy.data_ptr(), w.data_ptr(), dx.data_ptr(), part.data_ptr(), _stream(dy),
                     work=work)
            return dx, part
    _lib_call("sf_dwconv_dgrad", byref(d), dy.data_ptr(), w.data_ptr(), dx.data_ptr(), _stream(dy), work=work)
    return (dx, None) if sums else dx


def dwconv_wgrad(x, dy, geom, dw, zero_first=True, out_scale=1.0):
    _, _, ldx = rows_pitch(x)
    _, _, lddy = rows_pitch(dy)
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == geom.Cw_real * geom.taps
    lib = get_lib()
    d = geom.desc(ldx, lddy)
    if geom.ws_bytes is None:
        geom.ws_bytes = lib.call("sf_dwconv_wgrad_workspace", byref(d))
    ws = _workspace(x.device, geom.ws_bytes)
    lib.call("sf_dwconv_wgrad", byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), float(out_scale), int(zero_first),
             ws.data_ptr(), ws.numel(), _stream(x),
             work=dict(bytes=2.0 * geom.C * (3 * geom.rows_out + geom.rows_in), flops=2.0 * geom.rows_out * geom.C * geom.taps))
    return dw


# PAIR forms (round 6): two depthwise convolutions of one geometry on two tensors in ONE launch per direction (MViT pool_k / pool_v of
# a block).  Same arithmetic as two single calls; dwconv_pair_ok says whether the native library takes the geometry.
def dwconv_pair_ok(geom, ldx, ldy):
    key = ("pair_ok", ldx, ldy)
    cache = geom.__dict__.setdefault("_pair_cache", {})
    if key not in cache:
        cache[key] = get_lib().call("sf_dwconv_pair_ok", byref(geom.desc(ldx, ldy))) == 1
    return cache[key]


def _pair_pitches(a, b):
    Ma, Ca, lda = rows_pitch(a)
    Mb, Cb, ldb = rows_pitch(b)
    assert (Ma, Ca, lda) == (Mb, Cb, ldb), "the two tensors of a pair share rows, channels and pitch"
    return Ma, Ca, lda


def dwconv_fwd_pair(x, x2, w, w2, geom):
    M, C, ldx = _pair_pitches(x, x2)
    assert M == geom.rows_in and C == geom.C and w.dtype == w2.dtype == torch.float32 and w.is_contiguous() and w2.is_contiguous()
    y = torch.empty((geom.rows_out, C), dtype=_f16, device=x.device)
    y2 = torch.empty((geom.rows_out, C), dtype=_f16, device=x.device)
    get_lib().call("sf_dwconv_fwd_pair", byref(geom.desc(ldx, C)), x.data_ptr(), x2.data_ptr(), w.data_ptr(), w2.data_ptr(),
                   y.data_ptr(), y2.data_ptr(), _stream(x),
                   work=dict(bytes=4.0 * C * (geom.rows_in + geom.rows_out), flops=4.0 * geom.rows_out * C * geom.taps))
    return y, y2


def dwconv_dgrad_pair(dy, dy2, w, w2, geom, out, out2):
    M, C, lddy = _pair_pitches(dy, dy2)
    Mi, Ci, lddx = _pair_pitches(out, out2)
    assert M == geom.rows_out and C == geom.C and Mi == geom.rows_in and Ci == C
    get_lib().call("sf_dwconv_dgrad_pair", byref(geom.desc(lddx, lddy)), dy.data_ptr(), dy2.data_ptr(), w.data_ptr(), w2.data_ptr(),
                   out.data_ptr(), out2.data_ptr(), _stream(dy),
                   work=dict(bytes=4.0 * C * (geom.rows_in + geom.rows_out), flops=4.0 * geom.rows_out * C * geom.taps))
    return out, out2


def dwconv_wgrad_pair(x, x2, dy, dy2, geom, dw, dw2, zero_first=True, zero_first2=True, out_scale=1.0):
    _, C, ldx = _pair_pitches(x, x2)
    _, _, lddy = _pair_pitches(dy, dy2)
    for g in (dw, dw2):
        assert g.dtype == torch.float32 and g.is_contiguous() and g.numel() == geom.Cw_real * geom.taps
    lib = get_lib()
    d = geom.desc(ldx, lddy)
    if geom.ws_bytes is None:
        geom.ws_bytes = lib.call("sf_dwconv_wgrad_workspace", byref(d))
    ws = _workspace(x.device, 2 * geom.ws_bytes)
    lib.call("sf_dwconv_wgrad_pair", byref(d), x.data_ptr(), x2.data_ptr(), dy.data_ptr(), dy2.data_ptr(), dw.data_ptr(),
             dw2.data_ptr(), float(out_scale), int(zero_first), int(zero_first2), ws.data_ptr(), ws.numel(), _stream(x),
             work=dict(bytes=4.0 * geom.C * (3 * geom.rows_out + geom.rows_in), flops=4.0 * geom.rows_out * geom.C * geom.taps))


# ------------------------------------------------------------------------------------------------
# token max-pool (MultiScaleBlock.pool_skip)
def token_pool_fwd(x, B, thw, kernel, stride, padding, cls=True):
    """MaxPool3d([1,kH,kW], [1,sH,sW], [0,pH,pW]) over the non-cls tokens of x [B, cls+T*H*W, C]."""
    T, H, W = thw
    assert kernel[0] == 1 and stride[0] == 1 and padding[0] == 0, "temporal pooling of the skip path is not used"
    M, C, ldx = rows_pitch(x)
    Ho, Wo = (H + 2 * padding[1] - kernel[1]) // stride[1] + 1, (W + 2 * padding[2] - kernel[2]) // stride[2] + 1
    No = int(cls) + T * Ho * Wo
    out = torch.empty((B, No, C), dtype=_f16, device=x.device)
    arg = torch.empty((B, No, C), dtype=torch.uint8, device=x.device)
    _lib_call("sf_pool_fwd", B, T, H, W, C, kernel[1], kernel[2], stride[1], stride[2], padding[1], padding[2],
              x.data_ptr(), ldx, None, None, 0, out.data_ptr(), C, arg.data_ptr(), int(cls), _stream(x),
              work=dict(bytes=2.0 * C * (M + B * No)))
    return out, arg, (T, Ho, Wo)


def token_pool_bwd(dout, pooled, arg, B, thw, kernel, stride, padding, C, cls=True):
    T, H, W = thw
    Ni = int(cls) + T * H * W
    g = torch.empty((B, Ni, C), dtype=_f16, device=dout.device)
    _lib_call("sf_pool_bwd", B, T, H, W, C, kernel[1], kernel[2], stride[1], stride[2], padding[1], padding[2],
              pooled.data_ptr(), rows_pitch(pooled)[2], arg.data_ptr(), 0, dout.data_ptr(), rows_pitch(dout)[2],
              g.data_ptr(), C, int(cls), _stream(dout), work=dict(bytes=2.0 * C * (B * Ni + 2.5 * dout.numel() / C)))
    return g


# ------------------------------------------------------------------------------------------------
# pooled attention pieces
def attn_desc(B, heads, D, cls, q_thw, k_thw, rows_h=0, rows_w=0, rows_t=0):
    Nq = int(cls) + q_thw[0] * q_thw[1] * q_thw[2]
    Nk = int(cls) + k_thw[0] * k_thw[1] * k_thw[2]
    return AttnDesc(B, heads, D, int(cls), Nq, *q_thw, Nk, *k_thw, rows_h, rows_w, rows_t)


def relpos_tables16(tables):
    """Concatenated table [rel_pos_h; rel_pos_w; rel_pos_t] as fp16 GEMM operands: (Tab [TRp, D], Tab^T [D, TRp]),
    TRp = row count rounded up to 8 (zero rows)."""
    D = tables[0].shape[1]
    rows = [t.shape[0] for t in tables]
    TRp = (sum(rows) + 7) // 8 * 8
    tabs = [t.detach() for t in tables]
    if not all(t.dtype == torch.float32 and t.is_contiguous() for t in tabs):       # resampled tables (plan.interp)
        tabs = [t.float().contiguous() for t in tabs]
    t16 = torch.empty((TRp, D), dtype=_f16, device=tabs[0].device)
    t16t = torch.empty((D, TRp), dtype=_f16, device=tabs[0].device)
    _lib_call("sf_relpos_pack", tabs[0].data_ptr(), tabs[1].data_ptr(), tabs[2].data_ptr(), rows[0], rows[1], rows[2], D, TRp,
              t16.data_ptr(), t16t.data_ptr(), _stream(t16))
    return t16, t16t


def relpos_fwd(d, q, tables, idx, t16=None):
    """rq [B*Nq*heads, kH+kW+kT] fp32 from the UNSCALED q [B, Nq, heads*D]: G = q Tab^T on the MFMA GEMM, then the
    per-row gather of the kH+kW+kT columns the row's position selects."""
    if t16 is None:
        t16 = relpos_tables16(tables)[0]
    R = d.kH + d.kW + d.kT
    q2d = q.reshape(-1, d.D)
    G = gemm(q2d, t16)
    rq = torch.empty((q2d.shape[0], R), dtype=torch.float32, device=q.device)
    _lib_call("sf_relpos_gather", byref(d), G.data_ptr(), G.shape[1], idx[0].data_ptr(), idx[1].data_ptr(),
              idx[2].data_ptr(), rq.data_ptr(), _stream(q), work=dict(bytes=2.0 * G.numel()))
    return rq


def relpos_bwd(d, q, tables, idx, drq, dq, dtables, accumulate, t16t=None):
    """dq += E Tab; dtables[i] (+)= rows of E^T q, with E the scatter of drq onto the table-row columns."""
    if t16t is None:
        t16t = relpos_tables16(tables)[1]
    TRp = t16t.shape[1]
    q2d, dq2d = q.reshape(-1, d.D), dq.view(-1, d.D)
    E = torch.empty((q2d.shape[0], TRp), dtype=_f16, device=q.device)
    _lib_call("sf_relpos_scatter", byref(d), drq.data_ptr(), idx[0].data_ptr(), idx[1].data_ptr(), idx[2].data_ptr(),
              E.data_ptr(), TRp, _stream(q), work=dict(bytes=2.0 * E.numel()))
    gemm(E, t16t, resid=dq2d, out=dq2d)                     # in place: each element is read then written by one thread
    dtab = torch.empty((TRp, d.D), dtype=torch.float32, device=q.device)
    linear_wgrad(q2d, E, dtab, zero_first=True)
    rows = [t.shape[0] for t in tables]
    assert all(g.dtype == torch.float32 and g.is_contiguous() and tuple(g.shape) == (n, d.D) for g, n in zip(dtables, rows))
    _lib_call("sf_relpos_unpack", dtab.data_ptr(), rows[0], rows[1], rows[2], d.D, dtables[0].data_ptr(), dtables[1].data_ptr(),
              dtables[2].data_ptr(), int(accumulate[0]), int(accumulate[1]), int(accumulate[2]), _stream(dtab))


def softmax_fwd(d, s, scale, rq=None):
    lds = s.shape[-1]
    _lib_call("sf_softmax_fwd", byref(d), s.data_ptr(), lds, float(scale), _ptr(rq), _stream(s),
              work=dict(bytes=4.0 * s.numel()))
    return s


def softmax_bwd(d, dp, prob, scale, want_drq):
    lds = dp.shape[-1]
    drq = None
    if want_drq:
        drq = torch.empty((d.B * d.Nq * d.heads, d.kH + d.kW + d.kT), dtype=torch.float32, device=dp.device)
    _lib_call("sf_softmax_bwd", byref(d), dp.data_ptr(), prob.data_ptr(), lds, float(scale), _ptr(drq), _stream(dp),
              work=dict(bytes=6.0 * dp.numel()))
    return dp, drq


def attn_onehot(d, device):
    """Constant [roundup(Nk, 32), 64] fp16 matrix with ones at columns kh, kH + kw, kH + kW + kt of every non-cls key
    row (index bookkeeping of the decomposed rel-pos bias, built once per attention shape)."""
    NkP = (d.Nk + 31) // 32 * 32
    oh = torch.zeros((NkP, 64), dtype=_f16)
    pos = torch.arange(d.kT * d.kH * d.kW)
    kw, kh, kt = pos % d.kW, (pos // d.kW) % d.kH, pos // (d.kW * d.kH)
    rows = pos + d.cls
    oh[rows, kh] = 1
    oh[rows, d.kH + kw] = 1
    oh[rows, d.kH + d.kW + kt] = 1
    return oh.to(device)


def attn_fwd(d, q, k, v, scale, rq, residual, onehot=None):
    """Fused softmax(scale q k^T + bias) v (+ q): q [B, Nq, heads*D], k/v [B, Nk, heads*D] -> (o, lse)."""
    B, Nq, C = q.shape
    o = torch.empty((B, Nq, C), dtype=_f16, device=q.device)
    lse = torch.empty((B * d.heads * Nq,), dtype=torch.float32, device=q.device)
    assert (rq is None) == (onehot is None)
    flops = 4.0 * B * d.heads * Nq * d.Nk * d.D
    _lib_call("sf_attn_fwd", byref(d), q.data_ptr(), rows_pitch(q)[2], k.data_ptr(), v.data_ptr(), rows_pitch(k)[2],
              float(scale), _ptr(rq), _ptr(onehot), int(bool(residual)), o.data_ptr(), C, lse.data_ptr(), _stream(q),
              work=dict(bytes=2.0 * (2 * q.numel() + 2 * k.numel()), flops=flops))
    return o, lse


def attn_bwd(d, q, k, v, scale, rq, residual, o, do, lse, onehot=None, dq_out=None, dkv_out=None):
    """Backward of attn_fwd: (dq, dk, dv, drq).  ``dq_out`` / ``dkv_out = (dk, dv)`` let the gradients land directly in
    channel slices of a wider buffer (the d(qkv) tensor when q or k/v are not pooled); dk and dv share one pitch."""
    B, Nq, C = q.shape
    dq = torch.empty((B, Nq, C), dtype=_f16, device=q.device) if dq_out is None else dq_out
    if dkv_out is None:
        dk = torch.empty(k.shape, dtype=_f16, device=q.device)
        dv = torch.empty(k.shape, dtype=_f16, device=q.device)
    else:
        dk, dv = dkv_out
        assert rows_pitch(dk)[2] == rows_pitch(dv)[2]
    assert tuple(dq.shape) == tuple(q.shape) and tuple(dk.shape) == tuple(k.shape) == tuple(dv.shape)
    delta = torch.empty_like(lse)
    drq = torch.empty(rq.shape, dtype=torch.float32, device=q.device) if rq is not None else None
    nbytes = _lib_call("sf_attn_bwd_workspace", byref(d))
    ws = _workspace(q.device, nbytes) if nbytes > 0 else None
    # ALGORITHMIC flops of the attention backward: dV = P^T dO, dP = dO V^T, dQ = dS K, dK = dS^T Q -> 8 * Nq * Nk * D per
    # (batch, head).  The two kernels (query side, key side) each recompute S and dP on top of that (14 executed); the
    # roofline fraction bench.py reports is priced on the algorithmic count.
    flops = 8.0 * B * d.heads * Nq * d.Nk * d.D
    _lib_call("sf_attn_bwd", byref(d), q.data_ptr(), rows_pitch(q)[2], k.data_ptr(), v.data_ptr(), rows_pitch(k)[2],
              float(scale), _ptr(rq), _ptr(onehot), int(bool(residual)), o.data_ptr(), do.data_ptr(), rows_pitch(o)[2],
              lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), rows_pitch(dq)[2], dk.data_ptr(), dv.data_ptr(), rows_pitch(dk)[2],
              _ptr(drq), _ptr(ws), nbytes, _stream(q), work=dict(bytes=2.0 * (4 * q.numel() + 4 * k.numel()), flops=flops))
    return dq, dk, dv, drq


def row_scale_add(x, scale, rows_per_sample, resid=None, side=None):
    """y = resid + scale[row // rows_per_sample] * x on token rows (stochastic depth, common.py:46-59).
    ``side = (period, src32 | None, dst32)`` as in gemm()."""
    M, C, ldx = rows_pitch(x)
    y = torch.empty(x.shape, dtype=_f16, device=x.device)
    assert scale.dtype == torch.float32 and scale.numel() * rows_per_sample == M
    r_ptr, ldr = (resid.data_ptr(), rows_pitch(resid)[2]) if resid is not None else (None, 0)
    if side is not None:
        d = _rows32(M, C, side[0], side[1], side[2])
        _lib_call("sf_row_scale_add_rows32", x.data_ptr(), ldx, scale.data_ptr(), rows_per_sample, r_ptr, ldr, y.data_ptr(),
                  rows_pitch(y)[2], M, C, byref(d), _stream(x), work=dict(bytes=2.0 * x.numel() * (3 if resid is not None else 2)))
        return y
    _lib_call("sf_row_scale_add", x.data_ptr(), ldx, scale.data_ptr(), rows_per_sample, r_ptr, ldr, y.data_ptr(),
              rows_pitch(y)[2], M, C, _stream(x), work=dict(bytes=2.0 * x.numel() * (3 if resid is not None else 2)))
    return y


def transpose_heads(x, B, Nk, heads, D, ldk):
    """[B, Nk, heads*D] -> [B, heads, D, ldk] (zero padded keys)."""
    xt = torch.empty((B, heads, D, ldk), dtype=_f16, device=x.device)
    _lib_call("sf_transpose_heads", x.data_ptr(), rows_pitch(x)[2], xt.data_ptr(), ldk, B, Nk, heads, D, _stream(x),
              work=dict(bytes=4.0 * x.numel()))
    return xt
