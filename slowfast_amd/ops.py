"""Tensor-level wrappers over the C ABI (include/sfamd.h).

PyTorch is used here only for device memory and streams.  Activations are fp16 tensors of LOGICAL
shape (N, C, T, H, W) -- the reference's NCTHW convention (slowfast/models/video_model_builder.py:423)
-- whose MEMORY order is N,T,H,W,C with a row pitch ``ld`` (``torch.channels_last_3d`` strides, or a
channel slice of such a tensor).  Nothing in this file computes on the CPU or through ATen kernels.
"""
from ctypes import byref, c_int32

import torch

from . import lib as _sflib

from .lib import ConvDesc, SfError, get_lib

_f16 = _sflib.act_dtype()        # fp16, or bf16 under SF_ACT_DTYPE=bf16 (lib.ACT_MODE)


def _triple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


def _stream(t):
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    if get_lib().backend != "hostsim":
        raise SfError("slowfast_amd kernels need CUDA/HIP tensors (got a CPU tensor and the gfx950 library)")
    return None


def _ptr(t):
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------
# channels-last helpers
def cl_empty(shape, device, ld=None, zero=False):
    """fp16 tensor of logical shape (N,C,T,H,W) stored N,T,H,W,C with row pitch ``ld``."""
    N, C, T, H, W = shape
    ld = C if ld is None else ld
    base = (torch.zeros if zero else torch.empty)((N, T, H, W, ld), dtype=_f16, device=device)
    return base[..., :C].permute(0, 4, 1, 2, 3)


def cl_ld(x):
    """Row pitch of a channels-last activation; raises if ``x`` is not in that layout."""
    if x.dim() != 5 or x.dtype != _f16:
        raise SfError(f"expected a 5-D fp16 channels-last activation, got {tuple(x.shape)} {x.dtype}")
    N, C, T, H, W = x.shape
    if C > 1 and x.stride(1) != 1:
        raise SfError("activation is not channels-last (channel stride != 1); use ops.to_cl()")
    if W > 1:
        ld = x.stride(4)
    elif H > 1:
        ld = x.stride(3)
    elif T > 1:
        ld = x.stride(2)
    elif N > 1:
        ld = x.stride(0)
    else:
        ld = C
    exp = (T * H * W * ld, 1, H * W * ld, W * ld, ld)
    for dim, size in enumerate(x.shape):
        if size > 1 and x.stride(dim) != exp[dim]:
            raise SfError(f"activation strides {x.stride()} are not channels-last with pitch {ld}")
    if ld < C or ld % 8 or C % 8 or x.data_ptr() % 16:
        raise SfError(f"activation needs C % 8 == 0, pitch % 8 == 0, 16-byte base (C={C}, ld={ld})")
    return ld


def is_cl(x):
    try:
        cl_ld(x)
        return True
    except SfError:
        return False


def rows(x):
    N, C, T, H, W = x.shape
    return N * T * H * W


def ncthw_to_cl(x, Cp=None):
    """NCTHW fp32 clip -> channels-last fp16, channels zero-padded to ``Cp`` (sf_ncthw_to_cl)."""
    assert x.dim() == 5 and x.dtype == torch.float32
    x = x.contiguous()
    N, C, T, H, W = x.shape
    Cp = Cp or (C + 7) // 8 * 8
    out = cl_empty((N, Cp, T, H, W), x.device)
    get_lib().call("sf_ncthw_to_cl", x.data_ptr(), N, C, T * H * W, Cp, out.data_ptr(), _stream(x),
                   work=dict(bytes=4.0 * x.numel() + 2.0 * out.numel()))
    return out


def ncthw_to_cl_wpairs(x):
    """NCTHW fp32 clip with C <= 4 -> fp16 N,T,H,W,4 viewed as a channels-last tensor of logical shape
    (N, 8, T, H, W/2): channel index = (w & 1) * 4 + c.  Operand layout of the W-pair-folded stem convolution
    (engine.StemConvUnit)."""
    assert x.dim() == 5 and x.dtype == torch.float32 and x.shape[1] <= 4 and x.shape[4] % 2 == 0
    x = x.contiguous()
    N, C, T, H, W = x.shape
    base = torch.empty((N, T, H, W // 2, 8), dtype=_f16, device=x.device)
    get_lib().call("sf_ncthw_to_cl", x.data_ptr(), N, C, T * H * W, 4, base.data_ptr(), _stream(x),
                   work=dict(bytes=4.0 * x.numel() + 2.0 * base.numel()))
    return base.permute(0, 4, 1, 2, 3)


def cl_to_ncthw(x):
    """channels-last fp16 -> contiguous NCTHW fp32 (sf_cl_to_ncthw)."""
    ld = cl_ld(x)
    N, C, T, H, W = x.shape
    out = torch.empty((N, C, T, H, W), dtype=torch.float32, device=x.device)
    get_lib().call("sf_cl_to_ncthw", x.data_ptr(), ld, N, C, T * H * W, out.data_ptr(), _stream(x))
    return out


def to_cl(x):
    """Accept what a caller of the reference modules would pass: NCTHW fp32/fp16 in any layout."""
    if x.dtype == _f16 and is_cl(x):
        return x
    return ncthw_to_cl(x.float())


# ------------------------------------------------------------------------------------------------
class ConvGeom:
    """Geometry of one nn.Conv3d call (groups == 1) on a given input shape."""

    def __init__(self, in_shape, Co, kernel, stride=1, padding=0, dilation=1, Cw=None, out_dims=None):
        self.N, self.Ci, self.Ti, self.Hi, self.Wi = in_shape
        self.Cow = Co                       # channels of the fp32 weight on the output side
        self.Co = (Co + 7) // 8 * 8         # channels of the output activation buffer (zero padded)
        self.k, self.s, self.p, self.d = _triple(kernel), _triple(stride), _triple(padding), _triple(dilation)
        self.Cw = self.Ci if Cw is None else Cw
        self.To, self.Ho, self.Wo = [
            (i + 2 * p - d * (k - 1) - 1) // s + 1
            for i, k, s, p, d in zip((self.Ti, self.Hi, self.Wi), self.k, self.s, self.p, self.d)]
        if out_dims is not None:      # drop trailing output positions (asymmetric end padding)
            assert all(1 <= o <= f for o, f in zip(out_dims, (self.To, self.Ho, self.Wo)))
            self.To, self.Ho, self.Wo = out_dims
        self.taps = self.k[0] * self.k[1] * self.k[2]
        ldf, ldd = c_int32(), c_int32()
        get_lib().call("sf_conv_weight_ld", byref(self.desc(self.Ci, self.Co)), byref(ldf), byref(ldd))
        self.ldf, self.ldd = ldf.value, ldd.value
        self.ws_bytes = None   # wgrad split-partial workspace (queried lazily)

    @property
    def in_shape(self):
        return (self.N, self.Ci, self.Ti, self.Hi, self.Wi)

    @property
    def out_shape(self):
        return (self.N, self.Co, self.To, self.Ho, self.Wo)

    @property
    def out_rows(self):
        return self.N * self.To * self.Ho * self.Wo

    def work(self, reads_x=0, reads_y=0, writes_x=0, writes_y=0):
        """Algorithmic work of one launch: 2*MACs flops; fp16 activation bytes that must cross HBM once each."""
        xe = self.N * self.Cw * self.Ti * self.Hi * self.Wi
        ye = self.out_rows * self.Co
        return dict(flops=2.0 * self.out_rows * self.Co * self.Cw * self.taps,
                    bytes=2.0 * ((reads_x + writes_x) * xe + (reads_y + writes_y) * ye))

    def desc(self, ldx, ldy):
        return ConvDesc(self.N, self.Ci, self.Ti, self.Hi, self.Wi, self.Co, self.To, self.Ho, self.Wo,
                        *self.k, *self.s, *self.p, *self.d, self.Cw, ldx, ldy, self.Cow)


def prep_weights(w, geom, need_dgrad=True):
    """fp32 Conv3d weight -> fp16 GEMM operands (forward [Co][ldf], dgrad [Ci][ldd])."""
    assert w.dtype == torch.float32 and tuple(w.shape) == (geom.Cow, geom.Cw) + geom.k, (w.shape, geom.Cow, geom.Cw)
    w = w.contiguous()
    wf = torch.empty((geom.Co, geom.ldf), dtype=_f16, device=w.device)
    wd = torch.empty((geom.Ci, geom.ldd), dtype=_f16, device=w.device) if need_dgrad else None
    get_lib().call("sf_prep_weights", byref(geom.desc(geom.Ci, geom.Co)), w.data_ptr(), wf.data_ptr(), _ptr(wd),
                   _stream(w))
    return wf, wd


def _affine(in_affine):
    if in_affine is None:
        return None, None, 0
    scale, shift, relu = in_affine
    assert scale.dtype == torch.float32 and shift.dtype == torch.float32
    return scale, shift, int(bool(relu))


def conv_fwd(x, wf, geom, in_affine=None, bias=None, stats=True, out=None):
    """y = conv3d(act(x)); returns (y, stat_part or None).  act = producer BN(+ReLU) applied on the fly."""
    assert tuple(x.shape) == geom.in_shape, (x.shape, geom.in_shape)
    ldx = cl_ld(x)
    y = cl_empty(geom.out_shape, x.device) if out is None else out
    ldy = cl_ld(y)
    lib = get_lib()
    d = geom.desc(ldx, ldy)
    part = None
    if stats:
        mt = lib.call("sf_conv_fwd_mtiles", byref(d))
        part = torch.empty((mt, 2, geom.Co), dtype=torch.float32, device=x.device)
    sc, sh, relu = _affine(in_affine)
    lib.call("sf_conv_fwd", byref(d), x.data_ptr(), wf.data_ptr(), _ptr(sc), _ptr(sh), relu, _ptr(bias),
             y.data_ptr(), _ptr(part), _stream(x), work=geom.work(reads_x=1, reads_y=0, writes_y=1))
    return y, part


def conv_fwd_fused(x, wf, geom, bias=None, resid=None, relu=False, out=None):
    """y = relu?(conv3d(x) + bias [+ resid]): the eval-mode convolution with BatchNorm folded into ``wf`` / ``bias``
    (sf_conv_fwd_fused)."""
    assert tuple(x.shape) == geom.in_shape, (x.shape, geom.in_shape)
    y = cl_empty(geom.out_shape, x.device) if out is None else out
    assert tuple(y.shape) == geom.out_shape
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == geom.Co, "bias must cover the padded channel count"
    ldr = 0
    if resid is not None:
        assert tuple(resid.shape) == geom.out_shape
        ldr = cl_ld(resid)
    get_lib().call("sf_conv_fwd_fused", byref(geom.desc(cl_ld(x), cl_ld(y))), x.data_ptr(), wf.data_ptr(), _ptr(bias),
                   _ptr(resid), ldr, int(bool(relu)), y.data_ptr(), _stream(x),
                   work=geom.work(reads_x=1, reads_y=int(resid is not None), writes_y=1))
    return y


def conv_dgrad(dy, wd, geom, resid=None, out=None, resid_bits=None, bn=None):
    """dx = conv_transpose3d(dy, w) [+ resid].  ``resid_bits``: the bit mask bn_act(..., want_mask=True) wrote for the
    tensor ``resid`` is the gradient of; only residual elements whose bit is set are added.

    ``bn``: the kernel epilogue also takes the reduction pass of the BatchNorm backward(s) that consume dx (sums of g and
    g * y over the positions, g = dx under the ReLU mask) from the tile it stores.  Two forms:
      * ``(y, scale, shift)`` -- this convolution's input was the inner activation relu(bn(y)): mask recomputed;
      * ``{"bits": mask, "y0": yc}`` -- its input was the previous block's output relu(bn_c(yc) + shortcut): ``bits`` is that
        output's 1-bit image (the BatchNorm of a projection shortcut keeps its own reduction pass).
    Returns (dx, part): part [rows, 2, Ci] fp32 for bn_bwd(..., part=part) -- None when the geometry keeps the separate pass
    (strided data gradients)."""
    assert tuple(dy.shape) == geom.out_shape
    if resid is not None:
        assert tuple(resid.shape) == geom.in_shape
    if out is not None:
        assert tuple(out.shape) == geom.in_shape
    if bn is not None:
        if not isinstance(bn, dict):
            y0, sc, sh = bn                      # inner activation: mask recomputed from (y, scale, shift)
            bits = None
        else:
            bits, y0 = bn["bits"], bn["y0"]      # block input: the previous block's 1-bit output mask
            sc = sh = None
            assert bits.dtype == torch.uint8 and bits.numel() == rows(y0) * (geom.Ci // 8) and bits.is_contiguous()
        assert tuple(y0.shape) == geom.in_shape
        dx = cl_empty(geom.in_shape, dy.device) if out is None else out
        M = rows(y0)
        cap = (M + 127) // 128
        part = torch.empty((cap, 2, geom.Ci), dtype=torch.float32, device=dy.device)
        nrows = c_int32(0)
        if resid_bits is not None:
            assert resid is not None and resid_bits.dtype == torch.uint8 and resid_bits.numel() == rows(resid) * (geom.Ci // 8)
        get_lib().call("sf_conv_dgrad_bn", byref(geom.desc(cl_ld(dx), cl_ld(dy))), dy.data_ptr(), wd.data_ptr(), _ptr(resid),
                       cl_ld(resid) if resid is not None else 0, _ptr(resid_bits), dx.data_ptr(), _ptr(sc), _ptr(sh), _ptr(bits),
                       y0.data_ptr(), cl_ld(y0), part.data_ptr(), cap, byref(nrows), _stream(dy),
                       work=geom.work(reads_x=1 + int(resid is not None), reads_y=1, writes_x=1))
        return dx, (part[:nrows.value] if nrows.value > 0 else None)
    ldy = cl_ld(dy)
    dx = cl_empty(geom.in_shape, dy.device) if out is None else out
    ldx = cl_ld(dx)
    ldr = cl_ld(resid) if resid is not None else 0
    if resid_bits is not None:
        assert resid is not None and resid_bits.dtype == torch.uint8 and resid_bits.numel() == rows(resid) * (geom.Ci // 8)
    get_lib().call("sf_conv_dgrad", byref(geom.desc(ldx, ldy)), dy.data_ptr(), wd.data_ptr(), _ptr(resid), ldr,
                   _ptr(resid_bits), dx.data_ptr(), _stream(dy),
                   work=geom.work(reads_x=int(resid is not None), reads_y=1, writes_x=1))
    return dx


_workspaces = {}
_retired_workspaces = []


def _workspace(device, nbytes, key=None):
    """One grow-only scratch buffer per device: every user is enqueued on the same stream, so launches that
    share it are ordered (the callee allocates nothing, SURVEY.md 8b 'Ownership').

    A captured HIP graph (slowfast_amd.step.TrainStep) bakes the buffer's ADDRESS into its kernel nodes.  When a later,
    larger request replaces the buffer, the superseded one is therefore kept alive (never handed back to the caching
    allocator): a graph captured earlier keeps writing its split-K partials into memory that is still reserved for
    exactly that, instead of into whatever tensor the allocator would have placed there.  Growth happens a handful of
    times per process (the largest layer geometry wins), so the retained memory is bounded by ~2x the final size."""
    # launches on different streams (engine.run_pathways, engine.WGRAD_STREAM) must not share scratch memory: one buffer per
    # (device, stream) -- a captured graph bakes the stream's buffer into its branch
    sid = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
    k = (device, key, sid)
    ws = _workspaces.get(k)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _retired_workspaces.append(ws)
        ws = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[k] = ws
    return ws


def conv_wgrad(x, dy, geom, dw, in_affine=None, out_scale=1.0, zero_first=True, side=False):
    """dw (+)= out_scale * d(loss)/d(weight); dw is an fp32 tensor shaped like the Conv3d weight."""
    assert tuple(x.shape) == geom.in_shape and tuple(dy.shape) == geom.out_shape
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == geom.Cow * geom.Cw * geom.taps
    sc, sh, relu = _affine(in_affine)
    lib = get_lib()
    d = geom.desc(cl_ld(x), cl_ld(dy))
    if geom.ws_bytes is None:
        geom.ws_bytes = lib.call("sf_conv_wgrad_workspace", byref(d))
    ws = _workspace(x.device, geom.ws_bytes, "wgrad-side" if side else None)
    lib.call("sf_conv_wgrad", byref(d), x.data_ptr(), _ptr(sc), _ptr(sh), relu,
                   dy.data_ptr(), dw.data_ptr(), float(out_scale), int(zero_first), ws.data_ptr(), ws.numel(),
                   _ptr(_wgrad_rowtab(geom, d, x.device)) if sc is None else None, _stream(x),
                   work=geom.work(reads_x=1, reads_y=1))
    return dw


_rowtabs = {}       # (device, direction, spatial geometry) -> row table ({first source position, tap mask} per row)


def _capturing(device):
    """A table first needed while a HIP graph is being captured is allocated from the graph's private pool and only filled
    when the graph is replayed: such a table is used by the captured launch alone and NOT put into the process-wide cache
    (a later eager user of the same geometry, or a side stream, would read memory that is unordered with its fill).
    Geometries met during the eager warm-up iterations (the normal case) are cached as before."""
    return device.type == "cuda" and torch.cuda.is_current_stream_capturing()


def _wgrad_rowtab(geom, d, device):
    """The {first input position, tap mask} table of sf_conv_wgrad's large-K kernel is a function of the geometry alone:
    built once per (device, geometry) -- shared by every layer of that shape -- instead of once per call.  Built on first
    use, i.e. during the eager warm-up iterations of step.TrainStep; a build that happens while a graph is being captured
    is replayed with the graph and stays private to it (_capturing)."""
    nbytes = getattr(geom, "_rowtab_bytes", None)
    if nbytes is None:
        nbytes = geom._rowtab_bytes = get_lib().call("sf_conv_wgrad_rowtab_bytes", byref(d))
    if nbytes <= 0:
        return None                     # this layer takes another kernel
    key = (device, 0, geom.N, geom.Ti, geom.Hi, geom.Wi, geom.To, geom.Ho, geom.Wo, geom.k, geom.s, geom.p, geom.d)
    if key not in _rowtabs:
        tab = torch.empty(nbytes, dtype=torch.uint8, device=device)
        get_lib().call("sf_conv_wgrad_rowtab", byref(d), tab.data_ptr(), _stream(tab))
        if _capturing(device):
            return tab                  # lives in the graph's pool, rebuilt by every replay: never shared through the cache
        if device.type == "cuda":       # the cached table is read from OTHER streams too (engine.run_pathways): complete it first
            torch.cuda.current_stream(device).synchronize()     # (once per geometry, during the eager warm-up iteration)
        _rowtabs[key] = tab
    return _rowtabs[key]


# ------------------------------------------------------------------------------------------------
def bn_finalize(part, count, gamma, beta, running_mean, running_var, momentum, eps, training=True, C=None):
    """Per-tile sums -> (scale, shift, mean, rstd); updates running statistics in training mode.  ``C`` = channel
    count of the (zero padded) activation buffer when it exceeds the parameter length."""
    Creal = gamma.numel()
    C = C or (part.shape[2] if part is not None else Creal)
    dev = gamma.device
    scale, shift, mean, rstd = (torch.empty(C, dtype=torch.float32, device=dev) for _ in range(4))
    nblk = part.shape[0] if training else 0
    get_lib().call("sf_bn_finalize", _ptr(part) if training else None, nblk, C, Creal, float(count), gamma.data_ptr(),
                   beta.data_ptr(), _ptr(running_mean), _ptr(running_var), float(momentum), float(eps),
                   scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _stream(gamma))
    return scale, shift, mean, rstd


def bn_act(y, scale=None, shift=None, relu=False, resid=None, rscale=None, rshift=None, out=None, want_mask=False):
    """out = relu?(y*scale+shift [+ resid*rscale+rshift | + resid]) materialised in fp16.  ``want_mask``: also returns the
    1-bit mask ``out > 0`` ([rows, C/8] uint8) that bn_bwd / conv_dgrad read instead of ``out`` (1/16 of the bytes)."""
    ldy = cl_ld(y)
    N, C, T, H, W = y.shape
    out = cl_empty(y.shape, y.device) if out is None else out
    assert tuple(out.shape) == tuple(y.shape)
    mask = torch.empty((rows(y), C // 8), dtype=torch.uint8, device=y.device) if want_mask else None
    get_lib().call("sf_bn_act", rows(y), C, y.data_ptr(), ldy, _ptr(scale), _ptr(shift), _ptr(resid),
                   cl_ld(resid) if resid is not None else 0, _ptr(rscale), _ptr(rshift), int(bool(relu)),
                   out.data_ptr(), cl_ld(out), _ptr(mask), _stream(y),
                   work=dict(bytes=2.0 * y.numel() * (2 + int(resid is not None) + (1 / 16 if want_mask else 0))))
    return (out, mask) if want_mask else out


def bn_bwd(dz, y, gamma, mean, rstd, dgamma, dbeta, zmask=None, relu_affine=None, inv_loss_scale=1.0,
           accumulate=False, want_g=False, out=None, sync=None, part=None, sample_add=None):
    """BatchNorm3d (training) backward through an optional ReLU.

    dz: gradient w.r.t. act(bn(y)); the ReLU mask is ``zmask > 0`` (block output) or recomputed from
    ``relu_affine = (scale, shift)``; writes fp32 dgamma/dbeta and returns dy (and the masked g).
    ``part``: the [rows, 2, C] partial sums the producer of dz already took in its epilogue (conv_dgrad(..., bn=...)):
    the reduction pass over dz and y is skipped.  ``sample_add`` [N, C] fp32: dz lacks this per-sample constant (``part`` must hold
    the sums of the complete gradient); the apply pass adds it back (sf_bn_bwd_apply_sample)."""
    lib = get_lib()
    N, C, T, H, W = y.shape
    M = rows(y)
    s = _stream(y)
    fused_part = part
    if fused_part is None:
        nblk = lib.call("sf_bn_bwd_blocks", M, C)
        part = torch.empty((nblk, 2, C), dtype=torch.float32, device=y.device)
    else:
        assert part.dtype == torch.float32 and part.is_contiguous() and tuple(part.shape[1:]) == (2, C)
        nblk = part.shape[0]
    sc, sh = (relu_affine if relu_affine is not None else (None, None))
    relu_self = int(relu_affine is not None)
    bits = zmask is not None and zmask.dtype == torch.uint8     # the 1-bit mask of bn_act(..., want_mask=True)
    if bits:
        assert zmask.numel() == M * (C // 8) and zmask.is_contiguous()
    mcost = 0.0 if zmask is None else (1 / 16 if bits else 1.0)
    lddz, ldy, ldm = cl_ld(dz), cl_ld(y), (0 if bits or zmask is None else cl_ld(zmask))
    if fused_part is None:
        lib.call("sf_bn_bwd_reduce", M, C, dz.data_ptr(), lddz, _ptr(zmask), ldm, y.data_ptr(), ldy, _ptr(sc), _ptr(sh),
                 relu_self, part.data_ptr(), s, work=dict(bytes=2.0 * y.numel() * (2 + mcost)))
    coef = torch.empty((3, C), dtype=torch.float32, device=y.device)
    if sync is None:
        lib.call("sf_bn_bwd_finalize", part.data_ptr(), nblk, C, gamma.numel(), float(M), gamma.data_ptr(), mean.data_ptr(),
                 rstd.data_ptr(), float(inv_loss_scale), dgamma.data_ptr(), dbeta.data_ptr(), int(accumulate),
                 coef.data_ptr(), s)
    else:
        # synchronised BatchNorm (NaiveSyncBatchNorm3d): the affine gradients are sums over the LOCAL samples, the
        # input gradient uses the sums over the whole sync group (autograd of the differentiable all-reduce of the
        # batch moments): finalize once on the local sums for dgamma / dbeta, once on the all-reduced sums for coef
        import torch.distributed as dist
        group, gsize = sync
        tot = part.sum(0, keepdim=True)                   # [1, 2, C]
        loc = tot.clone()
        lib.call("sf_bn_bwd_finalize", loc.data_ptr(), 1, C, gamma.numel(), float(M), gamma.data_ptr(), mean.data_ptr(),
                 rstd.data_ptr(), float(inv_loss_scale), dgamma.data_ptr(), dbeta.data_ptr(), int(accumulate),
                 coef.data_ptr(), s)
        dist.all_reduce(tot, group=group)
        scratch = torch.empty((2, gamma.numel()), dtype=torch.float32, device=y.device)
        lib.call("sf_bn_bwd_finalize", tot.data_ptr(), 1, C, gamma.numel(), float(M) * gsize, gamma.data_ptr(),
                 mean.data_ptr(), rstd.data_ptr(), float(inv_loss_scale), scratch[0].data_ptr(), scratch[1].data_ptr(), 0,
                 coef.data_ptr(), s)
    dy = cl_empty(y.shape, y.device) if out is None else out
    if sample_add is not None:
        assert fused_part is not None and zmask is None and relu_affine is None and not want_g
        assert sample_add.dtype == torch.float32 and sample_add.is_contiguous() and tuple(sample_add.shape) == (N, C)
        lib.call("sf_bn_bwd_apply_sample", M, C, dz.data_ptr(), lddz, y.data_ptr(), ldy, coef.data_ptr(), sample_add.data_ptr(),
                 M // N, dy.data_ptr(), cl_ld(dy), s, work=dict(bytes=2.0 * y.numel() * 3))
        return dy
    g = cl_empty(y.shape, y.device) if want_g else None
    lib.call("sf_bn_bwd_apply", M, C, dz.data_ptr(), lddz, _ptr(zmask), ldm, y.data_ptr(), ldy, _ptr(sc), _ptr(sh),
             relu_self, coef.data_ptr(), dy.data_ptr(), cl_ld(dy), _ptr(g), cl_ld(g) if g is not None else 0, s,
             work=dict(bytes=2.0 * y.numel() * (3 + mcost + int(want_g))))
    return (dy, g) if want_g else dy


# ------------------------------------------------------------------------------------------------
def _pool_args(y, kernel, stride, padding):
    N, C, T, H, W = y.shape
    (kH, kW), (sH, sW), (pH, pW) = kernel, stride, padding
    return (N, T, H, W, C, kH, kW, sH, sW, pH, pW), ((H + 2 * pH - kH) // sH + 1, (W + 2 * pW - kW) // sW + 1)


def pool_fwd(y, kernel, stride, padding, affine=None, want_argmax=True):
    """MaxPool over (H,W) of act(y), act = producer BN(+ReLU) from ``affine = (scale, shift, relu)``.
    Returns (pooled, argmax): argmax = uint8 window-local index of the first maximum (for pool_bwd); with a ReLU in ``affine``,
    0xFF marks a window whose maximum is not positive (it passes no gradient)."""
    args, (Ho, Wo) = _pool_args(y, kernel, stride, padding)
    N, C, T, H, W = y.shape
    out = cl_empty((N, C, T, Ho, Wo), y.device)
    arg = torch.empty((N, T, Ho, Wo, C), dtype=torch.uint8, device=y.device) if want_argmax else None
    sc, sh, relu = _affine(affine)
    get_lib().call("sf_pool_fwd", *args, y.data_ptr(), cl_ld(y), _ptr(sc), _ptr(sh), relu, out.data_ptr(),
                   cl_ld(out), _ptr(arg), 0, _stream(y), work=dict(bytes=2.0 * (y.numel() + out.numel())))
    return out, arg


def pool_bwd(in_shape, pooled, argmax, dout, kernel, stride, padding, relu=True):
    """Gradient w.r.t. the BatchNorm output feeding relu -> max-pool (argmax gather; the ReLU mask pooled > 0 is
    the code 0xFF of the forward's table -- ``pooled`` itself is not read)."""
    N, C, T, H, W = in_shape
    (kH, kW), (sH, sW), (pH, pW) = kernel, stride, padding
    assert tuple(dout.shape) == tuple(pooled.shape)
    g = cl_empty(in_shape, dout.device)
    get_lib().call("sf_pool_bwd", N, T, H, W, C, kH, kW, sH, sW, pH, pW, pooled.data_ptr(), cl_ld(pooled),
                   argmax.data_ptr(), int(bool(relu)), dout.data_ptr(), cl_ld(dout), g.data_ptr(), cl_ld(g), 0,
                   _stream(dout), work=dict(bytes=2.0 * (g.numel() + 2.5 * dout.numel())))
    return g
