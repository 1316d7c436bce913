"""Forward/backward engine: the fused block schedules and their hand-written backward passes.

Granularity: one ``torch.autograd.Function`` per reference block (stem, FuseFastToSlow, ResBlock).
Inside a block nothing goes through ATen or the autograd tape: the Function runs a fixed schedule of
libsfamd kernels in forward and the mirrored schedule in backward (recomputing BN+ReLU on the fly
from the raw conv outputs it saved).  Autograd only chains blocks.

Fusion plan per bottleneck block (reference graph: slowfast/models/resnet_helper.py:377-392, :512-521):

    forward   ya = conv_a(x)            [+ per-channel sum/sumsq in the epilogue]
              yb = conv_b(relu(bn_a(ya)))    BN+ReLU of `a` applied while loading conv_b's operand
              yc = conv_c(relu(bn_b(yb)))
              out = relu(bn_c(yc) + shortcut)      shortcut = x | bn_1(conv_1(x)), one elementwise pass
    backward  (dyc[, dy1|g]) <- BN backward of dout masked by (out > 0)
              wgrad_c / dgrad_c with relu(bn_b(yb)) recomputed on the fly, ... down to
              dx = dgrad_a(dya) + (g | dgrad_1(dy1))      residual add fused in the dgrad epilogue

Parameter gradients are written straight into ``param.grad`` (fp32; accumulated when it already
exists, which is how GradReducer's flat bucket views receive them) and ``grad_ready`` listeners are
told which parameters are final, so the gradient all-reduce can overlap the rest of backward.
"""
import functools
import os

import torch

from . import lib as _sflib

from . import ops

_f16 = _sflib.act_dtype()        # fp16, or bf16 under SF_ACT_DTYPE=bf16 (lib.ACT_MODE)
_listeners = []
# While a training step is being captured into a HIP graph (slowfast_amd.step.TrainStep) the fp32 -> fp16 weight
# repack must be part of the captured work unconditionally: a replay runs after an optimizer update that the
# host-side version check below never sees.
FORCE_WEIGHT_PREP = False
# Bumped by writers that update parameters behind torch's back (slowfast_amd.optim.FlatOptimizer's fused update kernel does
# not touch tensor._version): part of every packed-weight cache key, so the eager path re-packs after such an update.
PARAM_EPOCH = 0
# Batched weight packing (WeightPackPlan below).  While PACK_RECORD is a list, every unit that packs its weight through a
# per-layer launch appends itself; step.TrainStep turns the record of its first iteration into one sf_prep_weights_batch
# launch per iteration.  SF_PACK_PLAN=0 keeps the per-layer launches (A/B).
PACK_RECORD = None
PACK_PLAN = os.environ.get("SF_PACK_PLAN", "1") != "0"
# Intermediate activations of a block (relu(bn_a(ya)), relu(bn_b(yb))) are materialised in fp16 (default) or recomputed
# in the consumer's operand loads (SF_MATERIALIZE=0, the round-1 schedule; kept for A/B runs).
MATERIALIZE = os.environ.get("SF_MATERIALIZE", "1") != "0"
# The reduction pass of an inner BatchNorm's backward (sums of g and g * y) is taken in the epilogue of the data gradient that
# PRODUCES its input gradient (ops.conv_dgrad(..., bn=...)): one read of the y tile instead of a pass over dz and y, one launch
# less per inner BatchNorm.  SF_BN_FUSE_REDUCE=0 keeps the separate sf_bn_bwd_reduce pass (A/B).
BN_FUSE_REDUCE = os.environ.get("SF_BN_FUSE_REDUCE", "1") != "0"
# Backward segmentation (slowfast_amd.step.TrainStep): models call cut() on the activations that cross a stage boundary.
# Normally the identity.  While a _Segments recorder is installed the tensors are replaced by detached leaves, so that the
# backward pass can be run -- and captured into HIP graphs -- stage by stage (head + res5 first), and the gradient all-reduce
# of a finished stage overlaps the backward of the stages before it.
SEGMENTS = None


class _Segments:
    def __init__(self):
        self.cuts = []          # [(original tensors, leaf tensors)] in forward order

    def cut(self, tensors):
        leaves = [t.detach().requires_grad_(True) if t.requires_grad else t for t in tensors]
        for l, t in zip(leaves, tensors):       # what a block hands its successor through the tensor object travels across the cut
            tag = getattr(t, "_sf_block_bn", None)
            if l is not t and tag is not None:
                l._sf_block_bn = tag
        if any(l is not t for l, t in zip(leaves, tensors)):
            self.cuts.append((list(tensors), leaves))
        return leaves


class WeightPackPlan:
    """fp32 -> fp16 operand packing of MANY weights in one launch (sf_prep_weights_batch).  A training step re-packs every
    layer after the optimizer update; per layer that is ~110 (SlowFast-R50) launches of a few microseconds of work each.
    Built from the units that packed during one recorded iteration (engine.PACK_RECORD); ``run()`` packs them all into
    persistent buffers and marks every unit's operand cache fresh, so their per-layer launches do not happen -- eagerly or
    inside a captured graph (the one launch is captured instead)."""

    def __init__(self, record):
        from ctypes import byref, sizeof
        import numpy as np
        from .lib import PrepItem, get_lib
        lib = get_lib()
        seen, self.entries = set(), []
        for unit, geom, need_dgrad in record:
            k = (id(unit), None if geom is None else geom.Ci)
            if k in seen:
                continue
            seen.add(k)
            self.entries.append((unit, geom, need_dgrad))
        self.device = None
        items = (PrepItem * max(1, len(self.entries)))()
        blk_item, blk_off = [], []
        self._bufs, self._src = [], []
        for i, (unit, geom, need_dgrad) in enumerate(self.entries):
            w, desc, wf, wd = unit.pack_item(geom, need_dgrad)
            assert w.dtype == torch.float32 and w.is_contiguous()
            self.device = w.device
            lib.call("sf_prep_item_fill", byref(desc), w.data_ptr(), wf.data_ptr(), None if wd is None else wd.data_ptr(),
                     byref(items[i]))
            nb = lib.call("sf_prep_item_blocks", byref(items[i]))
            blk_item += [i] * nb
            blk_off += list(range(nb))
            self._bufs.append((wf, wd))
            self._src.append((w, w.data_ptr()))
        self.nblocks = len(blk_item)
        if self.entries:
            raw = np.frombuffer(bytes(items), dtype=np.uint8).copy()
            self.items = torch.from_numpy(raw).to(self.device)
            self.blk_item = torch.tensor(blk_item, dtype=torch.int32, device=self.device)
            self.blk_off = torch.tensor(blk_off, dtype=torch.int32, device=self.device)

    def __len__(self):
        return len(self.entries)

    def valid(self):
        """False once a planned parameter's storage moved (e.g. an optimizer re-pointed it): build a new plan then."""
        return all(w.data_ptr() == ptr for w, ptr in self._src)

    def run(self):
        if not self.entries:
            return
        from .lib import get_lib
        stream = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else None
        get_lib().call("sf_prep_weights_batch", self.items.data_ptr(), self.blk_item.data_ptr(), self.blk_off.data_ptr(),
                       self.nblocks, stream)
        for (unit, geom, _), (wf, wd) in zip(self.entries, self._bufs):
            unit.pack_assign(geom, wf, wd)

    def release(self):
        for unit, _, _ in self.entries:
            unit._planned = False


def cut(x):
    """Stage boundary marker: x (a tensor or a list of tensors) passes through unchanged unless a backward-segmenting
    TrainStep is recording."""
    if SEGMENTS is None:
        return x
    if isinstance(x, (list, tuple)):
        return type(x)(SEGMENTS.cut(list(x)))
    return SEGMENTS.cut([x])[0]


# Weight gradients on a SIDE STREAM.  The backward critical path is the data-gradient chain (dgrad -> BatchNorm backward ->
# dgrad ...); a layer's weight gradient depends on the same dy but nothing downstream depends on IT until the optimizer (or
# the all-reduce of its bucket).  Forked to a second HIP stream it runs concurrently with the bandwidth-bound BatchNorm
# backward / elementwise kernels of the chain -- an MFMA- and latency-bound kernel beside HBM-bound ones, the pairing that
# actually overlaps.  Works under graph capture too (the fork / join become graph edges).  OPT-IN (SF_WGRAD_STREAM=1): on
# ROCm 7.2 a replayed hipGraph runs the two branches back to back -- measured 734 vs 735 clips/s on SlowFast, 523 vs 524
# on MViTv2-S, 1166 vs 1175 on X3D-M (profiles/r2/r2_v15_wgrad_stream_ab.txt) -- so the default keeps the single stream.
# Re-measured at the end of round 6 (profiles/r6_v43_wgrad_stream_ab.txt): X3D-M 1493 vs 1502, MViTv2-S 757 vs 758, SlowFast 825 vs 870.
WGRAD_STREAM = os.environ.get("SF_WGRAD_STREAM", "0") != "0"
_side_streams = {}
_side_keep = []          # tensors the side stream still reads: kept alive until the join (no allocator stream bookkeeping)
_side_pending = set()    # devices with un-joined side-stream work
_join_queued = False


def _fork_wgrad(device):
    """Side stream of `device`, ordered after everything already enqueued on the current stream."""
    global _join_queued
    side = _side_streams.get(device)
    if side is None:
        side = _side_streams[device] = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream(device))
    _side_pending.add(device)
    if not _join_queued:
        try:            # join when the backward pass that started this ends (plain ``loss.backward()`` callers)
            torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
            _join_queued = True
        except RuntimeError:
            pass        # not inside a backward pass: the caller joins explicitly
    return side


def join_side_streams():
    """The current stream waits for the forked weight-gradient work; called at the end of every backward pass / backward
    segment, and before parameters are announced final to a gradient reducer."""
    global _join_queued
    _join_queued = False
    for device in list(_side_pending):
        torch.cuda.current_stream(device).wait_stream(_side_streams[device])
    _side_pending.clear()
    _side_keep.clear()
    for device in {d for d, _ in _pathway_streams}:
        _join_pathways(device)


# Independent PATHWAYS on their own HIP streams (round 5).  Between two lateral connections the Slow and the Fast pathway of a
# SlowFast stage do not depend on each other (slowfast/models/video_model_builder.py:423-441: s_i runs every pathway, s_i_fuse
# joins them), in forward and in backward.  Issued on ONE stream their kernels run back to back: the Slow pathway's MFMA-bound
# convolutions leave a quarter of the CUs idle (196 / 392 tiles on 256 CUs at batch 32) and every kernel pays its own fill and
# drain, the Fast pathway's kernels are thin HBM streams.  Forked onto a second stream the two interleave on the chip -- as
# graph BRANCHES under a captured step -- and fill each other's holes: SlowFast-8x8-R50 41.4 -> 38.7 ms per step
# (profiles/r5_v4_pathway_streams_probe.txt, r5_v5_pathway_streams_ab.txt; the per-pair probe of round 2 had predicted 7 %).  run_pathways() forks before and joins
# after a stage; autograd runs every backward node on the stream of its forward, so the backward forks and joins by itself.
# Rules that keep it exact: scratch memory is per stream (ops._workspace), parameter-gradient consumers join every stream first
# (join_side_streams), tensors that cross a fork / join stay referenced by autograd until their consumers are enqueued.
# SF_PATHWAY_STREAMS=0: one stream (A/B runs).
PATHWAY_STREAMS = os.environ.get("SF_PATHWAY_STREAMS", "1") != "0"
CAT_IN_PLACE = os.environ.get("SF_CAT_IN_PLACE", "1") != "0"       # ResBlockFn writes into FuseFn's buffer (A/B switch)
_pathway_streams = {}    # (device, pathway) -> stream
_pathway_main = {}       # device -> the stream the last fork left from


def _pathway_stream(device, p):
    s = _pathway_streams.get((device, p))
    if s is None:
        s = _pathway_streams[(device, p)] = torch.cuda.Stream(device=device)
    return s


def run_pathways(n, fn, like):
    """[fn(p) for p in range(n)] -- pathway 0 on the current stream, pathways 1 .. n-1 each on its own side stream, forked after
    everything already enqueued on the current stream and joined before this returns.  ``like``: a tensor that tells the device
    (CPU tensors / one pathway / SF_PATHWAY_STREAMS=0: a plain loop)."""
    if not (PATHWAY_STREAMS and n > 1 and like.is_cuda):
        return [fn(p) for p in range(n)]
    dev = like.device
    main = torch.cuda.current_stream(dev)
    _pathway_main[dev] = main
    outs = [None] * n
    sides = []
    for p in range(1, n):
        side = _pathway_stream(dev, p)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            outs[p] = fn(p)
        sides.append(side)
    outs[0] = fn(0)
    for side in sides:
        main.wait_stream(side)
    return outs


# Independent BRANCHES inside a block (the q and the k / v pooling chains of MultiScaleAttention, attention.py:13-45, 320-351:
# three depthwise convolutions + LayerNorms on disjoint channel slices of qkv) on two streams: each of these launches covers a few
# hundred workgroups on a 256-CU chip -- side by side they fill it.  Same fork / join discipline as run_pathways.
BRANCH_STREAMS = os.environ.get("SF_BRANCH_STREAMS", "1") != "0"


def run_branches(fns, like):
    """[fn() for fn in fns] -- fns[0] on the current stream, the others each on a side stream, forked after everything already
    enqueued and joined before this returns."""
    if not (BRANCH_STREAMS and len(fns) > 1 and like.is_cuda):
        return [fn() for fn in fns]
    dev = like.device
    main = torch.cuda.current_stream(dev)
    outs = [None] * len(fns)
    sides = []
    for i in range(1, len(fns)):
        side = _pathway_stream(dev, ("branch", i))
        side.wait_stream(main)
        with torch.cuda.stream(side):
            outs[i] = fns[i]()
        sides.append(side)
    outs[0] = fns[0]()
    for side in sides:
        main.wait_stream(side)
    return outs


def _join_pathways(device):
    """The CURRENT stream of ``device`` waits for every pathway stream and for the stream they were forked from: whatever it
    enqueues next (a gradient all-reduce, an optimizer pass) sees the gradients of all pathways."""
    cur = torch.cuda.current_stream(device)
    for (d, _), s in _pathway_streams.items():
        if d == device and s != cur:
            cur.wait_stream(s)
    main = _pathway_main.get(device)
    if main is not None and main != cur:
        cur.wait_stream(main)


# Test hook: when a list, every Function with a ReLU or a max-pool appends the tensors that decide its backward routing (raw conv
# outputs + BatchNorm scale / shift, the block output, byte arg-max tables) together with its module, so that a parity test
# can hand the SAME masks / routes to the oracle's backward -- per block (tests/block_checks.py) and for a whole model
# (tests/model_checks.py:engine_masks).
CAPTURE = None


def add_grad_ready_listener(fn):
    _listeners.append(fn)
    return fn


def remove_grad_ready_listener(fn):
    if fn in _listeners:
        _listeners.remove(fn)


_sub_passes = 1
_sub_counts = {}


_sub_params = None


def hold_notifications(passes, params=None):
    """A SubBatchNorm3d(S) training step runs S sub-batch passes whose backward passes all add into the same
    parameter gradients (batchnorm.run_in_splits): a parameter is announced final after its S-th contribution.
    ``params`` restricts the rule to one model's parameters (others are announced immediately)."""
    global _sub_passes, _sub_params
    _sub_passes = max(int(passes), 1)
    _sub_params = None if params is None or _sub_passes == 1 else set(params)
    _sub_counts.clear()


def _always():
    return True


def _queue_join_at_end_of_backward():
    """Backward nodes of the Fast pathway / of the k-v branch write param.grad in place on THEIR streams.  TrainStep and the
    reducer join explicitly; a plain ``loss.backward(); optimizer.step()`` caller does not, so the join rides on the backward
    pass itself (the same end-of-pass callback the side-stream weight gradients use): whatever the caller enqueues after
    ``backward()`` returns sees every pathway's gradients.  (before this it held only because autograd's leaf-stream
    synchronisation happened to cover the AccumulateGrad nodes created under the side stream.)"""
    global _join_queued
    if _join_queued or not _pathway_streams:
        return
    try:
        torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
        _join_queued = True
    except RuntimeError:
        pass            # not inside a backward pass (a segment driven by TrainStep, which joins itself)


def _notify(params):
    _queue_join_at_end_of_backward()
    if GRADS_VIA_AUTOGRAD:          # the consumer of the gradients (DDP's reducer) hooks autograd itself
        return
    if (_side_pending or _pathway_streams) and any(getattr(fn, "needs_join", _always)() for fn in _listeners):
        join_side_streams()         # a listener is about to start the all-reduce of these gradients: they must be complete
    if _sub_passes > 1:
        final = []
        for p in params:
            if _sub_params is not None and p not in _sub_params:
                final.append(p)
                continue
            c = _sub_counts.get(p, 0) + 1
            if c >= _sub_passes:
                _sub_counts.pop(p, None)
                final.append(p)
            else:
                _sub_counts[p] = c
        params = final
        if not params:
            return
    for fn in _listeners:
        fn(params)


def _sync_of(bn):
    """(process group, group size) when ``bn`` exchanges its statistics across ranks (NaiveSyncBatchNorm3d), else None."""
    fn = getattr(bn, "sync_group", None)
    return fn() if fn is not None else None


def bn_statistics(bn, part, count, C, training):
    """Partial sums of a producer epilogue -> BNState (scale, shift, mean, rstd); running statistics updated in
    training mode.  Shared by ConvUnit and the X3D units.  With a sync group the [2, C] sums are all-reduced first
    (equal per-rank counts, as pytorchvideo's NaiveSyncBatchNorm assumes) and the running statistics follow that class:
    momentum update with the BIASED batch variance."""
    use_batch = training or bn.running_mean is None
    track = bn.track_running_stats and training
    sync = _sync_of(bn) if use_batch else None
    if sync is None:
        st = ops.bn_finalize(part if use_batch else None, count, bn.weight, bn.bias,
                             bn.running_mean if track or not use_batch else None,
                             bn.running_var if track or not use_batch else None, bn.momentum, bn.eps,
                             training=use_batch, C=C)
        return BNState(*st)
    import torch.distributed as dist
    group, gsize = sync
    tot = part.sum(0, keepdim=True)                      # [1, 2, C] fp32
    dist.all_reduce(tot, group=group)
    total = float(count) * gsize
    if track:
        Cr = bn.weight.numel()
        mean = tot[0, 0, :Cr] / total
        var = tot[0, 1, :Cr] / total - mean * mean
        bn.running_mean.add_(bn.momentum * (mean - bn.running_mean))
        bn.running_var.add_(bn.momentum * (var - bn.running_var))
    st = ops.bn_finalize(tot, total, bn.weight, bn.bias, None, None, bn.momentum, bn.eps, training=True, C=C)
    return BNState(*st)


def as_cl(t):
    """Gradients/activations entering a block from torch code may be fp32 or NCTHW-contiguous."""
    if t.dtype == _f16 and ops.is_cl(t):
        return t
    return ops.to_cl(t)


# Gradient delivery.  Default (False): backward kernels write d(loss)/d(param) straight into ``param.grad`` -- views of
# GradReducer's flat bucket memory -- and announce finished parameters to the grad-ready listeners.  True: every
# autograd.Function RETURNS its parameter gradients instead, so that they reach the parameters through autograd's
# AccumulateGrad nodes -- which is where torch.nn.parallel.DistributedDataParallel (the wrap the reference's build_model
# applies, slowfast/models/build.py:64-80) hooks its bucketed all-reduce and any register_comm_hook() hook.
GRADS_VIA_AUTOGRAD = False
_pending_grads = {}


def record_params(ctx, params):
    """Function.forward: remember the ``*params`` it received and the gradient-delivery mode of THIS iteration of THIS model.
    The global above says which mode the model that is running its forward is in (GradReducer.zero_grad() / the DDP forward
    pre-hook set it right before); the backward may run after another model's forward has changed it, so it goes by what the
    forward recorded (delivers_grads)."""
    ctx._sf_params = params
    ctx._sf_via_autograd = GRADS_VIA_AUTOGRAD


def delivers_grads(backward):
    """Decorator of a Function.backward that writes parameter gradients (_grad_dest / _notify / param_grads): runs it under the
    delivery mode its forward recorded."""
    @functools.wraps(backward)
    def wrapped(ctx, *grads):
        global GRADS_VIA_AUTOGRAD
        prev = GRADS_VIA_AUTOGRAD
        GRADS_VIA_AUTOGRAD = getattr(ctx, "_sf_via_autograd", prev)
        try:
            return backward(ctx, *grads)
        finally:
            GRADS_VIA_AUTOGRAD = prev
    return wrapped


def _grad_dest(param):
    """(tensor to write d(loss)/d(param) into, whether the kernel must clear it first)."""
    if GRADS_VIA_AUTOGRAD:
        g = _pending_grads.get(id(param))
        if g is not None:
            return g, False
        g = torch.empty_like(param, memory_format=torch.contiguous_format)
        _pending_grads[id(param)] = g
        return g, True
    if param.grad is not None:
        assert param.grad.dtype == torch.float32 and param.grad.is_contiguous()
        return param.grad, False
    g = torch.empty_like(param, memory_format=torch.contiguous_format)
    param.grad = g
    return g, True


def param_grads(ctx, lead, extra=()):
    """The entries of a Function.backward return that belong to the ``*params`` the forward received after its ``lead``
    leading arguments: Nones when gradients are written in place, the collected tensors under GRADS_VIA_AUTOGRAD.
    ``extra``: parameters passed among the leading arguments; their collected gradients are returned as a second tuple."""
    n = len(ctx.needs_input_grad) - lead
    if not GRADS_VIA_AUTOGRAD:
        return ((None,) * n, (None,) * len(extra)) if extra else (None,) * n
    ps = getattr(ctx, "_sf_params", ())
    assert len(ps) == n, "Function.forward must record its *params (ctx._sf_params)"
    out = tuple(_pending_grads.pop(id(p), None) for p in ps)
    if extra:
        return out, tuple(_pending_grads.pop(id(p), None) for p in extra)
    return out


class BNState:
    __slots__ = ("scale", "shift", "mean", "rstd")

    def __init__(self, scale, shift, mean, rstd):
        self.scale, self.shift, self.mean, self.rstd = scale, shift, mean, rstd


class ConvUnit:
    """Binds an nn.Conv3d (+ its nn.BatchNorm3d) parameter container to the conv/BN kernels.

    The nn modules stay plain parameter holders so state_dict keys, weight init
    (slowfast/utils/weight_init_helper.py:10-54) and optimizer param grouping
    (slowfast/models/optimizer.py:41-56) of the reference keep working on the drop-ins."""

    def __init__(self, conv, bn=None):
        assert conv.groups == 1, "grouped convolutions are handled by the depthwise path, not the implicit GEMM"
        assert conv.padding_mode == "zeros"
        assert bn is None or bn.momentum is not None, "cumulative-average BatchNorm is not supported"
        self.conv, self.bn = conv, bn
        self._geoms = {}
        self._wkey, self._w, self._planned = None, None, False

    def geom(self, in_shape):
        key = tuple(in_shape)
        g = self._geoms.get(key)
        if g is None:
            c = self.conv
            g = ops.ConvGeom(key, c.out_channels, c.kernel_size, c.stride, c.padding, c.dilation, Cw=c.in_channels)
            self._geoms[key] = g
        return g

    def weights(self, geom, fresh=False):
        """fp16 GEMM operands of the current weight.  ``fresh`` (forward pass) forces the repack while a training
        step is being captured; the backward pass of the same step reuses what its forward packed."""
        w = self.conv.weight
        key = (w.data_ptr(), w._version, geom.Ci, w.device, PARAM_EPOCH)
        if (fresh and FORCE_WEIGHT_PREP and not self._planned) or self._wkey != key:
            # no data-gradient operand for the RGB stems (3 of 8 channels real); channel padding (54 -> 56) keeps it
            need_dgrad = geom.Ci - geom.Cw < 8
            self._w = ops.prep_weights(w.detach(), geom, need_dgrad=need_dgrad)
            self._wkey = key
            if PACK_RECORD is not None:
                PACK_RECORD.append((self, geom, need_dgrad))
        return self._w

    # -- WeightPackPlan protocol: one entry of the batched packing launch ------------------------------------------------
    def pack_item(self, geom, need_dgrad):
        w = self.conv.weight
        wf = torch.empty((geom.Co, geom.ldf), dtype=_f16, device=w.device)
        wd = torch.empty((geom.Ci, geom.ldd), dtype=_f16, device=w.device) if need_dgrad else None
        return w, geom.desc(geom.Ci, geom.Co), wf, wd

    def pack_key(self, geom):
        w = self.conv.weight
        return (w.data_ptr(), w._version, geom.Ci, w.device, PARAM_EPOCH)

    def pack_assign(self, geom, wf, wd):
        self._w, self._wkey, self._planned = (wf, wd), self.pack_key(geom), True

    def forward(self, x, in_affine, training):
        geom = self.geom(x.shape)
        wf, _ = self.weights(geom, fresh=True)
        bn = self.bn
        if bn is None:
            y, _ = ops.conv_fwd(x, wf, geom, in_affine=in_affine, bias=self.conv.bias, stats=False)
            return y, None
        use_batch_stats = training or bn.running_mean is None
        y, part = ops.conv_fwd(x, wf, geom, in_affine=in_affine, bias=self.conv.bias, stats=use_batch_stats)
        return y, bn_statistics(bn, part, geom.out_rows, geom.Co, training)

    def backward(self, x, in_affine, dy, need_dx, resid=None, resid_bits=None, bn_fuse=None):
        """Weight gradient into conv.weight.grad; returns dx (+ resid, masked by resid_bits when given) when need_dx.
        ``bn_fuse = (y, BNState)`` of the BatchNorm-ReLU that produced x: returns (dx, part) with the reduction pass of that
        BatchNorm's backward taken in the data gradient's epilogue (part None: not available for this geometry);
        ``bn_fuse = {"bits", "y0"}`` (x is the previous block's output): (dx, part) for that block's final BatchNorm."""
        geom = self.geom(x.shape)
        w = self.conv.weight
        if w.requires_grad:
            dw, zero_first = _grad_dest(w)
            if WGRAD_STREAM and x.is_cuda and not GRADS_VIA_AUTOGRAD:
                side = _fork_wgrad(x.device)
                _side_keep.extend((x, dy))
                with torch.cuda.stream(side):
                    ops.conv_wgrad(x, dy, geom, dw, in_affine=in_affine, out_scale=1.0, zero_first=zero_first, side=True)
            else:
                ops.conv_wgrad(x, dy, geom, dw, in_affine=in_affine, out_scale=1.0, zero_first=zero_first)
        if not need_dx:
            return None
        _, wd = self.weights(geom)
        if bn_fuse is not None:
            if isinstance(bn_fuse, dict):       # block input = the previous block's output
                return ops.conv_dgrad(dy, wd, geom, resid=resid, resid_bits=resid_bits, bn=bn_fuse)
            y, st = bn_fuse
            return ops.conv_dgrad(dy, wd, geom, resid=resid, bn=(y, st.scale, st.shift))
        return ops.conv_dgrad(dy, wd, geom, resid=resid, resid_bits=resid_bits)

    def bn_backward(self, dz, y, st, zmask=None, relu_self=False, want_g=False, part=None):
        """BatchNorm3d backward (through ReLU) -> dy; writes bn.weight.grad / bn.bias.grad.  ``part``: partial sums already
        taken by the producer of dz (backward(..., bn_fuse=...))."""
        bn = self.bn
        if bn.weight.requires_grad:
            dgamma, zg = _grad_dest(bn.weight)
            dbeta, zb = _grad_dest(bn.bias)
            assert zg == zb
            accumulate = not zg
        else:
            dgamma = torch.empty_like(bn.weight)
            dbeta = torch.empty_like(bn.bias)
            accumulate = False
        return ops.bn_bwd(dz, y, bn.weight, st.mean, st.rstd, dgamma, dbeta, zmask=zmask,
                          relu_affine=(st.scale, st.shift) if relu_self else None, inv_loss_scale=1.0,
                          accumulate=accumulate, want_g=want_g, sync=_sync_of(bn), part=part)

    def params(self):
        p = [self.conv.weight]
        if self.conv.bias is not None:
            p.append(self.conv.bias)
        if self.bn is not None:
            p += [self.bn.weight, self.bn.bias]
        return p

    # ---- inference fusion (slowfast_amd.inference) --------------------------------------
    def _fold_source(self, w):
        """The fp32 Conv3d-layout weight the packed operand is made from (hook for the W-pair-folded stems)."""
        return w

    def fold(self):
        """Fold the eval-mode BatchNorm (an affine map of the running statistics, as F.batch_norm applies it with
        training=False) into the convolution: w' = w * scale[co], b' = (conv.bias) * scale + shift.  One-time
        preparation on the parameters' device; the packed fp16 operands are cached per input geometry."""
        w = self.conv.weight.detach().float()
        b = self.conv.bias.detach().float() if self.conv.bias is not None else None
        bn = self.bn
        if bn is not None:
            assert bn.running_mean is not None, "BatchNorm without running statistics cannot be folded"
            scale = torch.rsqrt(bn.running_var.detach().float() + bn.eps)
            if bn.weight is not None:
                scale = scale * bn.weight.detach().float()
            shift = -bn.running_mean.detach().float() * scale
            if bn.bias is not None:
                shift = shift + bn.bias.detach().float()
            w = w * scale.view(-1, 1, 1, 1, 1)
            b = shift if b is None else b * scale + shift
        self._fold_w, self._fold_b, self._fold_packed = self._fold_source(w).contiguous(), b, {}

    def infer(self, x, relu=False, resid=None, out=None):
        """relu?(bn(conv(x)) [+ resid]) in ONE launch from the folded operands (no statistics, nothing saved)."""
        geom = self.geom(x.shape)
        packed = self._fold_packed.get(geom.in_shape)
        if packed is None:
            wf, _ = ops.prep_weights(self._fold_w, geom, need_dgrad=False)
            b = self._fold_b
            if b is not None and b.numel() < geom.Co:      # channel padding: pad channels stay exact zeros
                b = torch.nn.functional.pad(b, (0, geom.Co - b.numel()))
            packed = self._fold_packed[geom.in_shape] = (wf, None if b is None else b.contiguous())
        return ops.conv_fwd_fused(x, packed[0], geom, bias=packed[1], resid=resid, relu=relu, out=out)


class StemConvUnit(ConvUnit):
    """Conv3d with <= 4 input channels (the RGB stems) as a W-pair-folded implicit GEMM.

    A 3-channel clip padded to 8 channels would waste 5/8 of every MFMA K-step.  Instead the clip is stored
    N,T,H,W,4 and read as N,T,H,W/2,8: one 16-byte operand group = two neighbouring pixels x 4 channels.  The
    (kT,kH,kW) stride-(.,.,sW) convolution becomes a (kT,kH,kW2) stride-(.,.,sW/2) convolution over that view with
    the kW axis zero-extended to an even length starting on an even pixel (one leading zero tap when pW is odd),
    so K = taps2*8 carries 3/4 * kW/(2 kW2) useful elements (66 % for 7 taps) instead of 3/8.  The virtual weight
    [Co][8][kT][kH][kW2] is gathered from / scattered to the nn.Conv3d parameter with a few tiny tensor ops."""

    def __init__(self, conv, bn=None):
        super().__init__(conv, bn)
        kT, kH, kW = conv.kernel_size
        sT, sH, sW = conv.stride
        pT, pH, pW = conv.padding
        assert conv.in_channels <= 4 and sW % 2 == 0 and conv.dilation == (1, 1, 1)
        self.lead = pW % 2
        self.kext = kW + self.lead + ((kW + self.lead) % 2)
        self.k2 = (kT, kH, self.kext // 2)
        self.s2 = (sT, sH, sW // 2)
        self.p2 = (pT, pH, (pW + self.lead) // 2)

    def prepare_input(self, x):
        """NCTHW fp32 clip -> the W-pair view (N, 8, T, H, W/2) fp16; clips packed by data.pack_pathways_u8 are already
        in that layout."""
        if getattr(x, "_sf_wpairs", False):
            return x
        return ops.ncthw_to_cl_wpairs(x.float())

    @staticmethod
    def prepare_shape(shape):
        N, C, T, H, W = shape
        return (N, 8, T, H, W // 2)

    def geom(self, in_shape):
        key = tuple(in_shape)
        g = self._geoms.get(key)
        if g is None:
            c = self.conv
            N, C2, T, H, W2 = key
            assert C2 == 8
            Wo = (2 * W2 + 2 * c.padding[2] - c.kernel_size[2]) // c.stride[2] + 1
            g = ops.ConvGeom(key, c.out_channels, self.k2, self.s2, self.p2, (1, 1, 1), Cw=8)
            g = ops.ConvGeom(key, c.out_channels, self.k2, self.s2, self.p2, (1, 1, 1), Cw=8,
                             out_dims=(g.To, g.Ho, Wo))
            self._geoms[key] = g
        return g

    def _virtual_weight(self, w):
        Co, Cin, kT, kH, kW = w.shape
        wv = torch.nn.functional.pad(w, (self.lead, self.kext - kW - self.lead, 0, 0, 0, 0, 0, 4 - Cin))
        wv = wv.view(Co, 4, kT, kH, self.kext // 2, 2).permute(0, 5, 1, 2, 3, 4)
        return wv.reshape(Co, 8, kT, kH, self.kext // 2).contiguous()

    def _fold_source(self, w):
        return self._virtual_weight(w)

    def weights(self, geom, fresh=False):
        w = self.conv.weight
        key = (w.data_ptr(), w._version, geom.Ci, w.device, PARAM_EPOCH)
        if (fresh and FORCE_WEIGHT_PREP) or self._wkey != key:
            self._w = ops.prep_weights(self._virtual_weight(w.detach()), geom, need_dgrad=False)
            self._wkey = key
        return self._w

    def backward(self, x, in_affine, dy, need_dx, resid=None):
        assert not need_dx, "the stems take the input clip: no data gradient"
        geom = self.geom(x.shape)
        w = self.conv.weight
        if w.requires_grad:
            Co, Cin, kT, kH, kW = w.shape
            dwv = torch.empty((Co, 8, kT, kH, self.kext // 2), dtype=torch.float32, device=w.device)
            ops.conv_wgrad(x, dy, geom, dwv, in_affine=in_affine, out_scale=1.0, zero_first=True)
            dw = dwv.view(Co, 2, 4, kT, kH, self.kext // 2).permute(0, 2, 3, 4, 5, 1).reshape(Co, 4, kT, kH, self.kext)
            dw = dw[:, :Cin, :, :, self.lead:self.lead + kW]
            dst, zero_first = _grad_dest(w)
            if zero_first:
                dst.copy_(dw)
            else:
                dst.add_(dw)
        return None


# ------------------------------------------------------------------------------------------------
class StemFn(torch.autograd.Function):
    """conv -> BN -> ReLU -> MaxPool3d([1,k,k]) (slowfast/models/stem_helper.py:196-201)."""

    @staticmethod
    def forward(ctx, x, mod, *params):
        record_params(ctx, params)
        unit = mod._unit
        xcl = unit.prepare_input(x) if isinstance(unit, StemConvUnit) else ops.to_cl(x)
        y, st = unit.forward(xcl, None, mod.training)
        k, s, p = mod.pool_layer.kernel_size, mod.pool_layer.stride, mod.pool_layer.padding
        assert k[0] == 1 and s[0] == 1 and p[0] == 0, "stem pooling is spatial-only in every reference config"
        out, arg = ops.pool_fwd(y, k[1:], s[1:], p[1:], affine=(st.scale, st.shift, True))
        if CAPTURE is not None:
            # code 0xFF = a window the ReLU did not pass (every entry is 0 after it; sf_pool_fwd): any route is the true maximum and
            # carries no gradient -> the checks are handed tap 0
            CAPTURE.append({"kind": "stem", "mod": mod, "raw": [y], "bn": [(st.scale, st.shift)], "out": out,
                            "argmax": torch.where(arg == 255, torch.zeros_like(arg), arg)})
        ctx.mod, ctx.xcl, ctx.y, ctx.st = mod, xcl, y, st
        ctx.pool = (tuple(k[1:]), tuple(s[1:]), tuple(p[1:]))
        ctx.pooled, ctx.arg = out, arg
        return out

    @staticmethod
    @delivers_grads
    def backward(ctx, dout):
        unit = ctx.mod._unit
        st = ctx.st
        g = ops.pool_bwd(ctx.y.shape, ctx.pooled, ctx.arg, as_cl(dout), *ctx.pool, relu=True)
        dy = unit.bn_backward(g, ctx.y, st)
        unit.backward(ctx.xcl, None, dy, need_dx=False)
        _notify(unit.params())
        ctx.xcl = ctx.y = ctx.pooled = ctx.arg = None
        return (None, None) + param_grads(ctx, 2)


class FuseFn(torch.autograd.Function):
    """FuseFastToSlow: cat([x_s, relu(bn(conv_f2s(x_f)))], 1) (video_model_builder.py:162-169).

    The lateral branch is written directly into its channel slice of the concatenated buffer."""

    @staticmethod
    def forward(ctx, x_s, x_f, mod, *params):
        record_params(ctx, params)
        unit = mod._unit
        x_s, x_f = as_cl(x_s), as_cl(x_f)
        yf, st = unit.forward(x_f, None, mod.training)
        N, Cs, T, H, W = x_s.shape
        Cf = yf.shape[1]
        assert tuple(yf.shape) == (N, Cf, T, H, W), "lateral connection does not match the Slow pathway shape"
        cat = getattr(x_s, "_sf_cat", None)
        if cat is not None and tuple(cat.shape) == (N, Cs + Cf, T, H, W) and cat.data_ptr() == x_s.data_ptr():
            pass                                        # the Slow block already wrote its slice (ResBlockFn, _cat_extra)
        else:
            cat = ops.cl_empty((N, Cs + Cf, T, H, W), x_s.device)
            ops.bn_act(x_s, out=cat[:, :Cs])
        ops.bn_act(yf, st.scale, st.shift, relu=True, out=cat[:, Cs:])
        if CAPTURE is not None:
            CAPTURE.append({"kind": "fuse", "mod": mod, "raw": [yf], "bn": [(st.scale, st.shift)]})
        ctx.mod, ctx.yf, ctx.st, ctx.Cs = mod, yf, st, Cs
        ctx.save_for_backward(x_f)
        return cat, x_f.view_as(x_f)

    @staticmethod
    @delivers_grads
    def backward(ctx, dcat, dxf):
        unit = ctx.mod._unit
        (x_f,) = ctx.saved_tensors
        dcat = as_cl(dcat)
        dyf = unit.bn_backward(dcat[:, ctx.Cs:], ctx.yf, ctx.st, relu_self=True)
        need_dx = ctx.needs_input_grad[1]
        dx_f = unit.backward(x_f, None, dyf, need_dx=need_dx, resid=as_cl(dxf) if dxf is not None else None)
        _notify(unit.params())
        ctx.yf = None
        dx_s = dcat[:, :ctx.Cs] if ctx.needs_input_grad[0] else None
        return (dx_s, dx_f, None) + param_grads(ctx, 3)


class ResBlockFn(torch.autograd.Function):
    """relu(shortcut(x) + transform(x)) (slowfast/models/resnet_helper.py:512-521) for a transform that is a chain of
    conv -> BN [-> ReLU] units ending in the block-final BN: BottleneckTransform (a, b, c; :377-392) and BasicTransform
    (a, b; :105-115).  Every unit after the first applies its producer's BatchNorm + ReLU in its operand loads."""

    @staticmethod
    def forward(ctx, x, mod, *params):
        record_params(ctx, params)
        # when x is the previous block's output, that block left what ITS final BatchNorm backward will reduce over (see the
        # end of this function): this block's last data gradient produces exactly that gradient and reduces it in its epilogue
        ctx.prev_bn = getattr(x, "_sf_block_bn", None) if BN_FUSE_REDUCE else None
        x = as_cl(x)
        units, P = mod.branch2._chain, mod._proj
        tr = mod.training
        raw, bn, act = [], [], []
        h, prologue = x, None
        for i, u in enumerate(units):
            h, st = u.forward(h, prologue, tr)
            raw.append(h)
            bn.append(st)
            if MATERIALIZE and i + 1 < len(units):
                # z = relu(bn(y)) written once (C/4-wide tensors): the consumer convolution and its weight gradient
                # then read a plain operand -- direct-to-LDS copies, no per-element work in the GEMM loaders
                h = ops.bn_act(h, st.scale, st.shift, relu=True)
                act.append(h)
                prologue = None
            else:
                act.append(None)
                prologue = (st.scale, st.shift, True)
        yc, sc = raw[-1], bn[-1]
        # the backward pass needs only the SIGN of the block output (ReLU mask): one bit per element, written here, stands
        # in for two full reads of `out` in BatchNorm backward (and for the masked gradient tensor of the identity shortcut)
        # The last Slow block of a stage that a lateral connection follows (video_models.SlowFast marks it: _cat_extra = channels
        # of the lateral branch) writes its output straight into the channel slice of the concatenated buffer FuseFn would
        # otherwise copy it into (one read + one write of the widest tensor of the stage less per lateral connection).
        out_view, cat = None, None
        extra = getattr(mod, "_cat_extra", 0)
        if extra and CAT_IN_PLACE:
            N_, C_, T_, H_, W_ = yc.shape
            cat = ops.cl_empty((N_, C_ + extra, T_, H_, W_), x.device)
            out_view = cat[:, :C_]
        if P is not None:
            y1, s1 = P.forward(x, None, tr)
            out, bits = ops.bn_act(yc, sc.scale, sc.shift, relu=True, resid=y1, rscale=s1.scale, rshift=s1.shift,
                                   want_mask=True, out=out_view)
        else:
            y1, s1 = None, None
            out, bits = ops.bn_act(yc, sc.scale, sc.shift, relu=True, resid=x, want_mask=True, out=out_view)
        if cat is not None:
            out._sf_cat = cat
        if CAPTURE is not None:
            CAPTURE.append({"kind": "resblock", "mod": mod, "raw": list(raw), "bn": [(b.scale, b.shift) for b in bn], "out": out})
        ctx.mod = mod
        ctx.raw = (raw, y1, act, bits)
        ctx.bn = (bn, s1)
        ctx.save_for_backward(x)
        if BN_FUSE_REDUCE and tr:
            # a plain attribute of the output tensor: it reaches the next block only when that block receives THIS tensor
            # (consecutive blocks of a stage); any op in between (fusion, pooling, a stage cut) drops it and nothing changes
            out._sf_block_bn = {"bits": bits, "y0": yc, "sync": _sync_of(units[-1].bn) is not None}
        return out

    @staticmethod
    @delivers_grads
    def backward(ctx, dout):
        mod = ctx.mod
        units, P = mod.branch2._chain, mod._proj
        (x,) = ctx.saved_tensors
        raw, y1, act, bits = ctx.raw
        bn, s1 = ctx.bn
        last = len(units) - 1
        # partial sums the consumer block's data gradient already took over dout (tagged with the tensor state they describe)
        part_c = tagged_bn_part(dout, raw[last])
        dout = as_cl(dout)
        need_dx = ctx.needs_input_grad[0]
        dy = units[last].bn_backward(dout, raw[last], bn[last], zmask=bits, part=part_c)
        if P is not None:
            dy1 = P.bn_backward(dout, y1, s1, zmask=bits)
        for i in range(last, 0, -1):
            part = None
            if act[i - 1] is not None and BN_FUSE_REDUCE and _sync_of(units[i - 1].bn) is None:
                d_in, part = units[i].backward(act[i - 1], None, dy, need_dx=True, bn_fuse=(raw[i - 1], bn[i - 1]))
            elif act[i - 1] is not None:
                d_in = units[i].backward(act[i - 1], None, dy, need_dx=True)
            else:
                d_in = units[i].backward(raw[i - 1], (bn[i - 1].scale, bn[i - 1].shift, True), dy, need_dx=True)
            dy = units[i - 1].bn_backward(d_in, raw[i - 1], bn[i - 1], relu_self=True, part=part)
        prev = ctx.prev_bn if need_dx else None
        if prev is not None and prev["sync"]:      # the PRODUCER's BatchNorm reduces its sums across ranks: not fused
            prev = None
        if P is not None:
            dx1 = P.backward(x, None, dy1, need_dx=need_dx)
            dx = units[0].backward(x, None, dy, need_dx=need_dx, resid=dx1, bn_fuse=prev)
        else:       # identity shortcut: dx = dgrad_a + dout * (out > 0), the mask applied to the residual in the epilogue
            dx = units[0].backward(x, None, dy, need_dx=need_dx, resid=dout, resid_bits=bits, bn_fuse=prev)
        if prev is not None:
            dx, pc = dx
            if pc is not None:
                tag_bn_part(dx, prev["y0"], pc)
        _notify(mod._param_list)
        ctx.raw = ctx.bn = ctx.prev_bn = None
        return (dx, None) + param_grads(ctx, 2)


CUT_BACKWARD = False     # True while step.TrainStep runs one segment of a cut backward pass
_cut_bn_tags = {}        # gradient storage -> tag recorded during such a segment (see retag_cut_grad)


def retag_cut_grad(g):
    """A gradient that crosses a stage cut arrives as ``leaf.grad``: autograd's accumulation detaches it into a new tensor
    object (same storage, same version counter) and the Python attribute tag_bn_part() put on it is gone -- the producer block
    would fall back to the separate reduction pass and the segmented backward would round differently from the unsegmented one
    (X3D / ResNet stages end in a block, SlowFast's in a fusion layer).  step.TrainStep calls this on every leaf gradient before
    it starts the next segment: the tag recorded for that storage is put back if it still describes the tensor."""
    tag = _cut_bn_tags.pop(g.data_ptr(), None)
    if tag is not None and getattr(g, "_sf_bn_part", None) is None and tag[2] == g.data_ptr() and tag[3] == g._version:
        g._sf_bn_part = tag


def tag_bn_part(dx, y0, part):
    """Attach the BatchNorm-backward partial sums a data-gradient epilogue took over ``dx`` (the gradient w.r.t. the previous
    block's output) to that tensor.  The tag names the tensor state it describes: the raw BatchNorm input it belongs to, the
    gradient's storage and its version counter -- autograd may accumulate another consumer's gradient IN PLACE into the same
    tensor object (feature taps, auxiliary heads, hooks), which keeps the Python attribute and changes the values."""
    dx._sf_bn_part = (y0.data_ptr(), part, dx.data_ptr(), dx._version)
    if CUT_BACKWARD:
        _cut_bn_tags[dx.data_ptr()] = dx._sf_bn_part


def tagged_bn_part(dout, y0):
    """The partial sums tagged onto ``dout`` if they still describe it (same BatchNorm input, same storage, not modified
    since), else None -- the caller then runs the separate reduction pass."""
    tag = getattr(dout, "_sf_bn_part", None)
    if tag is None or tag[0] != y0.data_ptr() or tag[2] != dout.data_ptr() or tag[3] != dout._version:
        return None
    return tag[1]


class ConvBNActFn(torch.autograd.Function):
    """Stand-alone conv -> BN -> (ReLU), materialised (used by BottleneckTransform.forward on its own)."""

    @staticmethod
    def forward(ctx, x, unit, relu, training, *params):
        record_params(ctx, params)
        x = as_cl(x)
        y, st = unit.forward(x, None, training)
        out = ops.bn_act(y, st.scale, st.shift, relu=relu)
        if CAPTURE is not None:
            CAPTURE.append({"kind": "convbnact", "mod": None, "raw": [y], "bn": [(st.scale, st.shift)], "out": out})
        ctx.unit, ctx.relu, ctx.y, ctx.st = unit, relu, y, st
        ctx.save_for_backward(x)
        return out

    @staticmethod
    @delivers_grads
    def backward(ctx, dout):
        unit = ctx.unit
        (x,) = ctx.saved_tensors
        dy = unit.bn_backward(as_cl(dout), ctx.y, ctx.st, relu_self=ctx.relu)
        dx = unit.backward(x, None, dy, need_dx=ctx.needs_input_grad[0])
        _notify(unit.params())
        ctx.y = None
        return (dx, None, None, None) + param_grads(ctx, 4)
