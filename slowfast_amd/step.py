"""One training iteration of the hot path as a replayable HIP graph.

Mirrors the inner loop of the reference's ``train_epoch`` (tools/train_net.py:104-172): zero_grad ->
``preds = model(inputs)`` -> ``loss = loss_fun(preds, labels)`` -> ``loss.backward()`` -> gradient all-reduce ->
``optimizer.step()``.  A SlowFast-R50 step is ~2000 short kernel launches issued from Python; on MI355X the
kernels of a bs32 step take tens of milliseconds, the same order as the host time needed to issue them one by
one.  ``TrainStep`` therefore captures [zero_grad, forward, loss, backward] ONCE into a HIP graph
(``torch.cuda.CUDAGraph`` is hipGraph on ROCm; every libsfamd launch goes to the capturing stream) and replays
it per iteration; shapes are static (synthetic or fixed-size clips), inputs are copied into static buffers.
The optimizer update runs eagerly after the replay.

Multi-GPU overlap.  The backward pass is SEGMENTED at the stage boundaries the models mark with ``engine.cut()``: the
forward (+ loss) is one graph, the backward of [head + res5], [res4], [res3 .. stem] are separate graphs replayed in that
order.  After each backward segment the buckets of the flat gradient buffer that are now complete are all-reduced
asynchronously (RCCL runs on its own stream; c10d orders it after the segment's kernels), i.e. the reduction of the late
stages -- res5 alone holds ~60 % of SlowFast-R50's parameters -- overlaps the backward of the early, activation-heavy
stages, as DDP's reducer does for the reference (slowfast/models/build.py:64-76).  With one rank the segments still
replay back to back and execute the same kernels as the unsegmented graph.

Without a GPU (host-simulator tests) or with ``use_graph=False`` the same sequence runs eagerly.
"""
import torch

from . import engine


class TrainStep:
    def __init__(self, model, reducer, optimizer, loss_fn, loss_scale=1.0, use_graph=None, warmup=2, clip_grad_l2norm=None,
                 clip_grad_val=None, track_stats=False, segmented=None):
        self.model, self.reducer, self.optimizer, self.loss_fn = model, reducer, optimizer, loss_fn
        self.loss_scale = float(loss_scale)
        # SOLVER.CLIP_GRAD_L2NORM / CLIP_GRAD_VAL (tools/train_net.py:156-166); the global gradient norm is always
        # computed, as the reference does, but stays on the device (self.grad_norm) -- no host sync per iteration
        self.clip_grad_l2norm, self.clip_grad_val = clip_grad_l2norm, clip_grad_val
        self.grad_norm = None
        dev = next(model.parameters()).device
        self.use_graph = (dev.type == "cuda") if use_graph is None else bool(use_graph)
        self.warmup = warmup
        self._graph = None
        self._static_in = None
        self._static_labels = None
        self._loss = None
        self._logits = None
        self._calls = 0
        from .optim import FlatOptimizer
        self._flat = isinstance(optimizer, FlatOptimizer)
        if self._flat:
            # the fused update owns unscale / clipping / loss scale (csrc/sf_optim.h): arguments given HERE are forwarded into
            # it when it has none of its own, and a disagreement is an error -- never silently dropped
            for name, attr in (("clip_grad_val", "clip_val"), ("clip_grad_l2norm", "clip_norm")):
                want = float(getattr(self, name) or 0.0)
                have = float(getattr(optimizer, attr))
                if want and not have and not (attr == "clip_norm" and optimizer.clip_val):
                    setattr(optimizer, attr, want)
                elif want and have and abs(want - have) > 1e-12 * max(want, have):
                    raise ValueError("TrainStep(%s=%g) disagrees with the FlatOptimizer's %g" % (name, want, have))
            if optimizer.clip_val:
                optimizer.clip_norm = 0.0       # tools/train_net.py:154-163: the value clip takes precedence
            if self.loss_scale != 1.0 and self.loss_scale != optimizer.init_loss_scale:
                raise ValueError("with a FlatOptimizer the loss scale lives in the optimizer (FlatOptimizer(loss_scale=%g)): "
                                 "TrainStep(loss_scale=%g) would be ignored" % (optimizer.init_loss_scale, self.loss_scale))
        # backward in per-stage segments (all-reduce of a finished stage overlaps the backward of the earlier ones):
        # default on whenever gradients are actually exchanged
        self.segmented = bool(reducer.collectives) if segmented is None else bool(segmented)
        self._bwd_graphs, self._seg_params = [], []
        self.overlap_log = []           # per iteration: collectives in flight when the LAST backward segment starts
        import collections
        self.track_stats, self.stats_depth = track_stats, 8
        self._stats = collections.deque()
        self._last_labels = None
        self._pack_plan, self._pack_recorded = None, False

    # ------------------------------------------------------------------------------------------------
    def _pack_weights(self):
        """All fp16 weight operands in one launch (engine.WeightPackPlan) -- eagerly, or as the first node of the captured
        forward graph.  The plan is built from the layers that packed during the first iteration."""
        plan = self._pack_plan
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        if plan is not None and not capturing and not plan.valid():
            plan.release()
            plan = self._pack_plan = None
            self._pack_recorded = False
        if plan is not None:
            plan.run()
        elif engine.PACK_PLAN and not self._pack_recorded and engine.PACK_RECORD is None and not capturing:
            engine.PACK_RECORD = []

    def _pack_recorded_done(self):
        if engine.PACK_RECORD is not None and not self._pack_recorded:
            record, engine.PACK_RECORD = engine.PACK_RECORD, None
            self._pack_recorded = True
            if record:
                self._pack_plan = engine.WeightPackPlan(record)

    def _fwd_bwd(self, inputs, labels):
        self.reducer.zero_grad()
        self._pack_weights()
        logits = self.model(inputs)
        loss = self.loss_fn(logits.float(), labels)
        # FlatOptimizer: the (dynamic) loss scale is a device scalar -- a captured graph reads its current value at replay
        scale = self.optimizer.loss_scale if self._flat else self.loss_scale
        (loss * scale).backward()
        engine.join_side_streams()      # weight gradients forked to the side stream (engine.WGRAD_STREAM)
        return logits, loss

    # ---- segmented iteration --------------------------------------------------------------------------------------
    def _forward_segmented(self, inputs, labels):
        """zero_grad + forward + loss with the stage boundaries cut (engine.cut -> detached leaves); returns the pieces
        _backward_segment() needs."""
        self.reducer.zero_grad()
        self._pack_weights()
        rec = engine._Segments()
        engine.SEGMENTS = rec
        try:
            logits = self.model(inputs)
        finally:
            engine.SEGMENTS = None
        loss = self.loss_fn(logits.float(), labels)
        scale = self.optimizer.loss_scale if self._flat else self.loss_scale
        return logits, loss, loss * scale, rec.cuts

    @staticmethod
    def _backward_segment(k, nseg, scaled_loss, cuts):
        """Segment k of the backward pass, k = nseg-1 (head side) ... 0 (input side)."""
        engine.CUT_BACKWARD = True
        try:
            if k == nseg - 1:
                engine._cut_bn_tags.clear()
                torch.autograd.backward([scaled_loss])
            else:
                origs, leaves = cuts[k]
                pairs = [(o, l.grad) for o, l in zip(origs, leaves) if l is not o and l.grad is not None]
                for _, g in pairs:
                    engine.retag_cut_grad(g)
                engine._cut_bn_tags.clear()     # tags of gradients that did not end at a cut describe nothing any more
                if pairs:
                    torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])
        finally:
            engine.CUT_BACKWARD = False
        engine.join_side_streams()      # a backward segment (and its graph) ends with every forked stream joined

    def _iteration_segmented_eager(self, inputs, labels, record=False):
        logits, loss, scaled, cuts = self._forward_segmented(inputs, labels)
        nseg = len(cuts) + 1
        if record or not self._seg_params:
            self._seg_params = [[] for _ in range(nseg)]
        for k in range(nseg - 1, -1, -1):
            if k == 0:
                self.overlap_log.append(len(self.reducer._handles))
            seen = []
            listener = engine.add_grad_ready_listener(lambda ps, seen=seen: seen.extend(ps))
            try:
                self._backward_segment(k, nseg, scaled, cuts)
            finally:
                engine.remove_grad_ready_listener(listener)
            if record or not self._seg_params[k]:
                self._seg_params[k] = seen
        return logits, loss

    def _finish(self):
        if self._flat:
            # one norm pass + one control launch + one fused update over the flat buffers (csrc/sf_optim.h): unscale, inf /
            # NaN check with skip-step, GradScaler update, clipping and the optimizer arithmetic -- no host sync
            self.optimizer.finish_and_step()       # the norm pass runs bucket by bucket under the remaining collectives
            self.grad_norm = self.optimizer.grad_norm
            self._queue_stats()
            return
        self.reducer.finish(loss_scale=self.loss_scale)
        if self.clip_grad_val:
            self.reducer.flat.clamp_(-float(self.clip_grad_val), float(self.clip_grad_val))
            self.grad_norm = self.reducer.grad_norm()
        else:
            self.grad_norm = self.reducer.grad_norm()
            if self.clip_grad_l2norm:       # torch.nn.utils.clip_grad_norm_: g *= min(1, max_norm / (norm + 1e-6))
                coef = torch.clamp(float(self.clip_grad_l2norm) / (self.grad_norm + 1e-6), max=1.0)
                self.reducer.flat.mul_(coef)
        if self.optimizer is not None:
            self.optimizer.step()
        self._queue_stats()

    # ------------------------------------------------------------------------------------------------
    def _queue_stats(self):
        """[loss, grad_norm, top1_err, top5_err] of this iteration as ONE device tensor, mean-all-reduced asynchronously
        (tools/train_net.py:225-241 does four blocking .item() calls and a du.all_reduce per iteration); read it later
        with pop_stats() -- by then the values have long arrived, so the read does not stall the launch queue."""
        if not self.track_stats:
            return
        logits, labels = self._logits, self._last_labels
        stats = torch.zeros(4, dtype=torch.float32, device=logits.device)
        stats[0] = self._loss.float()
        stats[1] = self.grad_norm.float() if torch.is_tensor(self.grad_norm) else float(self.grad_norm or 0.0)
        if labels is not None and labels.dim() == 1 and logits.dim() == 2:
            k = min(5, logits.shape[1])
            top = logits.float().topk(k, dim=1).indices
            hit = top.eq(labels.view(-1, 1))
            stats[2] = 100.0 * (1.0 - hit[:, :1].any(1).float().mean())
            stats[3] = 100.0 * (1.0 - hit.any(1).float().mean())
        handle = None
        import torch.distributed as dist
        if self.reducer.world > 1 and dist.is_available() and dist.is_initialized():
            stats /= self.reducer.world
            handle = dist.all_reduce(stats, group=self.reducer.group, async_op=True)
        self._stats.append((stats, handle))
        while len(self._stats) > self.stats_depth:
            self._stats.popleft()

    def pop_stats(self):
        """Oldest queued [loss, grad_norm, top1_err, top5_err] as floats (None when nothing is queued)."""
        if not self._stats:
            return None
        stats, handle = self._stats.popleft()
        if handle is not None:
            handle.wait()
        return [float(v) for v in stats.cpu()]

    @staticmethod
    def _static_clone(x):
        """The graph's own copy of an input: same strides (clone keeps a dense permuted layout), and the W-pair tag of a clip
        packed by data.pack_pathways_u8 travels with it -- engine.StemConvUnit must read the copy exactly as it would read the
        original, and a loader may later write packed clips straight into it (static_inputs())."""
        c = x.clone()
        if getattr(x, "_sf_wpairs", False):
            c._sf_wpairs = True
        return c

    def _capture(self, inputs, labels):
        self._is_list = isinstance(inputs, (list, tuple))
        self._static_in = [self._static_clone(x) for x in inputs] if self._is_list else self._static_clone(inputs)
        self._static_labels = labels.clone()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        engine.FORCE_WEIGHT_PREP = True
        self.reducer.capturing = True
        # with a process group alive, its watchdog thread polls events while this thread captures: restrict the capture
        # checks to the capturing thread then (the default "global" mode would fail the capture on such a query)
        import torch.distributed as dist
        mode = "thread_local" if dist.is_available() and dist.is_initialized() else "global"
        try:
            if not self.segmented:
                with torch.cuda.graph(g, capture_error_mode=mode):
                    logits, loss = self._fwd_bwd(self._static_in, self._static_labels)
            else:
                with torch.cuda.graph(g, capture_error_mode=mode):
                    logits, loss, scaled, cuts = self._forward_segmented(self._static_in, self._static_labels)
                nseg = len(cuts) + 1
                self._bwd_graphs = [None] * nseg
                self._seg_params = [[] for _ in range(nseg)]
                pool = g.pool()
                for k in range(nseg - 1, -1, -1):          # one graph per backward segment, same memory pool, replay order
                    gk = torch.cuda.CUDAGraph()
                    seen = []
                    listener = engine.add_grad_ready_listener(lambda ps, seen=seen: seen.extend(ps))
                    try:
                        with torch.cuda.graph(gk, pool=pool, capture_error_mode=mode):
                            self._backward_segment(k, nseg, scaled, cuts)
                    finally:
                        engine.remove_grad_ready_listener(listener)
                    self._bwd_graphs[k] = gk
                    self._seg_params[k] = seen
                # parameters that autograd accumulates itself (the torch head) finish with the head-side segment
                self._seg_params[nseg - 1] = self._seg_params[nseg - 1] + [p for p in self.reducer._hooked]
        finally:
            engine.FORCE_WEIGHT_PREP = False
            self.reducer.capturing = False
        self._graph, self._logits, self._loss = g, logits.detach(), loss.detach()

    def __call__(self, inputs, labels):
        """Runs one iteration; returns the (unscaled) loss of this iteration as a 0-d device tensor.  The tensor (and
        ``.logits``) are copies: the captured graph's own output buffers are overwritten by the next replay."""
        self._calls += 1
        self._last_labels = labels
        if not self.use_graph or self._calls <= self.warmup:
            if self.segmented:
                logits, loss = self._iteration_segmented_eager(inputs, labels)
            else:
                logits, loss = self._fwd_bwd(inputs, labels)
            self._logits, self._loss = logits.detach(), loss.detach()
            self._pack_recorded_done()
            self._finish()
            return self._loss
        if self._graph is None:
            self._capture(inputs, labels)
        else:
            pairs = zip(self._static_in, inputs) if self._is_list else [(self._static_in, inputs)]
            for dst, src in pairs:
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src, non_blocking=True)
            if self._static_labels.data_ptr() != labels.data_ptr():
                self._static_labels.copy_(labels, non_blocking=True)
        self.reducer.begin_replay()
        self._graph.replay()
        if self.segmented:
            n = len(self._bwd_graphs)
            for k in range(n - 1, -1, -1):
                if k == 0:
                    self.overlap_log.append(len(self.reducer._handles))
                self._bwd_graphs[k].replay()
                self.reducer._on_ready(self._seg_params[k])     # finished buckets: all-reduce overlaps the next segment
        self._finish()
        return self._loss.clone() if self._graph is not None else self._loss

    def static_inputs(self):
        """(inputs, labels) buffers the captured graph reads (None before the capture).  A loader that writes the next batch
        straight into them saves the per-iteration device-to-device copy __call__ otherwise makes from the tensors it is handed
        (0.28 ms for a 32-clip SlowFast batch).  For a step captured on clips packed by data.pack_pathways_u8 these ARE packed
        (tagged) buffers: ``pack_pathways_u8(frames, cfg, out=step.static_inputs()[0])`` then ``step(*step.static_inputs())``
        (tests/test_step.py::test_packed_loader_writes_static_inputs).  A step captured on fp32 NCTHW clips has fp32 buffers,
        which that loader rejects."""
        if self._graph is None:
            return None
        return (list(self._static_in) if self._is_list else self._static_in), self._static_labels

    @property
    def logits(self):
        return self._logits.clone() if self._graph is not None and self._logits is not None else self._logits
