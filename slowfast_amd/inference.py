"""Eval / multi-view test path (reference: tools/test_net.py:25-151 ``perform_test``).

Two pieces:

``fuse_for_inference(model)`` -- inference fusion of the ResNet-family drop-ins.  In eval mode ``nn.BatchNorm3d``
is a fixed per-channel affine map, so every ``conv -> BN [-> ReLU]`` unit of the reference graph
(stem_helper.py:196-201, resnet_helper.py:377-392 and 512-521, video_model_builder.py:162-169,
nonlocal_helper.py:139-144) collapses into ONE implicit-GEMM launch: BatchNorm folded into the packed fp16 weights
and a bias, ReLU and the residual addition in the epilogue (``sf_conv_fwd_fused``).  A bottleneck block is then 3
launches (4 with a projection shortcut) instead of 4-5 convolutions + 4-5 statistics finalisations + one
elementwise pass, no activation is ever re-normalised on load, and all 1x1x1 operands take the direct-to-LDS path.
X3D blocks fold their two 1x1x1 convolutions and the shortcut the same way (two elementwise passes fewer per block);
the depthwise 3x3x3 -> BN -> SE -> Swish middle and the X3D stem keep the running-statistics schedule.  MViT has no
BatchNorm to fold.

``TestStep`` -- one iteration of ``perform_test``: eval forward (captured once into a HIP graph and replayed, like
``step.TrainStep``), all-gather of (preds, labels, video_idx) across ranks (``du.all_gather``,
slowfast/utils/distributed.py:26-43) and the multi-view ensemble of ``TestMeter.update_stats``
(slowfast/utils/meters.py:305-336: video = clip_id // num_clips, "sum" | "max") kept ON THE DEVICE, which removes the
three ``.cpu()`` synchronisations per iteration of the reference loop (test_net.py:113-117).
"""
import torch
import torch.distributed as dist


def fuse_for_inference(model):
    """Switch ``model`` to eval mode and fold BatchNorm into every fusable unit.  The folded operands are snapshots of
    the current parameters / running statistics: call again after loading a checkpoint or training further.
    ``model.train()`` DROPS the fused state (the snapshots would be stale after the next optimizer step); a later
    ``model.eval()`` then runs the un-fused running-statistics schedule until this function is called again."""
    model.eval()
    n = 0
    for m in model.modules():
        hook = getattr(m, "_sf_fold", None)
        if hook is not None and hook() is not False:
            m.__dict__["_sf_infer"] = True
            n += 1
    model.__dict__["_sf_fused_modules"] = n
    # the folded operands are snapshots: going back to training invalidates them, so the usual
    # train -> eval-per-epoch loop can never evaluate with stale weights / statistics -- it falls back to the
    # running-statistics schedule until fuse_for_inference() is called again
    if "_sf_train_guard" not in model.__dict__:
        plain_train = model.train

        def train(mode=True):
            if mode:
                unfuse(model)
            return plain_train(mode)
        model.__dict__["_sf_train_guard"] = True
        model.train = train
    return model


def unfuse(model):
    for m in model.modules():
        m.__dict__.pop("_sf_infer", None)
    model.__dict__.pop("_sf_fused_modules", None)
    return model


def all_gather_cat(tensors, group=None):
    """``du.all_gather`` (slowfast/utils/distributed.py:26-43): every tensor gathered from all ranks and concatenated
    along dim 0 (equal shapes on every rank, as the test loader guarantees with drop_last=False + DistributedSampler
    padding).  World size 1 returns the inputs."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(tensors)
    world = dist.get_world_size(group)
    out = []
    for t in tensors:
        t = t.contiguous()
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        out.append(torch.cat(parts, dim=0))
    return out


def all_gather_unaligned(tensors, group=None):
    """The tensor case of ``du.all_gather_unaligned`` (slowfast/utils/distributed.py:225-258), used by the detection branch
    of perform_test (test_net.py:76-79) where every rank holds a different number of boxes: sizes are exchanged first,
    rows padded to the largest count, gathered, trimmed and concatenated along dim 0.  World size 1 returns the inputs."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(tensors)
    world = dist.get_world_size(group)
    out = []
    for t in tensors:
        t = t.contiguous()
        n = torch.tensor([t.shape[0]], dtype=torch.long, device=t.device)
        sizes = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(sizes, n, group=group)
        sizes = [int(v) for v in sizes]
        pad = torch.zeros((max(sizes),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        out.append(torch.cat([p[:k] for p, k in zip(parts, sizes)], dim=0))
    return out


class TestStep:
    """perform_test's loop body for the classification path.

    ``step(inputs, labels, video_idx)`` -> gathered (preds, labels, video_idx); the per-video ensemble lives in
    ``video_preds`` [num_videos, num_cls] / ``video_labels`` / ``clip_count`` on the model's device and
    ``finalize()`` returns top-k accuracies like TestMeter.finalize_metrics (meters.py:366-400)."""

    def __init__(self, model, num_videos, num_clips, num_cls, ensemble_method="sum", multi_label=False, use_graph=None,
                 warmup=1, process_group=None):
        if ensemble_method not in ("sum", "max"):
            raise NotImplementedError(f"Ensemble Method {ensemble_method} is not supported")
        self.model = model.eval()
        self.group = process_group
        self.num_clips, self.ensemble_method, self.multi_label = num_clips, ensemble_method, multi_label
        dev = next(model.parameters()).device
        self.device = dev
        self.video_preds = torch.zeros((num_videos, num_cls), dtype=torch.float32, device=dev)
        if multi_label:
            self.video_preds -= 1e10
        self.video_labels = (torch.zeros((num_videos, num_cls), device=dev) if multi_label
                             else torch.zeros((num_videos,), dtype=torch.long, device=dev))
        self.clip_count = torch.zeros((num_videos,), dtype=torch.long, device=dev)
        self.use_graph = (dev.type == "cuda") if use_graph is None else bool(use_graph)
        self.warmup = warmup
        self._graph, self._static_in, self._preds, self._calls = None, None, None, 0

    def _forward(self, inputs):
        with torch.no_grad():
            return self.model(inputs).float()

    def _capture(self, inputs):
        self._static_in = [x.clone() for x in inputs]
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        mode = "thread_local" if dist.is_available() and dist.is_initialized() else "global"
        with torch.cuda.graph(g, capture_error_mode=mode):
            preds = self._forward(self._static_in)
        self._graph, self._preds = g, preds

    def predict(self, inputs):
        """Softmax (or sigmoid) scores of this rank's clips, (B, num_cls) fp32 on the device."""
        self._calls += 1
        inputs = list(inputs)
        if not self.use_graph or self._calls <= self.warmup:
            return self._forward(inputs)
        if self._graph is None or any(tuple(a.shape) != tuple(b.shape) for a, b in zip(self._static_in, inputs)):
            self._capture(inputs)            # a new clip shape (last, smaller batch) re-captures
        else:
            for dst, src in zip(self._static_in, inputs):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src, non_blocking=True)
        self._graph.replay()
        # the graph's output buffer is overwritten by the next replay: hand out a copy (B x num_cls floats), so callers may
        # collect the scores of several iterations (e.g. to feed a reference TestMeter)
        return self._preds.clone()

    def update(self, preds, labels, clip_ids):
        """TestMeter.update_stats (meters.py:305-336) without leaving the device: clips of one video may arrive in any
        order and on any rank; 'sum' adds the scores, 'max' keeps the element-wise maximum."""
        vid = torch.div(clip_ids.to(self.device).long(), self.num_clips, rounding_mode="floor")
        preds = preds.to(self.device, torch.float32)
        labels = labels.to(self.device)
        if self.ensemble_method == "sum":
            self.video_preds.index_add_(0, vid, preds)
        else:
            self.video_preds.index_reduce_(0, vid, preds, "amax", include_self=True)
        self.video_labels[vid] = labels.to(self.video_labels.dtype)
        self.clip_count.index_add_(0, vid, torch.ones_like(vid))

    def step(self, inputs, labels, video_idx):
        preds = self.predict(inputs)
        preds, labels, video_idx = all_gather_cat([preds, labels.to(self.device), video_idx.to(self.device)], self.group)
        self.update(preds, labels, video_idx)
        return preds, labels, video_idx

    def step_detection(self, inputs, boxes, ori_boxes, metadata):
        """perform_test's detection branch (test_net.py:68-86): ``model(inputs, boxes)`` on the (R, 5) boxes of this rank's
        clips, then (preds, ori_boxes, metadata) gathered from all ranks (different R per rank) for the AVA meter.  Runs
        eagerly: the number of boxes changes from batch to batch, so there is no static graph to replay."""
        with torch.no_grad():
            preds = self.model(list(inputs), boxes.to(self.device)).float()
        return all_gather_unaligned([preds, ori_boxes.to(self.device), metadata.to(self.device)], self.group)

    def finalize(self, ks=(1, 5)):
        """{'top1_acc': .., 'top5_acc': ..} in percent over the ensembled videos (meters.py:366-400, single-label)."""
        assert not self.multi_label, "mAP of the multi-label datasets is computed by the dataset-side meters"
        complete = bool((self.clip_count == self.num_clips).all())
        kmax = min(max(ks), self.video_preds.shape[1])
        top = self.video_preds.topk(kmax, dim=1).indices
        hit = top == self.video_labels.view(-1, 1)
        stats = {f"top{k}_acc": float(hit[:, :min(k, kmax)].any(1).float().mean() * 100.0) for k in ks}
        stats["all_clips_seen"] = complete
        return stats
