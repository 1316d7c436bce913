"""Classification head (adjacent to the hot path: < 1 MMAC/clip).

``ResNetBasicHead`` keeps the reference's constructor and state_dict (slowfast/models/head_helper.py:198-350):
per-pathway average pool -> concat -> dropout -> Linear; raw logits in training, activation + spatial
mean in eval.  It runs on torch fp32 ops over the (tiny) res5 outputs."""
import torch
import torch.nn as nn

from . import engine, ops
import os

from .lib import get_lib

# the training-crop average pool of ResNetBasicHead on sf_tmean_* (SF_HEAD_FUSED_MEAN=0: torch fp32 ops, the round 1-5 path)
FUSED_MEAN = os.environ.get("SF_HEAD_FUSED_MEAN", "1") != "0"


class _GlobalMeanFn(torch.autograd.Function):
    """Mean over the whole (T, H, W) extent of a channels-last 16-bit activation -> (N, C, 1, 1, 1) fp32, on the temporal-mean
    kernels of the RoI head (sf_tmean_fwd / sf_tmean_bwd) instead of `x.float().mean()`: the fp32 copy of the res5 output (and its
    mirror image in backward: an fp32 broadcast that is then cast back) is never written.  The rows of a clip are read as
    [k][rows / k] so that every thread sums k = 8 rows; the [N, rows / k, C] fp32 partial means are averaged by torch."""
    K = 8

    @staticmethod
    def eligible(x):
        return x.dim() == 5 and ops.is_cl(x) and x.shape[1] % 8 == 0 and (x.shape[2] * x.shape[3] * x.shape[4]) % _GlobalMeanFn.K == 0

    @staticmethod
    def forward(ctx, x):
        N, C, T, H, W = x.shape
        k, rows = _GlobalMeanFn.K, T * H * W
        part = torch.empty((N * (rows // k), C), dtype=torch.float32, device=x.device)
        get_lib().call("sf_tmean_fwd", N, k, rows // k, C, x.data_ptr(), ops.cl_ld(x), part.data_ptr(), ops._stream(x),
                       work=dict(bytes=2.0 * x.numel()))
        ctx.geom = (N, C, T, H, W)
        return part.view(N, rows // k, C).mean(1).view(N, C, 1, 1, 1)

    @staticmethod
    def backward(ctx, dout):
        N, C, T, H, W = ctx.geom
        dout = dout.reshape(N, C).float().contiguous()
        dx = ops.cl_empty((N, C, T, H, W), dout.device)
        get_lib().call("sf_tmean_bwd", N, T * H * W, 1, C, dout.data_ptr(), dx.data_ptr(), ops.cl_ld(dx), ops._stream(dout),
                       work=dict(bytes=2.0 * dx.numel()))
        return dx


class ResNetBasicHead(nn.Module):
    def __init__(self, dim_in, num_classes, pool_size, dropout_rate=0.0, act_func="softmax", detach_final_fc=False,
                 cfg=None):
        super().__init__()
        assert len(pool_size) == len(dim_in), "pathway dimensions are not consistent."
        self.num_pathways = len(pool_size)
        self.detach_final_fc = detach_final_fc
        self.cfg = cfg
        for i, ps in enumerate(pool_size):
            pool = nn.AdaptiveAvgPool3d((1, 1, 1)) if ps is None else nn.AvgPool3d(tuple(ps), stride=1)
            self.add_module(f"pathway{i}_avgpool", pool)
        if dropout_rate > 0.0:
            self.dropout = nn.Dropout(dropout_rate)
        mlp_layers = cfg.CONTRASTIVE.NUM_MLP_LAYERS if cfg is not None else 1
        assert mlp_layers == 1, "MLP projection heads belong to the self-supervised models (out of scope)"
        self.projection = nn.Linear(sum(dim_in), num_classes, bias=True)
        if act_func == "softmax":
            self.act = nn.Softmax(dim=4)
        elif act_func == "sigmoid":
            self.act = nn.Sigmoid()
        elif act_func == "none":
            self.act = None
        else:
            raise NotImplementedError(f"{act_func} is not supported as an activationfunction.")

    def forward(self, inputs):
        assert len(inputs) == self.num_pathways, f"Input tensor does not contain {self.num_pathways} pathway"
        pooled = []
        for i, x in enumerate(inputs):
            pool = getattr(self, f"pathway{i}_avgpool")
            full = isinstance(pool, nn.AdaptiveAvgPool3d) or tuple(pool.kernel_size) == tuple(x.shape[2:])
            if full and _GlobalMeanFn.eligible(x) and FUSED_MEAN:
                pooled.append(_GlobalMeanFn.apply(x))
            elif full and x.dim() == 5 and x.stride(1) == 1:
                # the pool window is the whole (T,H,W) extent (training crop): mean over the channels-last rows
                N, C = x.shape[:2]
                rows = x.permute(0, 2, 3, 4, 1).reshape(N, -1, C)
                pooled.append(rows.float().mean(1).view(N, C, 1, 1, 1))
            else:
                pooled.append(pool(x.float().contiguous()))
        x = torch.cat(pooled, 1).permute(0, 2, 3, 4, 1)
        if hasattr(self, "dropout"):
            x = self.dropout(x)
        if self.detach_final_fc:
            x = x.detach()
        x = self.projection(x)
        if not self.training:
            if self.act is not None:
                x = self.act(x)
            x = x.mean([1, 2, 3])
        return x.view(x.shape[0], -1)


class _RoiPoolFn(torch.autograd.Function):
    """AvgPool3d([T,1,1]) -> ROIAlign(res, 1/scale_factor, sampling_ratio 0, aligned) -> MaxPool2d(res) of one pathway
    on the libsfamd kernels (sf_tmean_*, sf_roi_align_max_*): x channels-last fp16 (N,C,T,H,W) -> (R, C) fp32."""

    @staticmethod
    def forward(ctx, x, rois, res, scale, aligned):
        x = ops.to_cl(x)
        lib = get_lib()
        N, C, T, H, W = x.shape
        s = ops._stream(x)
        m = torch.empty((N * H * W, C), dtype=torch.float32, device=x.device)
        lib.call("sf_tmean_fwd", N, T, H * W, C, x.data_ptr(), ops.cl_ld(x), m.data_ptr(), s,
                 work=dict(bytes=2.0 * x.numel()))
        rois = rois.detach().to(device=x.device, dtype=torch.float32).contiguous()
        R = rois.shape[0]
        out = torch.empty((R, C), dtype=torch.float32, device=x.device)
        arg = torch.empty((R, C), dtype=torch.uint8, device=x.device)
        lib.call("sf_roi_align_max_fwd", R, N, H, W, C, res, float(scale), int(bool(aligned)), m.data_ptr(),
                 rois.data_ptr(), out.data_ptr(), C, 0, arg.data_ptr(), s, work=dict(bytes=4.0 * R * C * res * res))
        ctx.geom = (N, C, T, H, W, res, float(scale), int(bool(aligned)))
        ctx.save_for_backward(rois, arg)
        ctx.arg_bins = arg            # test hook: the arg-max bin of every (roi, channel), see ResNetRoIHead.forward
        return out

    @staticmethod
    def backward(ctx, dout):
        rois, arg = ctx.saved_tensors
        N, C, T, H, W, res, scale, aligned = ctx.geom
        lib = get_lib()
        dout = dout.contiguous().float()
        s = ops._stream(dout)
        dm = torch.zeros((N * H * W, C), dtype=torch.float32, device=dout.device)
        lib.call("sf_roi_align_max_bwd", rois.shape[0], N, H, W, C, res, scale, aligned, rois.data_ptr(), dout.data_ptr(),
                 C, 0, arg.data_ptr(), dm.data_ptr(), s)
        dx = ops.cl_empty((N, C, T, H, W), dout.device)
        lib.call("sf_tmean_bwd", N, T, H * W, C, dm.data_ptr(), dx.data_ptr(), ops.cl_ld(dx), s,
                 work=dict(bytes=2.0 * dx.numel()))
        return dx, None, None, None, None


class ResNetRoIHead(nn.Module):
    """ResNe(X)t RoI head with the reference's constructor and state_dict (slowfast/models/head_helper.py:20-144):
    per pathway temporal average pool -> ROIAlign -> spatial max pool, concat, dropout, Linear, activation (applied in
    training as well, as the reference does).  forward(inputs, bboxes) with bboxes (R, 5) = [batch index, x1, y1, x2, y2]."""

    def __init__(self, dim_in, num_classes, pool_size, resolution, scale_factor, dropout_rate=0.0, act_func="softmax",
                 aligned=True, detach_final_fc=False):
        super().__init__()
        assert len({len(pool_size), len(dim_in)}) == 1, "pathway dimensions are not consistent."
        self.num_pathways = len(pool_size)
        self.detach_final_fc = detach_final_fc
        self.pool_size, self.resolution, self.scale_factor, self.aligned = pool_size, resolution, scale_factor, aligned
        for p in range(self.num_pathways):       # parameter-free children kept for module-tree parity
            self.add_module(f"s{p}_tpool", nn.AvgPool3d([pool_size[p][0], 1, 1], stride=1))
            self.add_module(f"s{p}_spool", nn.MaxPool2d(resolution[p], stride=1))
        if dropout_rate > 0.0:
            self.dropout = nn.Dropout(dropout_rate)
        self.projection = nn.Linear(sum(dim_in), num_classes, bias=True)
        if act_func == "softmax":
            self.act = nn.Softmax(dim=1)
        elif act_func == "sigmoid":
            self.act = nn.Sigmoid()
        else:
            raise NotImplementedError(f"{act_func} is not supported as an activationfunction.")

    def forward(self, inputs, bboxes):
        assert len(inputs) == self.num_pathways, f"Input tensor does not contain {self.num_pathways} pathway"
        assert bboxes is not None and bboxes.dim() == 2 and bboxes.shape[1] == 5, "bboxes: (R, 5) = [batch idx, x1, y1, x2, y2]"
        pooled = []
        for p, x in enumerate(inputs):
            assert x.shape[2] == self.pool_size[p][0], "the temporal pool must cover the pathway's frames"
            res = self.resolution[p]
            assert res[0] == res[1]
            pooled.append(_RoiPoolFn.apply(x, bboxes, int(res[0]), 1.0 / self.scale_factor[p], self.aligned))
            if engine.CAPTURE is not None and pooled[-1].grad_fn is not None:
                engine.CAPTURE.append({"kind": "roi_pool", "mod": self, "pathway": p, "argmax": pooled[-1].grad_fn.arg_bins})
        x = torch.cat(pooled, 1)
        if hasattr(self, "dropout"):
            x = self.dropout(x)
        if self.detach_final_fc:
            x = x.detach()
        return self.act(self.projection(x))
