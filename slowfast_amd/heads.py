"""Classification head (adjacent to the hot path: < 1 MMAC/clip).

``ResNetBasicHead`` keeps the reference's constructor and state_dict (slowfast/models/head_helper.py:198-350):
per-pathway average pool -> concat -> dropout -> Linear; raw logits in training, activation + spatial
mean in eval.  It runs on torch fp32 ops over the (tiny) res5 outputs."""
import torch
import torch.nn as nn


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
            if full and x.dim() == 5 and x.stride(1) == 1:
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
