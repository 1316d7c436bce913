"""Input side of the hot path: decoded uint8 frames -> the stems' operand layout in one kernel per
pathway.

The reference normalises on the host in fp32 (``utils.tensor_normalize``, slowfast/datasets/utils.py:278-297), permutes
T,H,W,C -> C,T,H,W (datasets/kinetics.py:375-408), builds the pathway list (``pack_pathway_output``,
datasets/utils.py:78-111: Slow = ``index_select(frames, 1, linspace(0, T-1, T//ALPHA).long())``, optional BGR reversal)
and ships float32 clips to the GPU, where this engine would convert them to fp16 W-pair rows again.
``pack_pathways_u8`` takes the cropped uint8 clip batch (N, T, H, W, 3) on the device and writes, per pathway, the
normalised fp16 N,T,H,W,4 buffer the stems read directly (``engine.StemConvUnit`` skips its own conversion for tensors
produced here) -- a quarter of the bytes over PCIe and no fp32 clip in HBM."""
import torch

from . import lib as _sflib

from . import ops
from .lib import get_lib

_f16 = _sflib.act_dtype()        # fp16, or bf16 under SF_ACT_DTYPE=bf16 (lib.ACT_MODE)


def pathway_frame_indices(cfg, num_frames):
    """Source-frame indices of every pathway, as pack_pathway_output builds them (datasets/utils.py:89-105)."""
    if cfg.MODEL.ARCH in cfg.MODEL.SINGLE_PATHWAY_ARCH:
        return [None]
    if cfg.MODEL.ARCH in cfg.MODEL.MULTI_PATHWAY_ARCH:
        return [torch.linspace(0, num_frames - 1, num_frames // cfg.SLOWFAST.ALPHA).long(), None]
    raise NotImplementedError(f"Model arch {cfg.MODEL.ARCH} is not in "
                              f"{cfg.MODEL.SINGLE_PATHWAY_ARCH + cfg.MODEL.MULTI_PATHWAY_ARCH}")


def pack_pathways_u8(frames, cfg, out=None):
    """frames: uint8 (N, T, H, W, 3) device tensor (decoded, sampled, cropped).  Returns the model input list: one
    channels-last fp16 tensor per pathway in the W-pair view (N, 8, T', H, W/2), tagged so that the stems use it as is.
    ``out``: tensors of a previous call (e.g. the static input buffers of a captured step.TrainStep) to write into."""
    assert frames.dtype == torch.uint8 and frames.dim() == 5 and frames.shape[-1] == 3 and frames.shape[3] % 2 == 0
    frames = frames.contiguous()
    N, T, H, W, _ = frames.shape
    mean, std = [float(v) for v in cfg.DATA.MEAN], [float(v) for v in cfg.DATA.STD]
    dst, out = out, []
    for i, idx in enumerate(pathway_frame_indices(cfg, T)):
        Tout = T if idx is None else int(idx.numel())
        idx_dev = None if idx is None else idx.to(device=frames.device, dtype=torch.int32).contiguous()
        if dst is None:
            base = torch.empty((N, Tout, H, W // 2, 8), dtype=_f16, device=frames.device)
        else:
            base = dst[i].permute(0, 2, 3, 4, 1)
            assert tuple(base.shape) == (N, Tout, H, W // 2, 8) and base.is_contiguous() and base.dtype == _f16, \
                "out[i] must be a tensor a previous pack_pathways_u8 call returned for the same clip geometry"
        get_lib().call("sf_pack_clip_u8", frames.data_ptr(), N, T, H, W, ops._ptr(idx_dev), Tout, mean[0], mean[1], mean[2],
                       std[0], std[1], std[2], int(bool(cfg.DATA.REVERSE_INPUT_CHANNEL)), base.data_ptr(),
                       ops._stream(frames), work=dict(bytes=3.0 * N * Tout * H * W + 2.0 * base.numel()))
        x = base.permute(0, 4, 1, 2, 3)
        x._sf_wpairs = True                 # already the operand layout of engine.StemConvUnit
        out.append(x)
    return out
