"""CPU restatement (test infrastructure) of the reference's input-side tensor preparation for the hot path:

* ``tensor_normalize``      slowfast/datasets/utils.py:278-297   (uint8 -> float /255, - mean, / std)
* THWC -> CTHW permute       slowfast/datasets/kinetics.py:375-408
* ``pack_pathway_output``   slowfast/datasets/utils.py:78-111    (channel reversal, Slow pathway = temporal index_select)

Checked against the reference's own functions by tests/test_oracle.py when /root/reference is present (they import
cleanly through oracle/refshim.py); only tests may import this module."""
import torch


def pack_pathways(frames_u8, cfg):
    """frames_u8: uint8 (N, T, H, W, 3) -> list of float32 (N, 3, T', H, W) pathway clips, as the reference loader +
    collate would hand them to the model."""
    out_clips = []
    for n in range(frames_u8.shape[0]):
        t = frames_u8[n].float() / 255.0
        t = t - torch.tensor(cfg.DATA.MEAN)
        t = t / torch.tensor(cfg.DATA.STD)
        t = t.permute(3, 0, 1, 2)                              # T H W C -> C T H W
        if cfg.DATA.REVERSE_INPUT_CHANNEL:
            t = t[[2, 1, 0], :, :, :]
        if cfg.MODEL.ARCH in cfg.MODEL.SINGLE_PATHWAY_ARCH:
            lst = [t]
        elif cfg.MODEL.ARCH in cfg.MODEL.MULTI_PATHWAY_ARCH:
            idx = torch.linspace(0, t.shape[1] - 1, t.shape[1] // cfg.SLOWFAST.ALPHA).long()
            lst = [torch.index_select(t, 1, idx), t]
        else:
            raise NotImplementedError(cfg.MODEL.ARCH)
        out_clips.append(lst)
    return [torch.stack([c[p] for c in out_clips], 0) for p in range(len(out_clips[0]))]
