"""Which residual-stream storage puts MViTv2-S's full-size logits under 1e-3?  

CPU only: the pinned oracle graph (oracle/mvit_ref.py) in fp32 arithmetic with every stored tensor rounded to the 16-bit storage
type (video_ref.fp16_storage_model), the residual stream stored per ``mvit_ref.RESID_POLICY``.  Same case as
tests/model_checks.check_full_size("MVITv2_S_16x4"): batch 2, seed 99.  Prints logits relative L2 against the fp32 run.
"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import mvit_ref, video_ref


def main():
    import slowfast_amd as sa
    seed, batch = 99, 2
    cfg = sa.get_preset("MVITv2_S_16x4", ["NUM_GPUS", 0, "MODEL.DROPOUT_RATE", 0.0, "MVIT.DROPPATH_RATE", 0.0, "MIXUP.ENABLE", False])
    model = sa.MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)
    sd = mvit_ref.randomize_state({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed)
    # (tests/test_model_gpu.py FULL_SIZE["MVITv2_S_16x4"]: head_abs=False, no gamma damping)
    inputs, _ = video_ref.synthetic_batch(cfg, batch, seed + 1)
    with torch.no_grad():
        ref = mvit_ref.mvit_forward(sd, cfg, list(inputs), training=True)
    variants = [("storage model, everything rounded", None),
                ("cls row fp32", mvit_ref.engine_resid_policy(True, 99)),
                ("blocks 14-15 residual fp32", mvit_ref.engine_resid_policy(False, 14)),
                ("cls row fp32 + blocks 14-15 residual fp32", mvit_ref.engine_resid_policy(True, 14)),
                ("cls row fp32 + blocks 12-15 residual fp32", mvit_ref.engine_resid_policy(True, 12)),
                ("cls row fp32 + blocks 8-15 residual fp32", mvit_ref.engine_resid_policy(True, 8)),
                ("whole residual stream fp32", mvit_ref.engine_resid_policy(True, 0))]
    for name, pol in variants:
        mvit_ref.RESID_POLICY = pol
        with torch.no_grad(), video_ref.fp16_storage_model():
            z = mvit_ref.mvit_forward(sd, cfg, list(inputs), training=True)
        mvit_ref.RESID_POLICY = None
        print(f"| {name} | {float((z - ref).norm() / ref.norm()):.3e} | {float((z - ref).abs().max() / ref.abs().max()):.3e} |", flush=True)


if __name__ == "__main__":
    main()
