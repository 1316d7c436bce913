"""slowfast_amd -- MI355X-native forward/backward engine for the PySlowFast video backbones.

Hand-written gfx950 HIP kernels behind a C ABI (include/sfamd.h, csrc/), wrapped as torch.nn.Module
drop-ins with the reference's constructor signatures, cfg keys and state_dict names.
"""
from .config import CfgNode, get_cfg, get_preset  # noqa: F401
from .registry import MODEL_REGISTRY, build_model  # noqa: F401

__version__ = "0.1.0"
from . import video_models  # noqa: F401,E402  (registers SlowFast / ResNet in MODEL_REGISTRY)
from . import mvit  # noqa: F401,E402  (registers MViT)
from . import x3d  # noqa: F401,E402  (registers X3D, the x3d_stem and x3d_transform factories)
from .data import pack_pathways_u8  # noqa: F401,E402  (uint8 frames -> stem operand layout)
