"""Configuration node for the hot path.

A small yacs-style attribute tree with the reference's key names and default values for every key the
video-backbone path reads (slowfast/config/defaults.py: MODEL.* :393-441, RESNET.* :293-327,
SLOWFAST.* :633-648, NONLOCAL.* :363-385, BN.* :99-126, DATA.* :666-716, SOLVER.* :812-878).
It loads the reference's own YAML files unchanged (including string-encoded tuples such as
``PATCH_KERNEL: (3, 7, 7)``); keys outside the hot path are accepted and kept, not interpreted.
When the reference package is installed, its ``CfgNode`` can be passed to every constructor here
instead -- only attribute access is used.
"""
import ast
import copy

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def clone(self):
        return copy.deepcopy(self)

    @staticmethod
    def _coerce(value):
        if isinstance(value, str):
            try:
                return ast.literal_eval(value)
            except (ValueError, SyntaxError):
                return value
        return value

    def merge_from_dict(self, other):
        for k, v in other.items():
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], CfgNode):
                    self[k] = CfgNode()
                self[k].merge_from_dict(v)
            else:
                v = self._coerce(v)
                if isinstance(v, tuple) and isinstance(self.get(k), list):
                    v = list(v)
                self[k] = v

    def merge_from_file(self, path):
        with open(path) as f:
            self.merge_from_dict(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0, "opts must be KEY VALUE pairs"
        for key, value in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    node[p] = CfgNode()
                node = node[p]
            node[parts[-1]] = self._coerce(value)

    def dump(self):
        def plain(n):
            return {k: plain(v) if isinstance(v, CfgNode) else v for k, v in n.items()}
        return yaml.safe_dump(plain(self))


_DEFAULTS = {
    "BN": {"USE_PRECISE_STATS": False, "NUM_BATCHES_PRECISE": 200, "WEIGHT_DECAY": 0.0, "NORM_TYPE": "batchnorm",
           "NUM_SPLITS": 1, "NUM_SYNC_DEVICES": 1, "GLOBAL_SYNC": False},
    "TRAIN": {"ENABLE": True, "DATASET": "kinetics", "BATCH_SIZE": 64, "MIXED_PRECISION": False},
    "TEST": {"ENABLE": True, "DATASET": "kinetics", "BATCH_SIZE": 8},
    "RESNET": {"TRANS_FUNC": "bottleneck_transform", "NUM_GROUPS": 1, "WIDTH_PER_GROUP": 64, "INPLACE_RELU": True,
               "STRIDE_1X1": False, "ZERO_INIT_FINAL_BN": False, "ZERO_INIT_FINAL_CONV": False, "DEPTH": 50,
               "NUM_BLOCK_TEMP_KERNEL": [[3], [4], [6], [3]], "SPATIAL_STRIDES": [[1], [2], [2], [2]],
               "SPATIAL_DILATIONS": [[1], [1], [1], [1]]},
    "NONLOCAL": {"LOCATION": [[[]], [[]], [[]], [[]]], "GROUP": [[1], [1], [1], [1]], "INSTANTIATION": "dot_product",
                 "POOL": [[[1, 2, 2], [1, 2, 2]], [[1, 2, 2], [1, 2, 2]], [[1, 2, 2], [1, 2, 2]],
                          [[1, 2, 2], [1, 2, 2]]]},
    "MODEL": {"ARCH": "slowfast", "MODEL_NAME": "SlowFast", "NUM_CLASSES": 400, "LOSS_FUNC": "cross_entropy",
              "SINGLE_PATHWAY_ARCH": ["2d", "c2d", "i3d", "slow", "x3d", "mvit", "maskmvit"],
              "MULTI_PATHWAY_ARCH": ["slowfast"], "DROPOUT_RATE": 0.5, "DROPCONNECT_RATE": 0.0, "FC_INIT_STD": 0.01,
              "HEAD_ACT": "softmax", "ACT_CHECKPOINT": False, "DETACH_FINAL_FC": False, "FROZEN_BN": False,
              "FP16_ALLREDUCE": False},
    "SLOWFAST": {"BETA_INV": 8, "ALPHA": 8, "FUSION_CONV_CHANNEL_RATIO": 2, "FUSION_KERNEL_SZ": 5},
    "DATA": {"NUM_FRAMES": 8, "SAMPLING_RATE": 8, "MEAN": [0.45, 0.45, 0.45], "INPUT_CHANNEL_NUM": [3, 3],
             "STD": [0.225, 0.225, 0.225], "TRAIN_CROP_SIZE": 224, "TEST_CROP_SIZE": 256,
             "REVERSE_INPUT_CHANNEL": False},
    "SOLVER": {"BASE_LR": 0.1, "MOMENTUM": 0.9, "DAMPENING": 0.0, "NESTEROV": True, "WEIGHT_DECAY": 1e-4,
               "OPTIMIZING_METHOD": "sgd", "ZERO_WD_1D_PARAM": False, "CLIP_GRAD_VAL": None,
               "CLIP_GRAD_L2NORM": None, "LAYER_DECAY": 1.0},
    "DETECTION": {"ENABLE": False, "ALIGNED": True, "SPATIAL_SCALE_FACTOR": 16, "ROI_XFORM_RESOLUTION": 7},
    "MULTIGRID": {"SHORT_CYCLE": False, "LONG_CYCLE": False},
    "CONTRASTIVE": {"NUM_MLP_LAYERS": 1, "MLP_DIM": 2048, "BN_MLP": False, "BN_SYNC_MLP": False,
                    "PREDICTOR_DEPTHS": []},
    # slowfast/config/defaults.py:447-558
    "MVIT": {"MODE": "conv", "POOL_FIRST": False, "CLS_EMBED_ON": True, "PATCH_KERNEL": [3, 7, 7],
             "PATCH_STRIDE": [2, 4, 4], "PATCH_PADDING": [2, 4, 4], "PATCH_2D": False, "EMBED_DIM": 96, "NUM_HEADS": 1,
             "MLP_RATIO": 4.0, "QKV_BIAS": True, "DROPPATH_RATE": 0.1, "LAYER_SCALE_INIT_VALUE": 0.0, "DEPTH": 16,
             "NORM": "layernorm", "DIM_MUL": [], "HEAD_MUL": [], "POOL_KV_STRIDE": [], "POOL_KV_STRIDE_ADAPTIVE": None,
             "POOL_Q_STRIDE": [], "POOL_KVQ_KERNEL": None, "ZERO_DECAY_POS_CLS": True, "NORM_STEM": False,
             "SEP_POS_EMBED": False, "DROPOUT_RATE": 0.0, "USE_ABS_POS": True, "REL_POS_SPATIAL": False,
             "REL_POS_TEMPORAL": False, "REL_POS_ZERO_INIT": False, "RESIDUAL_POOLING": False, "DIM_MUL_IN_ATT": False,
             "SEPARATE_QKV": False, "HEAD_INIT_SCALE": 1.0, "USE_MEAN_POOLING": False, "USE_FIXED_SINCOS_POS": False,
             # slowfast/config/defaults.py:612-628
             "REV": {"ENABLE": False, "RESPATH_FUSE": "concat", "BUFFER_LAYERS": [], "RES_PATH": "conv",
                     "PRE_Q_FUSION": "avg"}},
    # slowfast/config/defaults.py:333-358
    "X3D": {"WIDTH_FACTOR": 1.0, "DEPTH_FACTOR": 1.0, "BOTTLENECK_FACTOR": 1.0, "DIM_C5": 2048, "DIM_C1": 12,
            "SCALE_RES2": False, "BN_LIN5": False, "CHANNELWISE_3x3x3": True},
    "MIXUP": {"ENABLE": False},
    "NUM_GPUS": 1, "NUM_SHARDS": 1, "SHARD_ID": 0, "RNG_SEED": 1, "LOG_MODEL_INFO": True, "DIST_BACKEND": "nccl",
    "OUTPUT_DIR": ".",
}


def get_cfg():
    """Defaults of the hot-path keys (values as in slowfast/config/defaults.py)."""
    return CfgNode(copy.deepcopy(_DEFAULTS))


# The two Kinetics configs named by BASELINE.json, as overlays on the defaults
# (values from configs/Kinetics/SLOWFAST_8x8_R50.yaml and configs/Kinetics/C2D_8x8_R50.yaml).
PRESETS = {
    "SLOWFAST_8x8_R50": {
        "DATA": {"NUM_FRAMES": 32, "SAMPLING_RATE": 2, "TRAIN_CROP_SIZE": 224, "TEST_CROP_SIZE": 256,
                 "INPUT_CHANNEL_NUM": [3, 3]},
        "SLOWFAST": {"ALPHA": 4, "BETA_INV": 8, "FUSION_CONV_CHANNEL_RATIO": 2, "FUSION_KERNEL_SZ": 7},
        "RESNET": {"ZERO_INIT_FINAL_BN": True, "WIDTH_PER_GROUP": 64, "NUM_GROUPS": 1, "DEPTH": 50,
                   "TRANS_FUNC": "bottleneck_transform", "STRIDE_1X1": False,
                   "NUM_BLOCK_TEMP_KERNEL": [[3, 3], [4, 4], [6, 6], [3, 3]],
                   "SPATIAL_STRIDES": [[1, 1], [2, 2], [2, 2], [2, 2]],
                   "SPATIAL_DILATIONS": [[1, 1], [1, 1], [1, 1], [1, 1]]},
        "NONLOCAL": {"LOCATION": [[[], []], [[], []], [[], []], [[], []]],
                     "GROUP": [[1, 1], [1, 1], [1, 1], [1, 1]], "INSTANTIATION": "dot_product"},
        "MODEL": {"NUM_CLASSES": 400, "ARCH": "slowfast", "MODEL_NAME": "SlowFast", "LOSS_FUNC": "cross_entropy",
                  "DROPOUT_RATE": 0.5},
    },
    "C2D_8x8_R50": {
        "DATA": {"NUM_FRAMES": 8, "SAMPLING_RATE": 8, "TRAIN_CROP_SIZE": 224, "TEST_CROP_SIZE": 256,
                 "INPUT_CHANNEL_NUM": [3]},
        "RESNET": {"ZERO_INIT_FINAL_BN": True, "WIDTH_PER_GROUP": 64, "NUM_GROUPS": 1, "DEPTH": 50,
                   "TRANS_FUNC": "bottleneck_transform", "STRIDE_1X1": False,
                   "NUM_BLOCK_TEMP_KERNEL": [[3], [4], [6], [3]]},
        "NONLOCAL": {"LOCATION": [[[]], [[]], [[]], [[]]], "GROUP": [[1], [1], [1], [1]],
                     "INSTANTIATION": "softmax"},
        "MODEL": {"NUM_CLASSES": 400, "ARCH": "c2d", "MODEL_NAME": "ResNet", "LOSS_FUNC": "cross_entropy",
                  "DROPOUT_RATE": 0.5},
    },
}


# configs/Kinetics/MVITv2_S_16x4.yaml
PRESETS["MVITv2_S_16x4"] = {
    "DATA": {"NUM_FRAMES": 16, "SAMPLING_RATE": 4, "TRAIN_CROP_SIZE": 224, "TEST_CROP_SIZE": 224,
             "INPUT_CHANNEL_NUM": [3]},
    "MVIT": {"ZERO_DECAY_POS_CLS": False, "USE_ABS_POS": False, "REL_POS_SPATIAL": True, "REL_POS_TEMPORAL": True,
             "DEPTH": 16, "NUM_HEADS": 1, "EMBED_DIM": 96, "PATCH_KERNEL": [3, 7, 7], "PATCH_STRIDE": [2, 4, 4],
             "PATCH_PADDING": [1, 3, 3], "MLP_RATIO": 4.0, "QKV_BIAS": True, "DROPPATH_RATE": 0.2, "NORM": "layernorm",
             "MODE": "conv", "CLS_EMBED_ON": True, "DIM_MUL": [[1, 2.0], [3, 2.0], [14, 2.0]],
             "HEAD_MUL": [[1, 2.0], [3, 2.0], [14, 2.0]], "POOL_KVQ_KERNEL": [3, 3, 3],
             "POOL_KV_STRIDE_ADAPTIVE": [1, 8, 8],
             "POOL_Q_STRIDE": [[0, 1, 1, 1], [1, 1, 2, 2], [2, 1, 1, 1], [3, 1, 2, 2]] + [[i, 1, 1, 1] for i in range(4, 14)]
             + [[14, 1, 2, 2], [15, 1, 1, 1]],
             "DROPOUT_RATE": 0.0, "DIM_MUL_IN_ATT": True, "RESIDUAL_POOLING": True},
    "SOLVER": {"BASE_LR": 0.0001, "MOMENTUM": 0.9, "WEIGHT_DECAY": 0.05, "OPTIMIZING_METHOD": "adamw",
               "ZERO_WD_1D_PARAM": True, "CLIP_GRAD_L2NORM": 1.0},
    "MODEL": {"NUM_CLASSES": 400, "ARCH": "mvit", "MODEL_NAME": "MViT", "LOSS_FUNC": "soft_cross_entropy",
              "DROPOUT_RATE": 0.5},
}


# configs/Kinetics/MVIT_B_16x4_CONV.yaml (MViTv1: separate learned position embeddings, dimension change after the Mlp,
# q pooling in blocks 1 / 3 / 14 only)
PRESETS["MVIT_B_16x4_CONV"] = {
    "DATA": {"NUM_FRAMES": 16, "SAMPLING_RATE": 4, "TRAIN_CROP_SIZE": 224, "TEST_CROP_SIZE": 224,
             "INPUT_CHANNEL_NUM": [3]},
    "MVIT": {"ZERO_DECAY_POS_CLS": False, "SEP_POS_EMBED": True, "DEPTH": 16, "NUM_HEADS": 1, "EMBED_DIM": 96,
             "PATCH_KERNEL": [3, 7, 7], "PATCH_STRIDE": [2, 4, 4], "PATCH_PADDING": [1, 3, 3], "MLP_RATIO": 4.0,
             "QKV_BIAS": True, "DROPPATH_RATE": 0.2, "NORM": "layernorm", "MODE": "conv", "CLS_EMBED_ON": True,
             "DIM_MUL": [[1, 2.0], [3, 2.0], [14, 2.0]], "HEAD_MUL": [[1, 2.0], [3, 2.0], [14, 2.0]],
             "POOL_KVQ_KERNEL": [3, 3, 3], "POOL_KV_STRIDE_ADAPTIVE": [1, 8, 8],
             "POOL_Q_STRIDE": [[1, 1, 2, 2], [3, 1, 2, 2], [14, 1, 2, 2]], "DROPOUT_RATE": 0.0},
    "SOLVER": {"BASE_LR": 0.0001, "MOMENTUM": 0.9, "WEIGHT_DECAY": 0.05, "OPTIMIZING_METHOD": "adamw",
               "ZERO_WD_1D_PARAM": True, "CLIP_GRAD_L2NORM": 1.0},
    "MODEL": {"NUM_CLASSES": 400, "ARCH": "mvit", "MODEL_NAME": "MViT", "LOSS_FUNC": "soft_cross_entropy",
              "DROPOUT_RATE": 0.5},
}

# configs/Kinetics/REV_MVIT_B_16x4_CONV.yaml (reversible MViT-B).  CLS_EMBED_ON is True in the shipped yaml, which the
# reference's constructor rejects ("rev does not allow cls token", video_model_builder.py:966): pass
# MVIT.CLS_EMBED_ON False, as a user of the reference has to.
PRESETS["REV_MVIT_B_16x4_CONV"] = {
    "DATA": {"NUM_FRAMES": 16, "SAMPLING_RATE": 4, "TRAIN_CROP_SIZE": 224, "TEST_CROP_SIZE": 224,
             "INPUT_CHANNEL_NUM": [3]},
    "MVIT": {"ZERO_DECAY_POS_CLS": False, "CLS_EMBED_ON": True, "SEP_POS_EMBED": True, "USE_ABS_POS": True, "DEPTH": 16,
             "NUM_HEADS": 1, "EMBED_DIM": 96, "PATCH_KERNEL": [3, 7, 7], "PATCH_STRIDE": [2, 4, 4],
             "PATCH_PADDING": [1, 3, 3], "MLP_RATIO": 4.0, "QKV_BIAS": False, "DROPPATH_RATE": 0.05, "NORM": "layernorm",
             "MODE": "conv", "DIM_MUL": [[1, 2.0], [3, 2.0], [14, 2.0]], "HEAD_MUL": [[1, 2.0], [3, 2.0], [14, 2.0]],
             "POOL_KVQ_KERNEL": [3, 3, 3], "POOL_KV_STRIDE_ADAPTIVE": [1, 8, 8],
             "POOL_Q_STRIDE": [[1, 1, 2, 2], [3, 1, 2, 2], [14, 1, 2, 2]],
             "REV": {"ENABLE": True, "RESPATH_FUSE": "concat", "BUFFER_LAYERS": [1, 3, 14], "RES_PATH": "conv"}},
    "SOLVER": {"BASE_LR": 0.0001, "MOMENTUM": 0.9, "WEIGHT_DECAY": 7e-2, "OPTIMIZING_METHOD": "adamw",
               "ZERO_WD_1D_PARAM": True},
    "MODEL": {"NUM_CLASSES": 400, "ARCH": "slow", "MODEL_NAME": "MViT", "LOSS_FUNC": "soft_cross_entropy",
              "DROPOUT_RATE": 0.5},
}

# configs/masked_ssl/k400_VIT_B_16x4_FT.yaml (plain video ViT-B fine-tuning: no pooling, mean pooling before the norm)
PRESETS["k400_VIT_B_16x4_FT"] = {
    "DATA": {"NUM_FRAMES": 16, "SAMPLING_RATE": 4, "TRAIN_CROP_SIZE": 224, "TEST_CROP_SIZE": 224,
             "INPUT_CHANNEL_NUM": [3]},
    "MVIT": {"ZERO_DECAY_POS_CLS": False, "SEP_POS_EMBED": True, "PATCH_KERNEL": [2, 16, 16], "PATCH_STRIDE": [2, 16, 16],
             "PATCH_PADDING": [0, 0, 0], "EMBED_DIM": 768, "NUM_HEADS": 12, "MLP_RATIO": 4.0, "QKV_BIAS": True,
             "NORM": "layernorm", "DEPTH": 12, "MODE": "conv", "DROPPATH_RATE": 0.1, "LAYER_SCALE_INIT_VALUE": 0.0,
             "USE_MEAN_POOLING": True, "HEAD_INIT_SCALE": 0.001},
    "SOLVER": {"BASE_LR": 6e-4, "WEIGHT_DECAY": 0.05, "OPTIMIZING_METHOD": "adamw", "ZERO_WD_1D_PARAM": True,
               "CLIP_GRAD_L2NORM": 5.0},
    "MODEL": {"NUM_CLASSES": 400, "ARCH": "mvit", "MODEL_NAME": "MViT", "LOSS_FUNC": "soft_cross_entropy",
              "DROPOUT_RATE": 0.3},
}


# configs/Kinetics/X3D_M.yaml
PRESETS["X3D_M"] = {
    "DATA": {"NUM_FRAMES": 16, "SAMPLING_RATE": 5, "TRAIN_CROP_SIZE": 224, "TEST_CROP_SIZE": 256, "INPUT_CHANNEL_NUM": [3]},
    "X3D": {"WIDTH_FACTOR": 2.0, "DEPTH_FACTOR": 2.2, "BOTTLENECK_FACTOR": 2.25, "DIM_C5": 2048, "DIM_C1": 12},
    "RESNET": {"ZERO_INIT_FINAL_BN": True, "TRANS_FUNC": "x3d_transform", "STRIDE_1X1": False},
    "SOLVER": {"BASE_LR": 0.1, "WEIGHT_DECAY": 5e-5, "OPTIMIZING_METHOD": "sgd"},
    "MODEL": {"NUM_CLASSES": 400, "ARCH": "x3d", "MODEL_NAME": "X3D", "LOSS_FUNC": "cross_entropy", "DROPOUT_RATE": 0.5},
}


def _variant(base, **model):
    v = copy.deepcopy(PRESETS[base])
    v["MODEL"].update(model)
    return v


PRESETS["SLOW_8x8_R50"] = _variant("C2D_8x8_R50", ARCH="slow")      # configs/Kinetics/SLOW_8x8_R50.yaml
PRESETS["SLOW_8x8_R50"]["NONLOCAL"]["INSTANTIATION"] = "dot_product"
PRESETS["I3D_8x8_R50"] = _variant("C2D_8x8_R50", ARCH="i3d")        # configs/Kinetics/I3D_8x8_R50.yaml


# configs/Kinetics/C2D_NLN_8x8_R50.yaml, configs/Kinetics/SLOWFAST_NLN_8x8_R50.yaml
PRESETS["C2D_NLN_8x8_R50"] = copy.deepcopy(PRESETS["C2D_8x8_R50"])
PRESETS["C2D_NLN_8x8_R50"]["NONLOCAL"] = {"LOCATION": [[[]], [[1, 3]], [[1, 3, 5]], [[]]], "GROUP": [[1], [1], [1], [1]],
                                          "INSTANTIATION": "softmax"}
PRESETS["SLOWFAST_NLN_8x8_R50"] = copy.deepcopy(PRESETS["SLOWFAST_8x8_R50"])
PRESETS["SLOWFAST_NLN_8x8_R50"]["SLOWFAST"]["FUSION_KERNEL_SZ"] = 5
PRESETS["SLOWFAST_NLN_8x8_R50"]["NONLOCAL"] = {"LOCATION": [[[], []], [[1, 3], []], [[1, 3, 5], []], [[], []]],
                                               "GROUP": [[1, 1], [1, 1], [1, 1], [1, 1]], "INSTANTIATION": "dot_product"}


# configs/AVA/c2/SLOWFAST_32x2_R101_50_50.yaml -- BASELINE config 5: SlowFast-R101 + Nonlocal with the AVA RoI head
# (DETECTION.ENABLE, legacy ROIAlign alignment, sigmoid outputs, BCE).
PRESETS["SLOWFAST_32x2_R101_50_50"] = copy.deepcopy(PRESETS["SLOWFAST_8x8_R50"])
PRESETS["SLOWFAST_32x2_R101_50_50"]["DATA"].update({"TRAIN_CROP_SIZE": 224, "TEST_CROP_SIZE": 256})
PRESETS["SLOWFAST_32x2_R101_50_50"]["SLOWFAST"]["FUSION_KERNEL_SZ"] = 5
PRESETS["SLOWFAST_32x2_R101_50_50"]["RESNET"].update({"DEPTH": 101, "SPATIAL_DILATIONS": [[1, 1], [1, 1], [1, 1], [2, 2]],
                                                      "SPATIAL_STRIDES": [[1, 1], [2, 2], [2, 2], [1, 1]]})
PRESETS["SLOWFAST_32x2_R101_50_50"]["NONLOCAL"] = {
    "LOCATION": [[[], []], [[], []], [[6, 13, 20], []], [[], []]], "GROUP": [[1, 1], [1, 1], [1, 1], [1, 1]],
    "INSTANTIATION": "dot_product",
    "POOL": [[[2, 2, 2], [2, 2, 2]], [[2, 2, 2], [2, 2, 2]], [[2, 2, 2], [2, 2, 2]], [[2, 2, 2], [2, 2, 2]]]}
PRESETS["SLOWFAST_32x2_R101_50_50"]["SOLVER"] = {"MOMENTUM": 0.9, "WEIGHT_DECAY": 1e-7, "OPTIMIZING_METHOD": "sgd"}
PRESETS["SLOWFAST_32x2_R101_50_50"]["MODEL"].update({"NUM_CLASSES": 80, "LOSS_FUNC": "bce", "HEAD_ACT": "sigmoid"})
PRESETS["SLOWFAST_32x2_R101_50_50"]["DETECTION"] = {"ENABLE": True, "ALIGNED": False}


def preset_for_yaml(yaml_rel):
    """Preset name for a reference YAML path such as 'configs/Kinetics/SLOWFAST_8x8_R50.yaml'."""
    import os
    name = os.path.splitext(os.path.basename(yaml_rel))[0]
    if name not in PRESETS:
        raise KeyError(f"no built-in preset for {yaml_rel}; load the YAML with CfgNode.merge_from_file")
    return name


def get_preset(name, opts=()):
    cfg = get_cfg()
    cfg.merge_from_dict(PRESETS[name])
    if opts:
        cfg.merge_from_list(list(opts))
    return cfg
