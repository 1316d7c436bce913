"""Model registry and factory with the reference's plugin surface.

Mirrors slowfast/models/build.py:13-81: ``MODEL_REGISTRY.register()`` decorates a class whose
constructor takes ``cfg``; ``build_model(cfg, gpu_id)`` instantiates ``cfg.MODEL.MODEL_NAME`` and
moves it to the current device.  Data parallelism is NOT torch DDP here: see data_parallel.py.
"""
import torch


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj = {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._add(o)
                return o
            return deco
        self._add(obj)
        return obj

    def _add(self, obj):
        name = obj.__name__
        assert name not in self._obj, f"'{name}' already registered in '{self._name}'"
        self._obj[name] = obj

    def get(self, name):
        if name not in self._obj:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return self._obj[name]

    def __contains__(self, name):
        return name in self._obj


MODEL_REGISTRY = Registry("MODEL")


def build_model(cfg, gpu_id=None):
    """Same contract as slowfast/models/build.py:22-81 (minus the DDP wrap, replaced by
    slowfast_amd.data_parallel.GradReducer when torch.distributed is initialised)."""
    from . import mvit, video_models, x3d  # noqa: F401  (registers the model classes)
    if torch.cuda.is_available():
        assert cfg.NUM_GPUS <= torch.cuda.device_count(), "Cannot use more GPU devices than available"
    model = MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)
    if cfg.NUM_GPUS:
        dev = torch.cuda.current_device() if gpu_id is None else gpu_id
        model = model.cuda(device=dev)
    return model
