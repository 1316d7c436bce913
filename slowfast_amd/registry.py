"""Model registry and factory with the reference's plugin surface.

Mirrors slowfast/models/build.py:13-81: ``MODEL_REGISTRY.register()`` decorates a class whose
constructor takes ``cfg``; ``build_model(cfg, gpu_id)`` instantiates ``cfg.MODEL.MODEL_NAME`` and
moves it to the current device and, for NUM_GPUS > 1, wraps it for data parallelism (see build_model).
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


def build_model(cfg, gpu_id=None, data_parallel=None):
    """Same contract as slowfast/models/build.py:22-81: construct ``MODEL_REGISTRY[cfg.MODEL.MODEL_NAME](cfg)``, move it
    to the process's GPU and -- with ``cfg.NUM_GPUS > 1`` inside an initialised process group -- return it wrapped for
    multi-process data parallelism, so tools/train_net.py keeps working unchanged.

    ``data_parallel`` (default: env SF_DATA_PARALLEL or "ddp"):
      * "ddp": ``torch.nn.parallel.DistributedDataParallel`` exactly as the reference wraps (device_ids / output_device /
        find_unused_parameters; backend "nccl" is RCCL over xGMI on ROCm).  The engine then returns parameter gradients
        through autograd (engine.GRADS_VIA_AUTOGRAD) so that DDP's reducer -- its bucketing, its overlap with backward and
        any ``register_comm_hook`` hook -- sees them.  MODEL.FP16_ALLREDUCE installs
        data_parallel.fp16_compress_hook (the reference's comm_hooks_default.fp16_compress_hook semantics); otherwise
        data_parallel.xgmi_allreduce_hook (few, large fp32 collectives sized for point-to-point xGMI links).
      * "reducer" / False: the bare module; the caller drives slowfast_amd.data_parallel.GradReducer + step.TrainStep
        (flat gradient memory written in place by the backward kernels, HIP-graph replay -- what bench.py times)."""
    import os

    from . import mvit, video_models, x3d  # noqa: F401  (registers the model classes)
    if torch.cuda.is_available():
        assert cfg.NUM_GPUS <= torch.cuda.device_count(), "Cannot use more GPU devices than available"
    model = MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)
    dev = None
    if cfg.NUM_GPUS:
        dev = torch.cuda.current_device() if gpu_id is None else gpu_id
        model = model.cuda(device=dev)
    mode = os.environ.get("SF_DATA_PARALLEL", "ddp") if data_parallel is None else data_parallel
    import torch.distributed as dist
    if cfg.NUM_GPUS > 1 and mode not in (False, "reducer") and dist.is_available() and dist.is_initialized():
        from . import data_parallel as dp
        model = dp.wrap_ddp(model, device=dev, find_unused_parameters=bool(cfg.MODEL.DETACH_FINAL_FC),
                            fp16_allreduce=bool(cfg.MODEL.FP16_ALLREDUCE))
    return model
