"""ctypes binding of libsfamd.so (C ABI declared in include/sfamd.h).

The product path is the hipcc-built gfx950 library next to this file.  There is NO fallback: if the
library is missing, or tensors are not on a GPU, calls raise.  The CPU test-suite points
``SFAMD_LIBRARY`` at the host functional simulator (tests/hostsim, the same kernel sources compiled
for the host) to exercise the host logic and the kernels' index math without a GPU; that build
identifies itself through ``sf_backend() == "hostsim"`` and is never picked up implicitly.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 23
# 16-bit storage type of activations / packed weights / MFMA operands, fixed per PROCESS: SF_ACT_DTYPE=fp16 (default) loads
# libsfamd.so, =bf16 loads libsfamd_bf16.so -- the same sources compiled with -DSF_ACT_BF16 (bfloat16 storage,
# v_mfma_f32_16x16x32_bf16); both are what torch.cuda.amp.autocast admits on the reference side (tools/train_net.py:101-118).
ACT_MODE = os.environ.get("SF_ACT_DTYPE", "fp16").lower()
if ACT_MODE not in ("fp16", "bf16"):
    raise ValueError(f"SF_ACT_DTYPE={ACT_MODE!r}: expected fp16 or bf16")
DEFAULT_LIBRARY = os.path.join(_HERE, "libsfamd.so" if ACT_MODE == "fp16" else "libsfamd_bf16.so")


def act_dtype():
    """torch dtype of every activation / packed-weight buffer handed to the library in this process."""
    import torch
    return torch.float16 if ACT_MODE == "fp16" else torch.bfloat16


def act_eps():
    """Unit round-off of the storage type (2^-11 fp16, 2^-8 bf16): test tolerances are stated for fp16 and scaled by
    act_eps() / 2^-11 in a bf16 process."""
    return 2.0 ** -11 if ACT_MODE == "fp16" else 2.0 ** -8


class ConvDesc(Structure):
    """Mirror of ``sf_conv_desc``."""

    _fields_ = [(n, c_int32) for n in (
        "N", "Ci", "Ti", "Hi", "Wi", "Co", "To", "Ho", "Wo", "kT", "kH", "kW", "sT", "sH", "sW",
        "pT", "pH", "pW", "dT", "dH", "dW", "Cw", "ldx", "ldy", "Cow")]


class PrepItem(Structure):
    """Mirror of ``sf_prep_item`` (one weight of a batched sf_prep_weights_batch launch)."""

    _fields_ = [("w", c_void_p), ("wf", c_void_p), ("wd", c_void_p)] + [(n, c_int32) for n in (
        "Co", "Cow", "Cw", "Cp", "taps", "ldf", "ldd", "pad")]


class DwDesc(Structure):
    """Mirror of ``sf_dw_desc``."""

    _fields_ = [(n, c_int32) for n in (
        "N", "C", "Cw", "cls", "Ti", "Hi", "Wi", "To", "Ho", "Wo", "kT", "kH", "kW", "sT", "sH", "sW",
        "pT", "pH", "pW", "ldx", "ldy", "Cwreal")]


class Rows32(Structure):
    """Mirror of ``sf_rows32``: fp32 side rows of a token residual stream."""

    _fields_ = [("in_", c_void_p), ("out", c_void_p), ("ld", c_int32), ("period", c_int32)]


class ColFinItem(Structure):
    """Mirror of ``sf_colfin_item`` (one finalize of a batched sf_colsum_finalize_batch launch)."""

    _fields_ = [("part", c_void_p), ("nblk", c_int32), ("C", c_int32), ("fold", c_int32), ("out0", c_void_p),
                ("out1", c_void_p), ("scale", c_float), ("accumulate", c_int32), ("row_stride", c_int32)]


class AttnDesc(Structure):
    """Mirror of ``sf_attn_desc``."""

    _fields_ = [(n, c_int32) for n in (
        "B", "heads", "D", "cls", "Nq", "qT", "qH", "qW", "Nk", "kT", "kH", "kW", "rows_h", "rows_w", "rows_t")]


_P = c_void_p
_F = c_void_p  # float* passed as raw address
_SIGNATURES = {
    "sf_abi_version": (c_int, []),
    "sf_backend": (c_char_p, []),
    "sf_act_dtype": (c_int, []),
    "sf_last_error": (c_char_p, []),
    "sf_build_id": (c_char_p, []),
    "sf_conv_weight_ld": (c_int, [POINTER(ConvDesc), POINTER(c_int32), POINTER(c_int32)]),
    "sf_prep_weights": (c_int, [POINTER(ConvDesc), _F, _P, _P, _P]),
    "sf_prep_item_fill": (c_int, [POINTER(ConvDesc), _F, _P, _P, POINTER(PrepItem)]),
    "sf_prep_item_blocks": (c_int64, [POINTER(PrepItem)]),
    "sf_prep_weights_batch": (c_int, [_P, _P, _P, c_int32, _P]),
    "sf_conv_fwd_mtiles": (c_int, [POINTER(ConvDesc)]),
    "sf_conv_fwd": (c_int, [POINTER(ConvDesc), _P, _P, _F, _F, c_int, _F, _P, _F, _P]),
    "sf_conv_fwd_fused": (c_int, [POINTER(ConvDesc), _P, _P, _F, _P, c_int32, c_int, _P, _P]),
    "sf_conv_dgrad": (c_int, [POINTER(ConvDesc), _P, _P, _P, c_int32, _P, _P, _P]),
    "sf_conv_dgrad_bn": (c_int, [POINTER(ConvDesc), _P, _P, _P, c_int32, _P, _P, _P, _P, _P, _P, c_int32, _P, c_int32,
                                 POINTER(c_int32), _P]),
    "sf_conv_wgrad_workspace": (c_int64, [POINTER(ConvDesc)]),
    "sf_conv_wgrad_rowtab_bytes": (c_int64, [POINTER(ConvDesc)]),
    "sf_conv_wgrad_rowtab": (c_int, [POINTER(ConvDesc), _P, _P]),
    "sf_conv_wgrad": (c_int, [POINTER(ConvDesc), _P, _F, _F, c_int, _P, _F, c_float, c_int, _P, c_int64, _P, _P]),
    "sf_bn_finalize": (c_int, [_F, c_int32, c_int32, c_int32, c_float, _F, _F, _F, _F, c_float, c_float, _F, _F, _F, _F, _P]),
    "sf_bn_act": (c_int, [c_int64, c_int32, _P, c_int32, _F, _F, _P, c_int32, _F, _F, c_int, _P, c_int32, _P, _P]),
    "sf_bn_bwd_blocks": (c_int, [c_int64, c_int32]),
    "sf_flat_blocks": (c_int, [c_int64]),
    "sf_flat_sumsq": (c_int, [_F, c_int64, _F, _P]),
    "sf_step_control": (c_int, [_F, c_int32, _F, c_float, c_float, c_int, c_float, c_float, c_int32, _P]),
    "sf_flat_sgd": (c_int, [_F, _F, _F, _P, _P, _P, c_int32, _F, _P, _P, c_int32, c_float, c_float, c_float, c_int, _P]),
    "sf_flat_adamw": (c_int, [_F, _F, _F, _F, _P, _P, _P, c_int32, _F, _P, _P, c_int32, c_float, c_float, c_float, c_float, _P]),
    "sf_bn_bwd_reduce": (c_int, [c_int64, c_int32, _P, c_int32, _P, c_int32, _P, c_int32, _F, _F, c_int, _F, _P]),
    "sf_bn_bwd_finalize": (c_int, [_F, c_int32, c_int32, c_int32, c_float, _F, _F, _F, c_float, _F, _F, c_int, _F, _P]),
    "sf_bn_bwd_apply": (c_int, [c_int64, c_int32, _P, c_int32, _P, c_int32, _P, c_int32, _F, _F, c_int, _F, _P,
                                c_int32, _P, c_int32, _P]),
    "sf_pool_fwd": (c_int, [c_int32] * 11 + [_P, c_int32, _F, _F, c_int, _P, c_int32, _P, c_int32, _P]),
    "sf_pool_bwd": (c_int, [c_int32] * 11 + [_P, c_int32, _P, c_int, _P, c_int32, _P, c_int32, c_int32, _P]),
    "sf_pool3d_fwd": (c_int, [c_int32] * 8 + [_P, c_int32, _P, c_int32, _P, _P]),
    "sf_pool3d_bwd": (c_int, [c_int32] * 8 + [_P, _P, c_int32, _P, c_int32, _P]),
    "sf_ncthw_to_cl": (c_int, [_F, c_int32, c_int32, c_int64, c_int32, _P, _P]),
    "sf_cl_to_ncthw": (c_int, [_P, c_int32, c_int32, c_int32, c_int64, _F, _P]),
    "sf_bgemm": (c_int, [c_int64, c_int32, c_int32, _P, c_int32, _P, c_int32, _F, _P, c_int32, _P, c_int32, c_int32,
                         c_int32] + [c_int64] * 8 + [c_int32, c_float, _P]),
    "sf_bgemm_tn": (c_int, [c_int64, c_int32, c_int32, _P, c_int32, _P, c_int32, _P, c_int32, c_float, c_int32, c_int32]
                    + [c_int64] * 6 + [_P]),
    "sf_layernorm_fwd": (c_int, [c_int64, c_int32, _P, c_int32, _F, _F, c_float, _P, c_int32, _F, _F, _P]),
    "sf_gemm_rows32": (c_int, [c_int64, c_int32, c_int32, _P, c_int32, _P, c_int32, _F, _P, c_int32, _P, c_int32,
                               POINTER(Rows32), _P]),
    "sf_layernorm_fwd_rows32": (c_int, [c_int64, c_int32, _P, c_int32, _F, _F, c_float, _P, c_int32, _F, _F,
                                        POINTER(Rows32), _P]),
    "sf_layernorm_bwd_blocks": (c_int, [c_int64, c_int32]),
    "sf_layernorm_bwd": (c_int, [c_int64, c_int32, _P, c_int32, _P, c_int32, _F, _F, _F, _P, c_int32, _P, c_int32, _F, _P]),
    "sf_layernorm_bwd_sums": (c_int, [c_int64, c_int32, _P, c_int32, _P, c_int32, _F, _F, _F, _P, c_int32, _P, c_int32, _F, _P]),
    "sf_colsum_blocks": (c_int, [c_int64, c_int32]),
    "sf_colsum": (c_int, [c_int64, c_int32, _P, c_int32, _F, _P]),
    "sf_colsum_finalize": (c_int, [_F, c_int32, c_int32, c_int32, _F, _F, c_float, c_int, _P]),
    "sf_colsum_finalize_batch": (c_int, [POINTER(ColFinItem), c_int32, _P]),
    "sf_rows_sum": (c_int, [_F, c_int32, c_int64, c_int64, c_int32, _F, c_float, c_int, _P]),
    "sf_gelu_fwd": (c_int, [c_int64, _P, _P, _P]),
    "sf_gelu_bwd": (c_int, [c_int64, _P, _P, _P, _P]),
    "sf_dwconv_fwd_blocks": (c_int, [POINTER(DwDesc)]),
    "sf_dwconv_fwd_sample_rows": (c_int, [POINTER(DwDesc)]),
    "sf_dwconv_pair_ok": (c_int, [POINTER(DwDesc)]),
    "sf_dwconv_fwd_pair": (c_int, [POINTER(DwDesc), _P, _P, _F, _F, _P, _P, _P]),
    "sf_dwconv_dgrad_pair": (c_int, [POINTER(DwDesc), _P, _P, _F, _F, _P, _P, _P]),
    "sf_dwconv_wgrad_pair": (c_int, [POINTER(DwDesc), _P, _P, _P, _P, _F, _F, c_float, c_int, c_int, _P, c_int64, _P]),
    "sf_dwconv_fwd": (c_int, [POINTER(DwDesc), _P, _F, _P, _F, _P]),
    "sf_dwconv_dgrad": (c_int, [POINTER(DwDesc), _P, _F, _P, _P]),
    "sf_dwconv_dgrad_sum_rows": (c_int, [POINTER(DwDesc)]),
    "sf_dwconv_dgrad_sums": (c_int, [POINTER(DwDesc), _P, _F, _P, _F, _P]),
    "sf_dwconv_wgrad_workspace": (c_int64, [POINTER(DwDesc)]),
    "sf_dwconv_wgrad": (c_int, [POINTER(DwDesc), _P, _P, _F, c_float, c_int, _P, c_int64, _P]),
    "sf_relpos_gather": (c_int, [POINTER(AttnDesc), _P, c_int32, _P, _P, _P, _F, _P]),
    "sf_relpos_scatter": (c_int, [POINTER(AttnDesc), _F, _P, _P, _P, _P, c_int32, _P]),
    "sf_relpos_pack": (c_int, [_F, _F, _F, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P, _P]),
    "sf_relpos_unpack": (c_int, [_F, c_int32, c_int32, c_int32, c_int32, _F, _F, _F, c_int32, c_int32, c_int32, _P]),
    "sf_softmax_fwd": (c_int, [POINTER(AttnDesc), _P, c_int32, c_float, _F, _P]),
    "sf_softmax_bwd": (c_int, [POINTER(AttnDesc), _P, _P, c_int32, c_float, _F, _P]),
    "sf_attn_fwd": (c_int, [POINTER(AttnDesc), _P, c_int32, _P, _P, c_int32, c_float, _F, _P, c_int32, _P, c_int32, _F, _P]),
    "sf_attn_bwd_workspace": (c_int64, [POINTER(AttnDesc)]),
    "sf_attn_bwd": (c_int, [POINTER(AttnDesc), _P, c_int32, _P, _P, c_int32, c_float, _F, _P, c_int32, _P, _P, c_int32, _F,
                            _F, _P, c_int32, _P, _P, c_int32, _F, _P, c_int64, _P]),
    "sf_tmean_fwd": (c_int, [c_int32, c_int32, c_int64, c_int32, _P, c_int32, _F, _P]),
    "sf_tmean_bwd": (c_int, [c_int32, c_int32, c_int64, c_int32, _F, _P, c_int32, _P]),
    "sf_roi_align_max_fwd": (c_int, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_int32, _F, _F, _F,
                                     c_int32, c_int32, _P, _P]),
    "sf_roi_align_max_bwd": (c_int, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_int32, _F, _F,
                                     c_int32, c_int32, _P, _F, _P]),
    "sf_gemm_act": (c_int, [c_int64, c_int32, c_int32, _P, c_int32, _P, c_int32, _F, _P, c_int32, c_int32, _P, c_int32, _P]),
    "sf_gemm_act_colsum": (c_int, [c_int64, c_int32, c_int32, _P, c_int32, _P, c_int32, _F, _P, c_int32, c_int32, _P, c_int32, _P, c_int32, POINTER(c_int32), _P]),
    "sf_pack_clip_u8": (c_int, [_P, c_int32, c_int32, c_int32, c_int32, _P, c_int32, c_float, c_float, c_float, c_float,
                                c_float, c_float, c_int32, _P, _P]),
    "sf_row_scale_add": (c_int, [_P, c_int32, _P, c_int64, _P, c_int32, _P, c_int32, c_int64, c_int32, _P]),
    "sf_row_scale_add_rows32": (c_int, [_P, c_int32, _P, c_int64, _P, c_int32, _P, c_int32, c_int64, c_int32,
                                        POINTER(Rows32), _P]),
    "sf_transpose_heads": (c_int, [_P, c_int32, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    "sf_sample_chunks": (c_int, [c_int64, c_int32]),
    "sf_sample_mean": (c_int, [c_int32, c_int64, c_int32, _P, c_int32, _F, _F, c_int, _F, _F, _P]),
    "sf_se_gate_fwd": (c_int, [c_int32, c_int32, c_int32, c_int32, _F, _F, _F, _F, _F, _F, _F, _P]),
    "sf_gate_act_fwd": (c_int, [c_int32, c_int64, c_int32, _P, c_int32, _F, _F, _F, c_int, _P, c_int32, _P]),
    "sf_gate_grad": (c_int, [c_int32, c_int64, c_int32, _P, c_int32, _F, _F, _P, c_int32, _F, c_int, _F, _F, _P]),
    "sf_se_gate_bwd": (c_int, [c_int32, c_int32, c_int32, c_int32, _F, _F, _F, _F, _F, _F, _F, _F, _P]),
    "sf_outer_sum": (c_int, [_F, c_int32, _F, c_int32, c_int32, c_int32, c_int32, _F, c_float, c_int, _P]),
    "sf_gate_act_bwd": (c_int, [c_int32, c_int64, c_int32, _P, c_int32, _F, _F, _F, c_int, _P, c_int32, _F, _P, c_int32, _P]),
    "sf_gate_bwd_sums": (c_int, [c_int32, c_int64, c_int32, _P, c_int32, _F, _F, _P, c_int32, _F, c_int, _P, c_int32, _F, _F, _P]),
    "sf_bn_bwd_apply_sample": (c_int, [c_int64, c_int32, _P, c_int32, _P, c_int32, _F, _F, c_int64, _P, c_int32, _P]),
    "sf_gate_act_bwd_bn_rows": (c_int, [c_int32, c_int64, c_int32]),
    "sf_gate_act_bwd_bn": (c_int, [c_int32, c_int64, c_int32, _P, c_int32, _F, _F, _F, c_int, _P, c_int32, _F, _P, c_int32, _F, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class SfError(RuntimeError):
    pass


class SfLibrary:
    def __init__(self, path):
        if not os.path.exists(path):
            raise SfError(
                f"native library not found: {path}. Build it with `python -m slowfast_amd.build_ext` "
                "(hipcc --offload-arch=gfx950); slowfast_amd has no CPU/PyTorch fallback.")
        self.path = path
        self.cdll = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(self.cdll, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        ver = self.cdll.sf_abi_version()
        if ver != ABI_VERSION:
            raise SfError(f"{path}: ABI version {ver}, expected {ABI_VERSION}")
        self.backend = self.cdll.sf_backend().decode()
        self.build_id = self.cdll.sf_build_id().decode()
        # build provenance: the binary carries the hash of the sources it was compiled from (build_ext.source_id() of its
        # target, passed as -DSF_BUILD_ID); a binary that is older than the csrc/ + include/ shipped beside it must not produce
        # results.  A deployment without the source tree has nothing to compare against: the check is skipped with a warning.
        from . import build_ext
        if build_ext.sources_present():
            want = build_ext.source_id(sim=self.backend == "hostsim")
            if self.build_id != want and os.environ.get("SF_ALLOW_STALE_LIBRARY", "0") == "0":
                raise SfError(f"{path} was built from other sources (build id {self.build_id}, sources {want}): rebuild with "
                              "`python -m slowfast_amd.build_ext` (SF_ALLOW_STALE_LIBRARY=1 overrides)")
        else:
            import warnings
            warnings.warn(f"{path}: kernel sources not found beside the package, build id {self.build_id} not verified")
        self.act_mode = ("fp16", "bf16")[self.cdll.sf_act_dtype()]
        if self.act_mode != ACT_MODE:
            raise SfError(f"{path} is the {self.act_mode} build of the library, this process runs with SF_ACT_DTYPE={ACT_MODE}")

    def call(self, name, *args, work=None):
        """Invoke an entry point; ``work`` = optional dict(bytes=, flops=) of algorithmic work (profiling only)."""
        fn = getattr(self.cdll, name)
        if _observer is not None:
            rc = _observer(name, lambda: fn(*args), work)
        else:
            rc = fn(*args)
        if rc < 0:
            raise SfError(f"{name}: {self.cdll.sf_last_error().decode()}")
        return rc


_lib = None
_observer = None


def set_call_observer(fn):
    """fn(name, thunk, work) -> rc wraps every native call (slowfast_amd.profiler); None disables."""
    global _observer
    _observer = fn


def get_lib():
    """The loaded native library (loads on first use; raises SfError if it cannot)."""
    global _lib
    if _lib is None:
        _lib = SfLibrary(os.environ.get("SFAMD_LIBRARY", DEFAULT_LIBRARY))
    return _lib


def reset_lib():
    global _lib
    _lib = None
