"""CPU: every token-space kernel (MViT path) through the host simulator against the torch fp32 reference ops."""
import pytest

from tests import token_checks as tc


def test_gemm_linear(sim):
    tc.check_gemm(sim, 200, 96, 288)
    tc.check_gemm(sim, 77, 32, 24, bias=False, resid=False)
    tc.check_gemm(sim, 130, 192, 8, bias=True, resid=False)


def test_rows32_side_rows(sim, monkeypatch):
    tc.check_rows32(sim, 3, 50, 96, 192)                               # class-token rows, round-1 GEMM kernel
    tc.check_rows32(sim, 2, 50, 64, 72, bias=False, src32=False)       # residual rows from the 16-bit operand, narrow N
    tc.check_rows32(sim, 2, 9, 32, 96, period_full=True)               # every row (last stage)
    tc.check_rows32(sim, 2, 150, 64, 160, igemm2=True, monkeypatch=monkeypatch)      # second-generation kernel (256-row tiles)
    tc.check_rows32(sim, 2, 20, 64, 128, period_full=True, igemm2=True, monkeypatch=monkeypatch)
    monkeypatch.setenv("SF_IGEMM2_T128", "2")                          # the 128 x 128 tile of the same kernel (plain matrix products)
    tc.check_rows32(sim, 2, 150, 64, 160, igemm2=True, monkeypatch=monkeypatch)
    tc.check_rows32(sim, 3, 47, 96, 136, src32=False, igemm2=True, monkeypatch=monkeypatch)


def test_layernorm(sim):
    tc.check_layernorm(sim, 300, 96)
    tc.check_layernorm(sim, 37, 32)
    tc.check_layernorm(sim, 50, 192)
    tc.check_layernorm(sim, 21, 384)
    tc.check_layernorm(sim, 19, 768)
    tc.check_layernorm(sim, 25000 + 5, 96)        # several passes per workgroup, ragged last one (768 workgroups x 16 rows x 2)


def test_gelu(sim):
    tc.check_gelu(sim, 4096)


def test_dwconv_tokens(sim):
    _dwconv_tokens(sim)


def test_dwconv_tokens_stencils(sim, monkeypatch):
    """The same geometries on the W-blocked / generic stencils the ring sweep (sf_dwsweep.h) replaced for 3x3x3, stride <= 2."""
    monkeypatch.setenv("SF_DW_SWEEP", "0")
    _dwconv_tokens(sim)


def test_dwconv_temporal(sim, monkeypatch):
    """(kT, 1, 1) stride-1 depthwise convolutions on the register-window walk (sf_dwtemporal.h; X3D stem conv_t): kT = 3 / 5 / 7,
    clips shorter than the window and its look-ahead, heads sharing the weights, a channel count that leaves idle lanes (24 = 3
    groups: 85 positions per pass), several passes per workgroup, and the stencils it replaces on the same cases."""
    cases = [(2, 1, 24, (6, 4, 4), (5, 1, 1)), (1, 1, 16, (1, 3, 3), (5, 1, 1)), (1, 1, 16, (2, 3, 3), (7, 1, 1)),
             (1, 2, 8, (3, 5, 5), (3, 1, 1)), (2, 1, 24, (9, 12, 12), (5, 1, 1)), (1, 1, 40, (4, 6, 6), (7, 1, 1))]
    for B, heads, Cw, thw, k in cases:
        tc.check_dwconv(sim, B, heads, Cw, thw, k, (1, 1, 1), cls=0)
    monkeypatch.setenv("SF_DWTP_BLOCKS", "2")                   # 288 positions / 85 per pass = 4 passes on 2 workgroups
    tc.check_dwconv(sim, 2, 1, 24, (9, 12, 12), (5, 1, 1), (1, 1, 1), cls=0)
    monkeypatch.delenv("SF_DWTP_BLOCKS")
    monkeypatch.setenv("SF_DW_TEMPORAL", "0")
    tc.check_dwconv(sim, 2, 1, 24, (6, 4, 4), (5, 1, 1), (1, 1, 1), cls=0)


def test_dwconv_dgrad_leaves_column_sums(sim):
    """sf_dwconv_dgrad_sums: the plane sweeps (stride 1: rotating body, stride 2: 2 x 2 block body) leave the column sums of dx, cls
    row included -- what MViT's qkv bias gradient is taken from; strides >= 3 and other windows report no table (the caller falls
    back to a pass).  The values are checked inside tc.check_dwconv for every geometry that has a table."""
    import torch
    from ctypes import byref
    from slowfast_amd import lib, tokens
    for stride, want in (((1, 1, 1), True), ((1, 2, 2), True), ((1, 4, 4), False)):
        geom = tokens.DwGeom(2, 64, 32, (2, 8, 8), (3, 3, 3), stride, (1, 1, 1), 1)
        rows = lib.get_lib().call("sf_dwconv_dgrad_sum_rows", byref(geom.desc(192, 64)))
        assert (rows > 0) == want, (stride, rows)
    tc.check_dwconv(sim, 2, 2, 32, (2, 8, 8), (3, 3, 3), (1, 2, 2), cls=1)
    tc.check_dwconv(sim, 1, 2, 32, (3, 7, 7), (3, 3, 3), (1, 1, 1), cls=1)


def test_dwconv_pair(sim, monkeypatch):
    """pool_k / pool_v in one launch per direction (sf_dwconv_*_pair; PAIR mode of the plane sweeps): strides 1 / 2 (the rotating-
    accumulator body and the stride-2 data-gradient body), two heads sharing each weight, cls rows, a forced ragged tiling."""
    tc.check_dwconv_pair(sim, 2, 1, 32, (3, 6, 6), (1, 2, 2), cls=1)
    tc.check_dwconv_pair(sim, 1, 2, 32, (2, 5, 5), (1, 1, 1), cls=1)
    tc.check_dwconv_pair(sim, 1, 1, 96, (2, 7, 7), (1, 2, 2), cls=0)
    monkeypatch.setenv("SF_DWR_SL", "4")
    monkeypatch.setenv("SF_DWR_NGRP", "1")
    tc.check_dwconv_pair(sim, 1, 1, 64, (2, 9, 9), (1, 1, 1), cls=1)


def _dwconv_tokens(sim):
    tc.check_dwconv(sim, 2, 2, 16, (2, 6, 6), (3, 3, 3), (1, 2, 2), cls=1)
    tc.check_dwconv(sim, 1, 1, 32, (4, 5, 5), (3, 3, 3), (1, 1, 1), cls=1)
    tc.check_dwconv(sim, 2, 1, 24, (6, 4, 4), (5, 1, 1), (1, 1, 1), cls=0)
    tc.check_dwconv(sim, 1, 1, 16, (2, 9, 9), (3, 3, 3), (1, 4, 4), cls=1)      # generic (non-blocked) kernels
    tc.check_dwconv(sim, 1, 2, 8, (2, 7, 7), (3, 3, 3), (1, 2, 2), cls=0)       # odd width, stride 2
    # rows that are a whole number of 4-column groups (only column 0 of a group can fall outside the row)
    tc.check_dwconv(sim, 1, 1, 16, (3, 5, 8), (3, 3, 3), (1, 1, 1), cls=0)
    tc.check_dwconv(sim, 1, 1, 8, (2, 6, 16), (3, 3, 3), (1, 2, 2), cls=1)
    tc.check_dwconv(sim, 1, 1, 120, (2, 4, 8), (3, 3, 3), (1, 1, 1), cls=0)     # taps * Cw > 3072: fp16 LDS weights
    tc.check_dwconv(sim, 1, 1, 16, (5, 4, 4), (5, 1, 1), (1, 1, 1), cls=0)      # X3D stem temporal conv, whole groups
    # MViTv1 stride+1 pooling kernels (configs/Kinetics/MVIT_B_32x3_CONV.yaml): more than 9 taps per plane
    tc.check_dwconv(sim, 1, 2, 8, (2, 9, 9), (1, 5, 5), (1, 4, 4), cls=1)
    tc.check_dwconv(sim, 1, 1, 16, (2, 17, 17), (1, 9, 9), (1, 8, 8), cls=1)


def test_dwconv_ring_sweep(sim, monkeypatch):
    """Ring-buffered plane sweep with the channels on the lanes (sf_dwsweep.h, round 6): forward (stride 1 | 2, with and without the
    BatchNorm partial sums), both data gradients, the weight gradient; whole and ragged tiles in H and W, several row groups and
    column segments per wave, 8- / 16- / 24-channel tail chunks, heads sharing a weight, T = 1, odd extents.  (SF_DW_ROT=0: the
    four-channel v_fma_mix body; by default only its stride-2 data gradient is dispatched.)"""
    monkeypatch.setenv("SF_DW_ROT", "0")
    tc.check_dwconv(sim, 1, 1, 32, (2, 14, 14), (3, 3, 3), (1, 1, 1), cls=1)     # MViT stage-3 plane: two row groups x two segments
    tc.check_dwconv(sim, 1, 1, 64, (3, 14, 14), (3, 3, 3), (1, 2, 2), cls=1)     # 14 -> 7, two chunks
    tc.check_dwconv(sim, 1, 2, 32, (2, 7, 9), (3, 3, 3), (1, 2, 2), cls=0)       # odd extents: 7x9 -> 4x5, no cls (partial sums ride)
    tc.check_dwconv(sim, 1, 1, 56, (2, 9, 10), (3, 3, 3), (1, 1, 1), cls=0)      # X3D width 54 -> 56: a 32- and a 24-channel chunk
    tc.check_dwconv(sim, 1, 1, 40, (1, 6, 11), (3, 3, 3), (1, 2, 2), cls=0)      # 8-channel tail chunk, T = 1
    monkeypatch.setenv("SF_DWS_TH", "5")                                         # ragged row tiles (14 = 5 + 5 + 4), ...
    monkeypatch.setenv("SF_DWS_TW", "6")                                         # ... ragged column tiles (14 = 6 + 6 + 2)
    tc.check_dwconv(sim, 1, 1, 32, (3, 14, 14), (3, 3, 3), (1, 1, 1), cls=1)
    tc.check_dwconv(sim, 1, 1, 32, (2, 14, 14), (3, 3, 3), (1, 2, 2), cls=0)
    monkeypatch.setenv("SF_DWS_TH", "11")                                        # two row groups, the second one partial
    monkeypatch.setenv("SF_DWS_TW", "20")
    monkeypatch.setenv("SF_DWS_NSEG", "3")                                       # segments of 7, 7, 6 columns (window rotation tails)
    tc.check_dwconv(sim, 1, 1, 16, (2, 12, 20), (3, 3, 3), (1, 1, 1), cls=1)
    monkeypatch.setenv("SF_DWS_NSEG", "1")                                       # one segment: tasks < waves
    tc.check_dwconv(sim, 1, 1, 16, (2, 12, 20), (3, 3, 3), (1, 2, 2), cls=1)


def test_dwconv_rotating_sweep(sim, monkeypatch):
    """Rotating-accumulator form (sf_dwrot_kernel): runs of 7 and 4 columns, every (row groups x segments) split of the four waves,
    ragged tiles, T = 1 / 2 / 3 / 4 / 5 (the plane loop is unrolled by three), tail chunks, heads sharing a weight, both strides,
    forward with and without partial sums, stride-1 data gradient, weight gradient."""
    for sl, grp, seg in ((7, 2, 2), (7, 4, 1), (7, 1, 4), (4, 2, 2), (4, 1, 3), (7, 1, 1), (4, 3, 1)):
        monkeypatch.setenv("SF_DWR_SL", str(sl))
        monkeypatch.setenv("SF_DWR_NGRP", str(grp))
        monkeypatch.setenv("SF_DWR_NSEG", str(seg))
        tc.check_dwconv(sim, 1, 1, 32, (3, 14, 14), (3, 3, 3), (1, 1, 1), cls=1)
        tc.check_dwconv(sim, 1, 1, 40, (2, 9, 10), (3, 3, 3), (1, 2, 2), cls=0)
    for k in ("SF_DWR_SL", "SF_DWR_NGRP", "SF_DWR_NSEG"):
        monkeypatch.delenv(k)
    for T in (1, 2, 3, 4, 5):
        tc.check_dwconv(sim, 1, 2, 16, (T, 6, 9), (3, 3, 3), (1, 1, 1), cls=T % 2)
        tc.check_dwconv(sim, 1, 1, 24, (T, 7, 7), (3, 3, 3), (1, 2, 2), cls=0)
    tc.check_dwconv(sim, 2, 1, 56, (2, 9, 10), (3, 3, 3), (1, 1, 1), cls=0)      # X3D width 54 -> 56: a 32- and a 24-channel chunk
    # strides >= 3 (MViT k / v pooling of the early stages): packed staging + the one-tap data gradient
    tc.check_dwconv(sim, 1, 1, 32, (3, 17, 17), (3, 3, 3), (1, 4, 4), cls=1)
    tc.check_dwconv(sim, 2, 2, 16, (2, 16, 24), (3, 3, 3), (1, 8, 8), cls=1)
    tc.check_dwconv(sim, 1, 1, 24, (4, 10, 13), (3, 3, 3), (1, 3, 3), cls=0)
    monkeypatch.setenv("SF_DWR_SL", "7")
    tc.check_dwconv(sim, 1, 1, 32, (2, 30, 30), (3, 3, 3), (1, 4, 4), cls=1)


def test_dwconv_tiled_plane_sweep(sim, monkeypatch):
    """LDS-tiled plane sweep (sf_dwtile.h): 32-channel chunks, strides 1 and 2 (forward, data gradient incl. the zero-upsampled
    stride-2 form, weight gradient in its three LDS classes), partial last row tiles, odd extents, 1 / 2 / 4 positions per thread."""
    monkeypatch.setenv("SF_DW_SWEEP", "0")      # the round-6 ring sweep would take every one of these geometries
    monkeypatch.setenv("SF_DW_TILED", "2")      # also the stride-2 forms (the library's statics are read on first use: the
    # fixture loads a fresh library handle per test, the env is read again)
    tc.check_dwconv(sim, 2, 2, 32, (3, 6, 6), (3, 3, 3), (1, 1, 1), cls=1)       # one tile, NP = 1
    tc.check_dwconv(sim, 1, 1, 32, (2, 14, 14), (3, 3, 3), (1, 1, 1), cls=1)     # MViT stage-3 plane, NP = 2 / 4
    tc.check_dwconv(sim, 1, 1, 64, (3, 14, 14), (3, 3, 3), (1, 2, 2), cls=1)     # 14 -> 7, two chunks
    tc.check_dwconv(sim, 1, 2, 32, (2, 7, 9), (3, 3, 3), (1, 2, 2), cls=0)       # odd extents: 7x9 -> 4x5, no cls
    tc.check_dwconv(sim, 1, 1, 32, (1, 5, 30), (3, 3, 3), (1, 1, 1), cls=1)      # wide rows: several row tiles, T = 1
    tc.check_dwconv(sim, 1, 1, 96, (4, 12, 12), (3, 3, 3), (1, 2, 2), cls=1)     # head width 96 (three chunks of one weight group)
    monkeypatch.setenv("SF_DWT_TH", "13")                                        # weight gradient: the 85 KiB class (182 positions
    tc.check_dwconv(sim, 1, 1, 32, (2, 14, 14), (3, 3, 3), (1, 1, 1), cls=1)     # per dy slot), partial second tile
    monkeypatch.delenv("SF_DWT_TH")
    monkeypatch.setenv("SF_DW_WGRAD_TILED", "0")                                 # ... and the stencil it replaces, same case
    tc.check_dwconv(sim, 1, 1, 32, (2, 14, 14), (3, 3, 3), (1, 1, 1), cls=1)
    monkeypatch.delenv("SF_DW_WGRAD_TILED")
    monkeypatch.setenv("SF_DWT_TH", "7")                                         # 98 positions per tile: 2 per thread
    tc.check_dwconv(sim, 1, 1, 32, (2, 14, 14), (3, 3, 3), (1, 1, 1), cls=1)
    monkeypatch.setenv("SF_DWT_TH", "14")                                        # 196 positions: 4 per thread (3 rounds up)
    tc.check_dwconv(sim, 1, 1, 32, (2, 14, 14), (3, 3, 3), (1, 1, 1), cls=1)
    tc.check_dwconv(sim, 1, 1, 32, (2, 14, 14), (3, 3, 3), (1, 2, 2), cls=1)


def test_token_pool(sim):
    tc.check_token_pool(sim, 2, 16, (2, 6, 6), (1, 2, 2))
    tc.check_token_pool(sim, 1, 8, (3, 5, 7), (1, 2, 2))


def test_attention_core(sim):
    tc.check_attention_core(sim, 2, 2, 32, (2, 4, 4), (2, 2, 2))
    tc.check_attention_core(sim, 1, 1, 96, (2, 3, 3), (2, 3, 3))


@pytest.mark.parametrize("case", [
    (1, 1, 32, (2, 3, 3), (2, 3, 3), True, True, True),        # one partial key chunk, cls, rel-pos, residual pooling
    (2, 2, 32, (2, 6, 6), (2, 3, 3), True, True, True),        # 73 queries (two query tiles), two heads
    (1, 2, 96, (2, 8, 8), (2, 4, 4), True, True, True),        # MViTv2 head dim, 33 keys (two key chunks)
    (1, 1, 64, (1, 5, 9), (1, 5, 9), False, False, False),     # no cls / no rel-pos / no residual
    (1, 1, 32, (2, 16, 16), (2, 3, 3), True, True, True),      # 513 queries: the dK/dV kernel splits them in two
    (1, 1, 32, (4, 5, 5), (4, 15, 15), True, True, True),      # kH + kW + kT = 34 > 32: second bias K-step
])
def test_attention_fused(sim, case):
    tc.check_attention_fused(sim, *case)


def test_gemm_gelu_epilogues(sim):
    tc.check_gemm_gelu(sim, 200, 96, 384)
    tc.check_gemm_gelu(sim, 77, 32, 128, seed=1)
