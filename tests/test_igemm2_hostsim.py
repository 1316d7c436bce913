"""CPU: the second-generation implicit GEMM (csrc/sf_igemm2.h: wave-uniform taps, direct-to-LDS gathered operands, 256-row
tiles, three-stage ring) through the host functional simulator, against F.conv3d on identical fp16-rounded operands.

The launcher only takes this kernel for deep contractions on many rows; the ``force_v2`` fixture lowers the thresholds
(environment knobs the dispatcher reads on every call) so that small shapes run it too."""
import pytest

from tests import kernel_checks as kc


@pytest.fixture()
def force_v2(monkeypatch):
    monkeypatch.setenv("SF_IGEMM2", "1")
    monkeypatch.setenv("SF_IGEMM2_MINK", "32")
    monkeypatch.setenv("SF_IGEMM2_MINROWS", "1")

CASES = [
    # in_shape (N,Ci,T,H,W), Co, kernel, stride, pad, dil
    ((1, 64, 2, 9, 9), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),       # BK 64, 9 taps, M=162 (one ragged tile), BN 64
    ((2, 64, 3, 12, 12), 136, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),    # M=864 (4 tiles), two N tiles, ragged N
    ((1, 128, 4, 6, 6), 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),      # temporal taps, two channel chunks per tap
    ((1, 32, 2, 10, 10), 40, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1)),     # BK 32, stride 2 (forward only takes it)
    ((1, 96, 1, 8, 8), 72, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),       # BK 32 (96 = 3 x 32), dilation 2
    ((1, 64, 8, 4, 4), 128, (7, 1, 1), (4, 1, 1), (3, 0, 0), (1, 1, 1)),      # lateral: 7 temporal taps, stride 4
    ((2, 192, 1, 20, 20), 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),   # plain GEMM, K=192 (3 steps), 800 rows
    ((1, 64, 3, 5, 5), 96, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),       # 27 taps
]


@pytest.mark.parametrize("case", CASES)
def test_igemm2_fwd(sim, force_v2, case):
    kc.check_conv_fwd(sim, *case)


@pytest.mark.parametrize("case", CASES)
def test_igemm2_dgrad(sim, force_v2, case):
    kc.check_conv_dgrad(sim, *case)


def test_igemm2_dgrad_residual(sim, force_v2):
    kc.check_conv_dgrad(sim, (1, 64, 2, 9, 9), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), resid=True)


def test_igemm2_fused_epilogue(sim, force_v2):
    kc.check_conv_fwd_fused(sim, (1, 64, 2, 9, 9), 72, (1, 3, 3), (1, 1, 1), (0, 1, 1), resid=True, relu=True)


def test_igemm2_channel_slice_input(sim, force_v2):
    kc.check_conv_fwd(sim, (1, 64, 1, 9, 9), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), ldx_extra=16)


# ---- stride-1 multi-tap shapes whose tiles cross frame / sample borders (written for the round-3 STRIP experiment, kept as
# gather-kernel cases)
BORDER_CASES = [
    ((1, 64, 2, 9, 9), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),       # BN 64, 2 chunks, ragged tile
    ((2, 64, 3, 12, 12), 136, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),    # 4 tiles crossing frame and sample borders, two N tiles
    ((1, 128, 4, 6, 6), 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),      # 3 temporal taps (row delta +-36), 4 chunks
    ((1, 96, 1, 8, 8), 96, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),       # dilation 2 (delta up to +-18), 3 chunks
    ((1, 64, 3, 5, 5), 96, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),       # 27 taps, delta +-31
    ((2, 32, 1, 40, 40), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),     # ONE chunk, 13 tiles, wide rows (row delta +-41)
    ((1, 160, 2, 7, 7), 256, (1, 5, 5), (1, 1, 1), (0, 2, 2), (1, 1, 1)),     # 25 taps, 5 chunks, two full N tiles
]


@pytest.mark.parametrize("case", BORDER_CASES)
def test_igemm2_border_crossing_tiles(sim, force_v2, case):
    kc.check_conv_fwd(sim, *case)
    kc.check_conv_dgrad(sim, *case)


# strided data gradients: one launch per stride-residue class (sf_api.hip: try_igemm2_strided_dgrad)
STRIDED = [
    ((1, 64, 2, 10, 10), 64, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1)),     # 3x3 stride 2: classes with 1 / 2 / 2 / 4 taps
    ((1, 32, 2, 9, 9), 64, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 1, 1)),       # 1x1 stride 2, odd extent: three tap-less classes
    ((1, 32, 9, 4, 4), 64, (7, 1, 1), (4, 1, 1), (3, 0, 0), (1, 1, 1)),       # lateral connection: temporal stride 4
    ((1, 32, 1, 11, 11), 32, (1, 3, 3), (1, 2, 2), (0, 2, 2), (1, 2, 2)),     # stride 2 with dilation 2: one class owns all taps
    ((2, 24, 3, 8, 8), 96, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1)),       # strides on all three axes, Ci = 24 (BN 32 tile)
]


@pytest.mark.parametrize("case", STRIDED)
def test_igemm2_strided_dgrad(sim, force_v2, case):
    kc.check_conv_dgrad(sim, *case)


def test_igemm2_strided_dgrad_residual(sim, force_v2):
    kc.check_conv_dgrad(sim, (1, 32, 9, 4, 4), 64, (7, 1, 1), (4, 1, 1), (3, 0, 0), resid=True)
    kc.check_conv_dgrad(sim, (1, 32, 2, 9, 9), 64, (1, 1, 1), (1, 2, 2), (0, 0, 0), resid=True)


# ---- second-generation weight gradient (csrc/sf_wgrad2.h: row table, direct-to-LDS operands, transpose reads)
@pytest.fixture()
def force_w2(monkeypatch):
    monkeypatch.setenv("SF_WGRAD2", "1")
    monkeypatch.setenv("SF_WGRAD2_MINK", "32")
    monkeypatch.setenv("SF_WGRAD2_MINROWS", "1")
    monkeypatch.setenv("SF_WGRAD2_BLOCKS", "6")       # several splits even on tiny shapes


WGRAD2_CASES = [
    ((1, 64, 2, 9, 9), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),       # BMW 64, K 576 = 2.25 tiles (partial last tile)
    ((2, 64, 3, 12, 12), 136, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),    # BMW 128, two co tiles (ragged), 4 taps per k tile
    ((1, 128, 4, 6, 6), 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),      # temporal taps, 2 taps per k tile
    ((1, 32, 2, 10, 10), 40, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1)),     # stride 2, 32 channels: 8 taps in the first k tile
    ((1, 96, 1, 8, 8), 72, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),       # dilation 2, taps straddle k-tile boundaries (96 ch)
    ((1, 64, 8, 4, 4), 128, (7, 1, 1), (4, 1, 1), (3, 0, 0), (1, 1, 1)),      # lateral: 7 temporal taps, stride 4
    ((2, 320, 1, 20, 20), 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),   # plain GEMM K 320, 800 rows
    ((1, 40, 3, 5, 5), 96, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),       # 27 taps, 40 channels (taps split inside 16-byte-chunk runs)
    ((1, 24, 2, 7, 7), 48, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),       # K 216 < one tile
]


@pytest.mark.parametrize("case", WGRAD2_CASES)
def test_wgrad2(sim, force_w2, case):
    kc.check_conv_wgrad(sim, *case)


# two splits per 1024-thread workgroup (sf_wgrad2_kernel<., true>): both co-tile heights, even and odd split counts (the last
# pair's second half has no rows), a second half shorter than the first
WGRAD2_DUAL_CASES = [
    (((2, 64, 3, 12, 12), 136, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)), 24),      # BMW 128, 6 tiles x 4 splits of 224 rows (864)
    (((2, 64, 3, 12, 12), 136, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)), 30),      # 5 splits: three pairs, the last one half empty
    (((2, 320, 1, 20, 20), 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)), 12),     # plain GEMM, 4 tiles x 3 splits (288 / 288 / 224)
    (((1, 64, 2, 9, 9), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)), 15),         # BMW 64, 5 splits of 64 rows (162: last one 34)
]


@pytest.mark.parametrize("case,blocks", WGRAD2_DUAL_CASES)
def test_wgrad2_two_splits_per_workgroup(sim, force_w2, monkeypatch, case, blocks):
    monkeypatch.setenv("SF_WGRAD2_BLOCKS", str(blocks))
    kc.check_conv_wgrad(sim, *case)
    monkeypatch.setenv("SF_WGRAD2_DUAL", "0")           # the one-split-per-workgroup kernel on the same plan
    kc.check_conv_wgrad(sim, *case)


def test_wgrad2_accumulate_and_scale(sim, force_w2):
    kc.check_conv_wgrad(sim, (1, 64, 2, 9, 9), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), out_scale=0.25)


def test_wgrad2_row_table_is_built_once_per_geometry(sim, force_w2):
    """ops.conv_wgrad keeps the {first input position, tap mask} row table per (device, geometry): the second layer of the
    same shape -- and every later call -- passes the cached table instead of rebuilding it; same result as the per-call build
    (the plain check above runs the first call of a geometry)."""
    import torch
    from slowfast_amd import lib, ops
    counts = {}

    def observer(name, thunk, work):
        counts[name] = counts.get(name, 0) + 1
        return thunk()
    ops._rowtabs.clear()
    g = torch.Generator().manual_seed(3)
    shape = (1, 64, 2, 9, 9)
    geoms = [ops.ConvGeom(shape, co, (1, 3, 3), 1, (0, 1, 1)) for co in (64, 72)]
    lib.set_call_observer(observer)
    try:
        outs = []
        for rep in range(2):
            for geom in geoms:
                x = kc.host_to_cl(torch.randn(shape, generator=torch.Generator().manual_seed(5)), sim)
                dy = kc.host_to_cl(torch.randn((1, geom.Cow, 2, 9, 9), generator=torch.Generator().manual_seed(6)), sim)
                dw = torch.zeros((geom.Cow, 64, 1, 3, 3), dtype=torch.float32)
                outs.append(ops.conv_wgrad(x, dy, geom, dw).clone())
    finally:
        lib.set_call_observer(None)
    assert counts["sf_conv_wgrad"] == 4 and counts["sf_conv_wgrad_rowtab"] == 1
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3])
    ref = torch.nn.grad.conv3d_weight(
        kc.cl_to_host(kc.host_to_cl(torch.randn(shape, generator=torch.Generator().manual_seed(5)), sim)).float(), (64, 64, 1, 3, 3),
        kc.cl_to_host(kc.host_to_cl(torch.randn((1, 64, 2, 9, 9), generator=torch.Generator().manual_seed(6)), sim)).float(),
        padding=(0, 1, 1))
    assert float((outs[0] - ref).abs().max() / ref.abs().max()) < 2e-3


# ---- thin weight gradient (sf_wgrad2t_kernel: <= 32 output channels, waves split the positions of a stage)
@pytest.fixture()
def force_w2t(monkeypatch, force_w2):
    monkeypatch.setenv("SF_WGRAD2T", "1")
    monkeypatch.setenv("SF_WGRAD2T_MINROWS", "1")
    monkeypatch.setenv("SF_WGRAD2T_BLOCKS", "5")


WGRAD2T_CASES = [
    ((2, 16, 3, 10, 10), 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),    # Fast res3 b: BMW 16, K 144 (two 128-wide tiles), 600 rows
                                                                             # (the 8-channel res2 b takes sf_stem.h since round 3)
    ((1, 32, 6, 8, 8), 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),       # Fast res2 a: temporal taps, K 96
    ((2, 8, 2, 9, 9), 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),       # Fast res2 c: BMW 32, K 8 -> 32-wide tile, 3 stages
    ((1, 16, 2, 12, 12), 16, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),    # BMW 16, K 16 -> 32-wide tile
    ((1, 64, 4, 6, 6), 16, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),      # K 192: two 128-wide k tiles
    ((1, 16, 2, 11, 11), 24, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1)),    # stride 2, Co 24 (BMW 32, ragged), K 144
    ((1, 24, 5, 7, 7), 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),      # 27 taps x 24 channels: taps straddle chunks and tiles
    ((3, 8, 1, 7, 7), 8, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),        # 147 rows: splits of one ragged stage (dilation 2)
]


@pytest.mark.parametrize("case", WGRAD2T_CASES)
def test_wgrad2_thin(sim, force_w2t, case):
    kc.check_conv_wgrad(sim, *case)


def test_wgrad2_thin_is_taken(sim, force_w2t):
    """The thin plan must actually be the one that runs (row-table bytes > 0 for a 16-channel layer)."""
    from ctypes import byref
    from slowfast_amd import ops
    from slowfast_amd.lib import get_lib
    geom = ops.ConvGeom((2, 16, 3, 10, 10), 16, (1, 3, 3), 1, (0, 1, 1))
    assert get_lib().call("sf_conv_wgrad_rowtab_bytes", byref(geom.desc(16, 16))) > 0


# ---- thin layers (<= 32 output columns, K <= 128: the Fast pathway's shapes) through the general kernels
# (in_shape, Co, kernel, stride, pad, dilation)
THIN_CASES = [
    ((2, 8, 3, 10, 10), 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),      # Fast res2 b: BN 16, K 72 (128-wide slices), 600 rows
    ((1, 32, 6, 8, 8), 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),       # Fast res2 a: fwd K 96 / dgrad K 24 (32-wide), BN 32
    ((2, 8, 2, 9, 9), 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),       # Fast res2 c: fwd BN 32 K 8 / dgrad BN 16 K 32
    ((1, 16, 2, 12, 12), 16, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),    # K 16, 288 rows: ragged last slice
    ((1, 8, 2, 11, 11), 24, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1)),     # stride 2: forward only (Co 24: BN 32, ragged columns)
    ((3, 8, 1, 7, 7), 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),        # 147 rows: two stages, mostly padding taps
    ((1, 8, 4, 6, 6), 16, (3, 3, 1), (1, 1, 1), (1, 1, 0), (1, 1, 1)),       # 9 taps over T and H
    ((1, 16, 2, 9, 9), 8, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),       # dilation 2; fwd K 144 is NOT thin, dgrad K 72 is
]


@pytest.mark.parametrize("case", THIN_CASES)
def test_thin_layers(sim, case):
    kc.check_conv_fwd(sim, *case)
    kc.check_conv_dgrad(sim, *case)

