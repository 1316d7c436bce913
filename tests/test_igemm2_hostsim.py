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


@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[3], CASES[6]])
def test_igemm2_staggered_copy_issue(sim, force_v2, case, monkeypatch):
    """SF_IGEMM2_STAGGER=1: the upper four waves issue the next stage's copies between the two MFMA halves of a stage."""
    monkeypatch.setenv("SF_IGEMM2_STAGGER", "1")
    kc.check_conv_fwd(sim, *case)
    kc.check_conv_dgrad(sim, *case)


# ---- STRIP variant: one staged strip of source rows per channel chunk serves every tap (row offsets + fragment masks)
STRIP_CASES = [
    ((1, 64, 2, 9, 9), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),       # BN 64 (upper waves carry no weight copy), 2 chunks, ragged tile
    ((2, 64, 3, 12, 12), 136, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),    # 4 tiles (strips cross frame and sample borders), two N tiles
    ((1, 128, 4, 6, 6), 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),      # 3 temporal taps (delta +-36), 4 chunks: a strip issue at every third step
    ((1, 96, 1, 8, 8), 96, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),       # dilation 2 (delta up to +-18), 3 chunks
    ((1, 64, 3, 5, 5), 96, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),       # 27 taps, delta +-31
    ((2, 32, 1, 40, 40), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),     # ONE chunk (no second strip), 13 tiles, wide rows (delta +-41)
    ((1, 160, 2, 7, 7), 256, (1, 5, 5), (1, 1, 1), (0, 2, 2), (1, 1, 1)),     # 25 taps, 5 chunks, two full N tiles
]


@pytest.mark.parametrize("case", STRIP_CASES)
def test_igemm2_strip_fwd_dgrad(sim, force_v2, case, monkeypatch, capfd):
    monkeypatch.setenv("SF_IGEMM2_STRIP", "2")
    monkeypatch.setenv("SF_TRACE", "1")
    kc.check_conv_fwd(sim, *case)
    kc.check_conv_dgrad(sim, *case)
    err = capfd.readouterr().err
    # the data gradient contracts over Co (needs Co % 32 == 0) and produces Ci columns (needs Ci > 32) to run this kernel family
    want = 2 if (case[1] % 32 == 0 and case[0][1] > 32) else 1
    assert err.count("igemm2 strip") >= want, "the strip variant must be taken: " + err[-400:]


def test_igemm2_strip_epilogues(sim, force_v2, monkeypatch):
    monkeypatch.setenv("SF_IGEMM2_STRIP", "2")
    kc.check_conv_dgrad(sim, (1, 64, 2, 9, 9), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), resid=True)
    kc.check_conv_fwd_fused(sim, (1, 64, 2, 9, 9), 72, (1, 3, 3), (1, 1, 1), (0, 1, 1), resid=True, relu=True)
    kc.check_conv_fwd(sim, (1, 64, 1, 9, 9), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), ldx_extra=16)
    kc.check_conv_dgrad_bn(sim, (2, 64, 2, 9, 9), 64, (1, 3, 3), (0, 1, 1))


def test_igemm2_strip_not_taken_when_ineligible(sim, force_v2, monkeypatch, capfd):
    """Strided forward, two-tap and wide-displacement geometries keep the gather kernel."""
    monkeypatch.setenv("SF_IGEMM2_STRIP", "2")
    monkeypatch.setenv("SF_TRACE", "1")
    kc.check_conv_fwd(sim, (1, 32, 2, 10, 10), 40, (1, 3, 3), (1, 2, 2), (0, 1, 1))            # stride 2
    kc.check_conv_fwd(sim, (1, 64, 4, 12, 12), 64, (3, 1, 1), (1, 1, 1), (1, 0, 0))            # delta +-144 rows
    assert "igemm2 strip" not in capfd.readouterr().err


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


@pytest.mark.parametrize("case", [WGRAD2_CASES[0], WGRAD2_CASES[-1]])
def test_wgrad2_deep_ring(sim, force_w2, monkeypatch, case):
    """SF_WGRAD2_NST=6: one workgroup per CU with a six-stage ring (more stages in flight than a short split has steps)."""
    monkeypatch.setenv("SF_WGRAD2_NST", "6")
    monkeypatch.setenv("SF_WGRAD2_BLOCKS", "2")
    kc.check_conv_wgrad(sim, *case)
    monkeypatch.setenv("SF_WGRAD2_BLOCKS", "64")      # splits shorter than the ring
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
    ((2, 8, 3, 10, 10), 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),      # Fast res2 b: BMW 16, K 72 in one 128 tile, 600 rows
    ((1, 32, 6, 8, 8), 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),       # Fast res2 a: temporal taps, K 96
    ((2, 8, 2, 9, 9), 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),       # Fast res2 c: BMW 32, K 8 -> 32-wide tile, 3 stages
    ((1, 16, 2, 12, 12), 16, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),    # BMW 16, K 16 -> 32-wide tile
    ((1, 64, 4, 6, 6), 16, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1)),      # K 192: two 128-wide k tiles
    ((1, 16, 2, 11, 11), 24, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1)),    # stride 2, Co 24 (BMW 32, ragged), K 144
    ((1, 24, 5, 7, 7), 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),      # 27 taps x 24 channels: taps straddle chunks and tiles
    ((3, 8, 1, 7, 7), 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),        # 147 rows: splits of one ragged stage
]


@pytest.mark.parametrize("case", WGRAD2T_CASES)
def test_wgrad2_thin(sim, force_w2t, case):
    kc.check_conv_wgrad(sim, *case)


def test_wgrad2_thin_is_taken(sim, force_w2t):
    """The thin plan must actually be the one that runs (row-table bytes > 0 for a 16-channel layer)."""
    from ctypes import byref
    from slowfast_amd import ops
    from slowfast_amd.lib import get_lib
    geom = ops.ConvGeom((2, 8, 3, 10, 10), 8, (1, 3, 3), 1, (0, 1, 1))
    assert get_lib().call("sf_conv_wgrad_rowtab_bytes", byref(geom.desc(8, 8))) > 0


# ---- thin forward / data gradient (sf_igemm2t.h: <= 32 output columns, K <= 128, independent waves, weights in registers)
@pytest.fixture()
def force_thin(monkeypatch):
    monkeypatch.setenv("SF_IGEMM2T", "1")
    monkeypatch.setenv("SF_IGEMM2T_MINROWS", "1")
    monkeypatch.setenv("SF_IGEMM2T_BLOCKS", "3")      # several stages per workgroup even on tiny shapes


# (in_shape, Co, kernel, stride, pad, dilation): forward is thin when Co <= 32 and taps*Ci <= 128, the data gradient when
# Ci <= 32, taps*Co <= 128 and the stride is 1
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
def test_igemm2_thin(sim, force_thin, case):
    kc.check_conv_fwd(sim, *case)
    kc.check_conv_dgrad(sim, *case)


def test_igemm2_thin_is_taken(sim, force_thin):
    from ctypes import byref
    from slowfast_amd import ops
    from slowfast_amd.lib import get_lib
    geom = ops.ConvGeom((2, 8, 3, 10, 10), 8, (1, 3, 3), 1, (0, 1, 1))
    d = geom.desc(8, 8)
    assert get_lib().call("sf_conv_thin_rowtab_bytes", byref(d), 0) > 0 and get_lib().call("sf_conv_thin_rowtab_bytes", byref(d), 1) > 0
    assert get_lib().call("sf_conv_thin_blocks", byref(d), 0) == 3
    geom2 = ops.ConvGeom((1, 64, 2, 9, 9), 64, (1, 3, 3), 1, (0, 1, 1))
    assert get_lib().call("sf_conv_thin_rowtab_bytes", byref(geom2.desc(64, 64)), 0) == 0
