// Direct convolution for the thin RGB stems (forward + weight gradient), LDS-patch based -- and, since round 3, for 8-channel
// (kT, kH, 3) stride-1 layers (the Fast pathway's res2 1x3x3 bottleneck: forward, data gradient with the fused BatchNorm-backward
// sums of sf_conv_dgrad_bn, weight gradient), whose pixel IS one 16-byte chunk: three real taps and a zero fourth one per slice.
//
// Reference call sites: slowfast/models/stem_helper.py:182-189 (ResNetBasicStem.conv, the Fast pathway's
// Conv3d(3, 8, [5,7,7], stride [1,2,2], padding [2,3,3])), slowfast/models/resnet_helper.py:331-369 (BottleneckTransform.b of the
// Fast pathway's res2 stage) and their autograd backward.
//
// The generic implicit GEMM (sf_igemm.h) gathers every im2col element with its own global load: for the Fast
// stem that is M*K*2 B = 12.8M x 1120 x 2 = 28.7 GB through the vector L1 per pass, although the clip itself is
// only 0.41 GB.  Here a workgroup stages the input PATCH of its output tile in LDS once and the MFMA operands are
// read straight out of that patch -- no im2col matrix exists anywhere:
//
//   * the clip is the W-pair view N,T,H,W/2,8 (engine.StemConvUnit): one 16-byte chunk = 2 pixels x 4 channels,
//     the kernel is (kT, kH, 4 pairs) with pair stride 1, pair padding 2;
//   * for a fixed (kt, kh) the K-slice of an output pixel (4 pairs x 8 = 32 halfs) is CONTIGUOUS in the patch row
//     and the slices of neighbouring pixels overlap, 16 bytes apart: a 16x16x32 MFMA operand is one ds_read_b128
//     per lane at  patch[frame][row][pixel + lane/16];
//   * forward:  D[co][pixel] += W[co][slice] * patch-slice[pixel]      (A = weights, B = pixels)
//     wgrad:    D[co][k]     += dy^T[co][pixel] * patch-slice[pixel][k] (32 pixels reduced per MFMA; both operands
//               come from ds_read_b64_tr_b16 with per-lane row addresses, the "rows" being overlapping windows).
//
// Tile: 4 output frames x 8 rows x 16 pixels per workgroup; patch = (3 sT + kT) x (7 sH + kH) x 19 chunks.
#pragma once
#include "sf_common.h"

#define SF_STEM_TT 4
#define SF_STEM_TH 8
#define SF_STEM_TW 16
#define SF_STEM_PC (SF_STEM_TW + 3)
#define SF_STEM_CHUNKS 3328          // 16-byte chunks of LDS for the patch (52 KiB): the stems
#define SF_STEM_CHUNKS_SMALL 768     // 12 KiB: patches of the 8-channel (1, 3, 3) layers (4 x 10 x 19 chunks) -- the 52 KiB
                                     // allocation held them at three workgroups per CU, and they are latency-bound
#define SF_STEM_WG_THREADS 512
#define SF_STEM_MAX_SLICES 40        // kT * kH

struct StemParams {
    const f16* x; int ldx;            // W-pair view rows (n, t, h, w2), 8 channels
    int N, Ti, Hi, Wi;                // Wi counts pairs
    int To, Ho, Wo, Co;
    int kT, kH, sT, sH, pT, pH;       // kW = 4 chunks (the last may be a zero tap), sW = 1 chunk
    int pW;                           // chunks of left padding: 2 pairs (stems), 1 pixel (8-channel 1x3x3 layers)
    // weight of (slice s = kt*kH + kh, chunk g): wmat[co][wo0 + s*wos + g*wog .. +8]; chunks g >= kwc carry no weight.
    // stems: (0, 32, 8, 4); 8-channel (kT,3,3) forward on the packed forward operand: (0, 24, 8, 3)
    int wo0, wos, wog, kwc;
    const f16* wmat; int ldw;         // [Co][ldw], k = (kt*kH + kh)*32 + pair*8 + e
    f16* y; int ldy;
    float* stat_part; int stat_rows;  // [stat_rows][2][Co]; rows beyond the workgroup count are zeroed here
    const float* bias; int out_relu;  // inference-fused forward (sf_conv_fwd_fused): y = relu?(conv + bias[co])
    // data-gradient use (thin3): the BatchNorm-backward partial sums of sf_conv_dgrad_bn instead of the forward statistics --
    // stat_part rows become [2][Co] = {sum g, sum g * bnb_y}, g = stored output masked by (bnb_y * bnb_scale + bnb_shift > 0)
    const f16* bnb_y; int bnb_ld; const float* bnb_scale; const float* bnb_shift;
    const uint8_t* bnb_bits;          // optional [rows][Co/8] bit mask replacing the recomputed one (a block output's ReLU)
    int tiles_w, tiles_h, tiles_t, ntiles;
    FastDiv fd_tw, fd_th, fd_tt;
    int F, PR;                        // patch frames / rows
    FastDiv fd_pc, fd_prpc;
    // weight gradient
    const f16* dy;
    float* ws; int Kpad;              // per-workgroup slab [16][Kpad] fp32
    int tiles_per_block;
    int wgroups;                      // > 1 (few slices): wave w works on slice w % nsl for the tile rows of group w / nsl; one slab per group
    int plain_order;                  // weight gradient: 1 = workgroup b takes tile run b (SF_STEM_XCD=0, A/B); 0 = XCD-contiguous
    int seg_groups;                   // sliding forward: groups of SF_STEM_TT output frames a workgroup walks (tiles_t counts the runs)
};

struct StemTile { int n, t0, h0, w0; };
__device__ __forceinline__ StemTile stem_tile(const StemParams& p, uint32_t tile) {
    StemTile t;
    uint32_t a = fd_div(tile, p.fd_tw);
    t.w0 = (int)(tile - a * (uint32_t)p.tiles_w) * SF_STEM_TW;
    uint32_t b = fd_div(a, p.fd_th);
    t.h0 = (int)(a - b * (uint32_t)p.tiles_h) * SF_STEM_TH;
    uint32_t c = fd_div(b, p.fd_tt);
    t.t0 = (int)(b - c * (uint32_t)p.tiles_t) * SF_STEM_TT;
    t.n = (int)c;
    return t;
}

// stage the input patch of one tile: chunk (f, r, c) <- x[n][t0*sT - pT + f][h0*sH - pH + r][w0 - pW + c], zeros outside
template <int NTHREADS, int U>      // U: 16-byte loads in flight per thread (the Fast stem's patch is 12.5 / 6.2 per thread -> 7; a
                                     // 768-chunk patch is 3 / 1.5 per thread)
__device__ __forceinline__ void stem_load_patch(const StemParams& p, f16* patch, const StemTile& t, int tid) {
    const int nch = p.F * p.PR * SF_STEM_PC;
    const int tin0 = t.t0 * p.sT - p.pT, hin0 = t.h0 * p.sH - p.pH, win0 = t.w0 - p.pW;
    for (int base = 0; base < nch; base += NTHREADS * U) {
        f16x8 v[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t c = (uint32_t)(base + tid + NTHREADS * u);
            const uint32_t f = fd_div(c, p.fd_prpc);
            const uint32_t rem = c - f * (uint32_t)(p.PR * SF_STEM_PC);
            const uint32_t r = fd_div(rem, p.fd_pc);
            const int col = (int)(rem - r * SF_STEM_PC);
            const int tin = tin0 + (int)f, hin = hin0 + (int)r, win = win0 + col;
            ok[u] = (int)c < nch && (unsigned)tin < (unsigned)p.Ti && (unsigned)hin < (unsigned)p.Hi &&
                    (unsigned)win < (unsigned)p.Wi;
            const int64_t off = ok[u] ? ((((int64_t)t.n * p.Ti + tin) * p.Hi + hin) * p.Wi + win) * p.ldx : 0;
            v[u] = ld16(p.x + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = base + tid + NTHREADS * u;
            if (c < nch) st16(patch + (int64_t)c * 8, ok[u] ? v[u] : zero8());
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward: wave w owns output frame t0 + w of the tile (8 rows x 16 pixels = 8 MFMA column tiles)
// SH / KH > 0: the row stride and kernel height are compile-time (the Fast stem: 2 / 7, the 8-channel 3x3 layers: 1 / 3) and the
// loop runs over PATCH rows: patch row r of a slice frame is the operand of every (output row i, tap kh) with i * SH + kh == r, so
// it is read from LDS ONCE and multiplied against up to ceil(KH / SH) weight slices -- 21 reads per frame slice instead of 56 for
// the stem, 10 instead of 24 for the 3x3 layers (the kernel was LDS-read-bound: one ds_read_b128 per MFMA).  SH = 0: generic loop.
template <int PCH, int SH = 0, int KH = 0>
__global__ __launch_bounds__(SF_THREADS) void sf_stem_fwd_kernel(StemParams p) {
    __shared__ __attribute__((aligned(16))) f16 patch[PCH * 8];
    __shared__ float s_red[4][2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g4 = lane >> 4;
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const StemTile t = stem_tile(p, bid);
    stem_load_patch<SF_THREADS, (PCH <= SF_STEM_CHUNKS_SMALL ? 3 : 7)>(p, patch, t, tid);

    f32x4 acc[SF_STEM_TH];
#pragma unroll
    for (int i = 0; i < SF_STEM_TH; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool co_ok = pl < p.Co && g4 < p.kwc;
    const f16* wrow = p.wmat + (co_ok ? (int64_t)pl * p.ldw + p.wo0 + p.wog * g4 : 0);
    if constexpr (SH > 0) {
        constexpr int PRC = (SF_STEM_TH - 1) * SH + KH;            // patch rows of a frame (== p.PR)
        __syncthreads();
        for (int kt = 0; kt < p.kT; ++kt) {
            f16x8 w[KH];
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) w[kh] = co_ok ? ld16(wrow + (kt * KH + kh) * p.wos) : zero8();
            const f16* base = patch + (((wave * p.sT + kt) * PRC) * SF_STEM_PC + pl + g4) * 8;
#pragma unroll
            for (int r = 0; r < PRC; ++r) {
                const f16x8 px = ld16(base + r * (SF_STEM_PC * 8));
#pragma unroll
                for (int kh = r % SH; kh < KH; kh += SH) {
                    const int i = (r - kh) / SH;                        // compile-time after unrolling
                    if (r >= kh && i < SF_STEM_TH) acc[i] = SF_MFMA16(w[kh], px, acc[i]);
                }
            }
        }
    } else {
        const int nsl = p.kT * p.kH;
        f16x8 wf = ld16(wrow);
        __syncthreads();
        int kt = 0, kh = 0;
        const int row_step = p.sH * SF_STEM_PC * 8;
        for (int s = 0; s < nsl; ++s) {
            const int sn = s + 1 < nsl ? s + 1 : s;
            const f16x8 wnext = ld16(wrow + (co_ok ? sn * p.wos : 0));
            const f16x8 a = co_ok ? wf : zero8();
            const f16* base = patch + (((wave * p.sT + kt) * p.PR + kh) * SF_STEM_PC + pl + g4) * 8;
            f16x8 px[SF_STEM_TH];
#pragma unroll
            for (int i = 0; i < SF_STEM_TH; ++i) px[i] = ld16(base + i * row_step);
#pragma unroll
            for (int i = 0; i < SF_STEM_TH; ++i) acc[i] = SF_MFMA16(a, px[i], acc[i]);
            wf = wnext;
            if (++kh == p.kH) { kh = 0; ++kt; }
        }
    }

    // lane: channels 4*g4 .. 4*g4+3 of pixel (t0 + wave, h0 + i, w0 + pl)
    const int to = t.t0 + wave, wo = t.w0 + pl;
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && 4 * g4 < p.Co) {
#pragma unroll
        for (int r = 0; r < 4; ++r) b4[r] = p.bias[4 * g4 + r];
    }
    const float lo = p.out_relu ? 0.f : -INFINITY;
    float msc[4] = {0.f, 0.f, 0.f, 0.f}, msh[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bnb_y && !p.bnb_bits && 4 * g4 < p.Co) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { msc[r] = p.bnb_scale[4 * g4 + r]; msh[r] = p.bnb_shift[4 * g4 + r]; }
    }
#pragma unroll
    for (int i = 0; i < SF_STEM_TH; ++i) {
        const int ho = t.h0 + i;
        const bool ok = to < p.To && ho < p.Ho && wo < p.Wo;
        if (ok) {
            const int64_t m = (((int64_t)t.n * p.To + to) * p.Ho + ho) * p.Wo + wo;
            f16x4 o = {(f16)fmaxf(acc[i][0] + b4[0], lo), (f16)fmaxf(acc[i][1] + b4[1], lo),
                       (f16)fmaxf(acc[i][2] + b4[2], lo), (f16)fmaxf(acc[i][3] + b4[3], lo)};
            if (p.bnb_y) {
                if (4 * g4 < p.Co) {
                    const f16x4 yv = *reinterpret_cast<const f16x4*>(p.bnb_y + m * p.bnb_ld + 4 * g4);
                    const uint32_t mb = p.bnb_bits ? (uint32_t)p.bnb_bits[m * (p.Co >> 3) + (g4 >> 1)] >> (4 * (g4 & 1)) : 0u;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool open = p.bnb_bits ? ((mb >> r) & 1u) != 0u : ((float)yv[r] * msc[r] + msh[r] > 0.f);
                        const float g = open ? (float)o[r] : 0.f;
                        s4[r] += g;
                        q4[r] += g * (float)yv[r];
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s4[r] += acc[i][r]; q4[r] += acc[i][r] * acc[i][r]; }
            }
            if (4 * g4 < p.Co) *reinterpret_cast<f16x4*>(p.y + m * p.ldy + 4 * g4) = o;
        }
    }
    if (p.stat_part) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int mask = 1; mask < 16; mask <<= 1) {
                s4[r] += __shfl_xor(s4[r], mask);
                q4[r] += __shfl_xor(q4[r], mask);
            }
            if (pl == 0) {
                s_red[wave][0][4 * g4 + r] = s4[r];
                s_red[wave][1][4 * g4 + r] = q4[r];
            }
        }
        __syncthreads();
        if (tid < 2 * p.Co) {
            const int which = tid / p.Co, co = tid % p.Co;
            const float v = s_red[0][which][co] + s_red[1][which][co] + s_red[2][which][co] + s_red[3][which][co];
            p.stat_part[((int64_t)bid * 2 + which) * p.Co + co] = v;
            // the table has one row per 128 output positions: rows no workgroup owns must read as zero
            for (int64_t row = (int64_t)bid + gridDim.x; row < p.stat_rows; row += gridDim.x)
                p.stat_part[(row * 2 + which) * p.Co + co] = 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward, SLIDING over T with TWO output frames per MFMA (round 6; sT == 1, 1 < kT <= 5, Co <= 8: the Fast pathway's stem).
// The tile kernel above stages (TT - 1) sT + kT = 8 input frames for 4 output frames: every input frame goes through LDS twice (1.28 GB
// of patch loads per launch at batch 32 for a 0.31 GB clip), each 16-byte chunk with its own div / mod addressing, and its MFMAs use 8
// of their 16 rows.  Here
//   * a workgroup owns an (8 rows x 16 pixels) column of the output and WALKS a run of `seg_groups` groups of 4 output frames: the 8
//     frame slots are a ring (input frame j of the run lives in slot j & 7), a step stages only its 4 new frames -- requested BEFORE
//     the step's MFMAs, written into the slots the step retires after them -- and a thread's two chunk offsets inside a frame are
//     computed once per run;
//   * wave w owns output frames A = 2 (w & 1), A + 1 of the step and rows 4 (w >> 1) .. + 3 of the tile: MFMA rows 0-7 carry the
//     weights of tap kt = jj for frame A, rows 8-15 those of tap kt = jj - 1 for frame A + 1, both against input frame A + jj:
//     kT + 1 = 6 frame passes per 2 output frames instead of 2 kT = 10 (168 MFMAs and 78 ds_read_b128 per wave and step, 280 / 105
//     before);
//   * the weight slices of the next pass are requested before the MFMAs of the current one.
template <int SH, int KH>
__global__ __launch_bounds__(SF_THREADS) void sf_stem_fwd_slide_kernel(StemParams p) {
    constexpr int PRC = (SF_STEM_TH - 1) * SH + KH;                // patch rows of a frame
    constexpr int FCH = PRC * SF_STEM_PC;                          // chunks of a frame
    constexpr int LPF = (FCH + SF_THREADS - 1) / SF_THREADS;       // chunks of a frame per thread
    constexpr int NEWF = SF_STEM_TT;                               // frames a step adds
    constexpr int RH = SF_STEM_TH / 2;                             // output rows of a wave
    constexpr int WR = (RH - 1) * SH + KH;                         // patch rows of a wave's window
    static_assert(8 * FCH <= SF_STEM_CHUNKS, "the ring is the tile kernel's patch");
    static_assert(SF_STEM_TT == 4 && SF_THREADS == 256, "4 waves = 2 frame pairs x 2 row halves");
    __shared__ __attribute__((aligned(16))) f16 ring[8 * FCH * 8];
    __shared__ float s_red[8][2][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g4 = lane >> 4;
    const int fp = wave & 1, rh = wave >> 1;
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const StemTile t = stem_tile(p, bid);                          // t.t0 / TT = the run
    const int t_begin = (t.t0 / SF_STEM_TT) * p.seg_groups * SF_STEM_TT;
    const int t_end = t_begin + p.seg_groups * SF_STEM_TT < p.To ? t_begin + p.seg_groups * SF_STEM_TT : p.To;
    const int G = (t_end - t_begin + SF_STEM_TT - 1) / SF_STEM_TT;

    int coff[LPF];
    bool cok[LPF];
    {
        const int hin0 = t.h0 * SH - p.pH, win0 = t.w0 - p.pW;
#pragma unroll
        for (int u = 0; u < LPF; ++u) {
            const int c = tid + SF_THREADS * u;
            const int r = c / SF_STEM_PC, col = c - r * SF_STEM_PC;
            const int hin = hin0 + r, win = win0 + col;
            cok[u] = c < FCH && (unsigned)hin < (unsigned)p.Hi && (unsigned)win < (unsigned)p.Wi;
            coff[u] = cok[u] ? (hin * p.Wi + win) * p.ldx : 0;
        }
    }
    const int64_t fstride = (int64_t)p.Hi * p.Wi * p.ldx;
    const f16* xn = p.x + (int64_t)t.n * p.Ti * fstride;
    const int tin0 = t_begin - p.pT;                               // input frame of ring index 0 (sT == 1)
    f16x8 nv[NEWF][LPF];
    // request ring frames j0 .. j0 + NEWF - 1 / write them to their slots
    auto request = [&](int j0) {
#pragma unroll
        for (int f = 0; f < NEWF; ++f) {
            const int tin = tin0 + j0 + f;
            const f16* xf = xn + ((unsigned)tin < (unsigned)p.Ti ? tin * fstride : 0);
#pragma unroll
            for (int u = 0; u < LPF; ++u) nv[f][u] = ld16(xf + coff[u]);
        }
    };
    auto deposit = [&](int j0) {
#pragma unroll
        for (int f = 0; f < NEWF; ++f) {
            const bool tok = (unsigned)(tin0 + j0 + f) < (unsigned)p.Ti;
            f16* slot = ring + ((j0 + f) & 7) * (FCH * 8);
#pragma unroll
            for (int u = 0; u < LPF; ++u) {
                const int c = tid + SF_THREADS * u;
                if (c < FCH) st16(slot + c * 8, tok && cok[u] ? nv[f][u] : zero8());
            }
        }
    };
    request(0);
    deposit(0);
    request(NEWF);
    deposit(NEWF);

    // A operand of lane (pl, g4): MFMA row pl = (frame pl >> 3 of the pair, channel pl & 7), k chunk g4; pass jj -> tap jj - (pl >> 3)
    const int which = pl >> 3, co = pl & 7;
    const bool co_ok = co < p.Co && g4 < p.kwc;
    const f16* wrow = p.wmat + (co_ok ? (int64_t)co * p.ldw + p.wo0 + p.wog * g4 : 0);
    f16x8 wc[KH], wn[KH];
    auto load_w = [&](int jj, f16x8* w) {
        const int kt = jj - which;
        const bool ok = co_ok && kt >= 0 && kt < p.kT;
        const f16* wp = wrow + (ok ? kt * KH * p.wos : 0);
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            const f16x8 v = ld16(wp + kh * p.wos);
            w[kh] = ok ? v : zero8();
        }
    };
    load_w(0, wc);
    // D of lane (pl, g4): rows 4 g4 .. + 3 = frame g4 >> 1 of the pair, channels 4 (g4 & 1) .. + 3, pixel w0 + pl
    const int wo = t.w0 + pl, c4 = 4 * (g4 & 1);
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && c4 < p.Co) {
#pragma unroll
        for (int r = 0; r < 4; ++r) b4[r] = p.bias[c4 + r];
    }
    const float lo = p.out_relu ? 0.f : -INFINITY;

    for (int g = 0; g < G; ++g) {
        const bool more = g + 1 < G;
        if (more) request(NEWF * (g + 2));
        __syncthreads();                                           // the slots of this step are written
        f32x4 acc[RH];
#pragma unroll
        for (int i = 0; i < RH; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int jj = 0; jj <= p.kT; ++jj) {
            load_w(jj < p.kT ? jj + 1 : 0, wn);
            const f16* base = ring + ((((NEWF * g + 2 * fp + jj) & 7) * PRC + RH * SH * rh) * SF_STEM_PC + pl + g4) * 8;
#pragma unroll
            for (int r = 0; r < WR; ++r) {
                const f16x8 px = ld16(base + r * (SF_STEM_PC * 8));
#pragma unroll
                for (int kh = r % SH; kh < KH; kh += SH) {
                    const int i = (r - kh) / SH;                        // compile-time after unrolling
                    if (r >= kh && i < RH) acc[i] = SF_MFMA16(wc[kh], px, acc[i]);
                }
            }
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) wc[kh] = wn[kh];
        }
        const int to = t_begin + SF_STEM_TT * g + 2 * fp + (g4 >> 1);
#pragma unroll
        for (int i = 0; i < RH; ++i) {
            const int ho = t.h0 + RH * rh + i;
            if (to < t_end && ho < p.Ho && wo < p.Wo && c4 < p.Co) {
                const int64_t m = (((int64_t)t.n * p.To + to) * p.Ho + ho) * p.Wo + wo;
#pragma unroll
                for (int r = 0; r < 4; ++r) { s4[r] += acc[i][r]; q4[r] += acc[i][r] * acc[i][r]; }
                *reinterpret_cast<f16x4*>(p.y + m * p.ldy + c4) =
                    (f16x4){(f16)fmaxf(acc[i][0] + b4[0], lo), (f16)fmaxf(acc[i][1] + b4[1], lo),
                            (f16)fmaxf(acc[i][2] + b4[2], lo), (f16)fmaxf(acc[i][3] + b4[3], lo)};
            }
        }
        if (more) {
            __syncthreads();                                       // every wave is done with the slots this step retires
            deposit(NEWF * (g + 2));
        }
    }
    if (p.stat_part) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int mask = 1; mask < 16; mask <<= 1) {
                s4[r] += __shfl_xor(s4[r], mask);
                q4[r] += __shfl_xor(q4[r], mask);
            }
            if (pl == 0) {
                s_red[2 * wave + (g4 >> 1)][0][c4 + r] = s4[r];
                s_red[2 * wave + (g4 >> 1)][1][c4 + r] = q4[r];
            }
        }
        __syncthreads();
        if (tid < 2 * p.Co) {
            const int wh = tid / p.Co, ch = tid % p.Co;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) v += s_red[k][wh][ch];
            p.stat_part[((int64_t)bid * 2 + wh) * p.Co + ch] = v;
            for (int64_t row = (int64_t)bid + gridDim.x; row < p.stat_rows; row += gridDim.x)
                p.stat_part[(row * 2 + wh) * p.Co + ch] = 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient: persistent workgroups (8 waves) over a contiguous range of tiles; wave w accumulates the
// K-slices s = w, w+8, ... (s = kt*kH + kh, 32 weights each, two 16-column MFMA tiles) for all 16 channel rows.
template <int COP, int PCH>
__global__ __launch_bounds__(SF_STEM_WG_THREADS, 4) void sf_stem_wgrad_kernel(StemParams p) {
    constexpr int NW = SF_STEM_WG_THREADS / 64;
    constexpr int NQ = SF_STEM_MAX_SLICES / NW;
    constexpr int NPIX = SF_STEM_TT * SF_STEM_TH * SF_STEM_TW;
    __shared__ __attribute__((aligned(16))) f16 patch[PCH * 8];
    __shared__ __attribute__((aligned(16))) f16 dyt[NPIX * COP + 8];     // + 8 zeros: the channel chunks beyond COP
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g4 = lane >> 4;
    const int nsl = p.kT * p.kH;
    // few slices (the 8-channel 3x3 layers have 3): instead of idling five waves, `wgroups` groups of nsl waves share the rows
    const int ng = p.wgroups > 1 ? p.wgroups : 1;
    const int grp = ng > 1 ? wave / nsl : 0, sw = ng > 1 ? wave - grp * nsl : wave;
    const bool wactive = grp < ng;

    f32x4 acc[NQ][2];
    int koff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        acc[q][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[q][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int s = sw + NW * q;
        const int kt = s / p.kH, kh = s - kt * p.kH;
        koff[q] = (kt * p.PR + kh) * SF_STEM_PC;
    }
    if (tid < 8) dyt[NPIX * COP + tid] = (f16)0;

    // per-lane pixel of the transposed reads: kk = 8*g4 + 4*h + (pl>>2) -> (row offset kk>>4, column kk&15)
    int pix_row[2], pix_col[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int kk = 8 * g4 + 4 * h + (pl >> 2);
        pix_row[h] = kk >> 4;
        pix_col[h] = kk & 15;
    }
    const int cchunk = 4 * (pl & 3);
    const int frame_step = p.sT * p.PR * SF_STEM_PC;

    // XCD-contiguous block order (sf_common.h: xcd_remap): a block takes a run of consecutive tiles, but the tile that shares a
    // patch's 4-frame temporal halo is tiles_w * tiles_h tiles away -- with the plain order behind another XCD's L2
    const int bxs = p.plain_order ? (int)blockIdx.x : (int)xcd_remap(blockIdx.x, gridDim.x);
    const int tile_begin = bxs * p.tiles_per_block;
    int tile_end = tile_begin + p.tiles_per_block;
    if (tile_end > p.ntiles) tile_end = p.ntiles;
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const StemTile t = stem_tile(p, (uint32_t)tile);
        __syncthreads();
        stem_load_patch<SF_STEM_WG_THREADS, (PCH <= SF_STEM_CHUNKS_SMALL ? 2 : 7)>(p, patch, t, tid);
        for (int idx = tid; idx < NPIX * (COP / 8); idx += SF_STEM_WG_THREADS) {
            const int pix = idx / (COP / 8), cg = idx % (COP / 8);
            const int w = pix % SF_STEM_TW, hh = (pix / SF_STEM_TW) % SF_STEM_TH, tt = pix / (SF_STEM_TW * SF_STEM_TH);
            const int to = t.t0 + tt, ho = t.h0 + hh, wo = t.w0 + w;
            const bool ok = to < p.To && ho < p.Ho && wo < p.Wo && cg * 8 < p.Co;
            const int64_t m = (((int64_t)t.n * p.To + to) * p.Ho + ho) * p.Wo + wo;
            const f16x8 v = ld16(p.dy + (ok ? m * p.ldy + cg * 8 : 0));
            st16(dyt + pix * COP + cg * 8, ok ? v : zero8());
        }
        __syncthreads();
#pragma unroll 1
        for (int tc = grp; tc < SF_STEM_TT * (SF_STEM_TH / 2); tc += ng) {
            if (wactive) {
                const int tt = tc / (SF_STEM_TH / 2), c = tc % (SF_STEM_TH / 2);
                f16x8 af;
                int xoff[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int hh = 2 * c + pix_row[h];
                    const int pix = (tt * SF_STEM_TH + hh) * SF_STEM_TW + pix_col[h];
                    const f16* ptr = cchunk < COP ? dyt + pix * COP + cchunk : dyt + NPIX * COP + (cchunk & 4);
                    f16x4 tv = as_f16x4(SF_LDS_TR16(ptr));
                    af[4 * h + 0] = tv[0]; af[4 * h + 1] = tv[1]; af[4 * h + 2] = tv[2]; af[4 * h + 3] = tv[3];
                    xoff[h] = (tt * frame_step + hh * p.sH * SF_STEM_PC + pix_col[h]) * 8 + cchunk;
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (sw + NW * q < nsl) {
#pragma unroll
                        for (int jh = 0; jh < 2; ++jh) {
                            f16x8 bf;
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                f16x4 tv = as_f16x4(SF_LDS_TR16(patch + xoff[h] + (koff[q] + 2 * jh) * 8));
                                bf[4 * h + 0] = tv[0]; bf[4 * h + 1] = tv[1]; bf[4 * h + 2] = tv[2]; bf[4 * h + 3] = tv[3];
                            }
                            acc[q][jh] = SF_MFMA16(af, bf, acc[q][jh]);
                        }
                    }
                }
            }
        }
    }
    // slab[co][k]: co = 4*g4 + r, k = s*32 + 16*jh + pl
    float* slab = p.ws + ((int64_t)bxs * ng + grp) * 16 * p.Kpad;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int s = sw + NW * q;
        if (wactive && s < nsl) {
#pragma unroll
            for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[(int64_t)(4 * g4 + r) * p.Kpad + s * 32 + 16 * jh + pl] = acc[q][jh][r];
        }
    }
}
