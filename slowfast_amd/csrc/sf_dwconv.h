// Depthwise 3-D convolution on channels-last fp16 tensors (HBM-bound stencil, no MFMA).
//
// Reference call sites: MViT attention pooling pool_q / pool_k / pool_v = Conv3d(96, 96, 3x3x3, stride (1,s,s),
// groups=96) applied per head with the cls token routed around it (slowfast/models/attention.py:13-45, 227-266);
// X3D X3DTransform.b = Conv3d(C, C, 3x3x3, stride (1,s,s), groups=C) (resnet_helper.py:214-224) and the X3D stem's
// temporal conv Conv3d(24, 24, (5,1,1), groups=24) (stem_helper.py:267-275).
//
// Rows of a tensor are (n, [cls], t, h, w) with a row pitch; row(n, pos) = n*(S + cls) + cls + pos.  The weight is
// the nn.Conv3d parameter itself, fp32 [Cw][taps]; C may be a multiple of Cw (heads side by side share the
// weight: channel c uses weight channel c % Cw).
#pragma once
#include "sf_bn.h"
#include "sf_common.h"

#define SF_DW_MAX_W 12288   // taps * Cw floats staged in LDS (48 KiB): 27 x 432 for the widest X3D-M stage

struct DwParams {
    const f16* x; int ldx;          // fwd/wgrad: input;  dgrad: unused
    const float* w;                 // [Cw][taps]
    f16* y; int ldy;                // fwd: output;       dgrad: dx (rows of the INPUT space)
    const f16* dy; int lddy;        // dgrad/wgrad: output gradient
    int N, C, Cw, cls;
    int Cwreal;                     // rows of w; weight channels [Cwreal, Cw) are zero padding
    int Ti, Hi, Wi, To, Ho, Wo;
    int kT, kH, kW, sT, sH, sW, pT, pH, pW;
    RowTile rt;                     // rows iterated: fwd/wgrad N*(So+cls), dgrad N*(Si+cls)
    float* stat_part;               // fwd, optional: [gridDim.x][2][C] per-block sum / sum of squares (BatchNorm)
    float* wpart;                   // wgrad: [gridDim.x][taps][C] per-block partial weight gradients
    FastDiv fdRow, fdW, fdH;        // row -> (n, r); r - cls -> (t, h, w) of the iterated space
    FastDiv fdsT, fdsH, fdsW;
    int xcd_order;                  // 1: row blocks are handed out XCD-contiguously (dw_block_id)
};

// Row block of this workgroup.  Workgroup b runs on XCD b % 8 (sf_common.h: xcd_remap): with the plain order the row blocks of
// neighbouring lines and frames -- which share most of their 3x3x3 input window -- sit behind eight different L2s, and every
// input line is fetched by several of them (measured on the stage-3 pooling of MViTv2-S, profiles/r4/r4_v3_pmc_tokens.md: L2 hit
// rate 25 %, FETCH_SIZE 3.4x the input).  The remap gives every XCD a contiguous range of row blocks (whole samples), so a
// window is re-read from ONE L2.  A bijection of the block ids: partial-sum tables are still folded in index order.
// The weight-gradient grids carry the kt plane (or tap chunk) in blockIdx.z: the planes of one row block go to the same XCD.
__device__ __forceinline__ int dw_block_id(const DwParams& p, int* bz = nullptr) {
    if (!p.xcd_order || gridDim.y != 1) {
        if (bz) *bz = (int)blockIdx.z;
        return (int)blockIdx.x;
    }
    const uint32_t lin = blockIdx.x + gridDim.x * blockIdx.z;          // dispatch order: XCD = lin % 8
    const uint32_t r = xcd_remap(lin, gridDim.x * gridDim.z);
    if (bz) *bz = (int)(r % gridDim.z);
    return (int)(r / gridDim.z);
}

__device__ __forceinline__ void dw_stage_weights(const DwParams& p, float* s_w) {
    // LDS layout [tap][Cw] so that a thread's 8 channels are contiguous
    const int taps = p.kT * p.kH * p.kW;
    for (int i = threadIdx.x; i < taps * p.Cw; i += SF_THREADS) {
        const int cw = i / taps, tap = i % taps;
        s_w[tap * p.Cw + cw] = cw < p.Cwreal ? p.w[i] : 0.f;
    }
}

// fp16 copy of the weights for the W-blocked kernels ([tap][Cw] like dw_stage_weights): half the LDS (one more resident
// workgroup per CU for the wide layers), one ds_read_b128 per tap, and the products run as v_fma_mix (fp16 x fp16 -> fp32
// accumulate) without separate conversions -- the same operand precision as the MFMA convolutions.
__device__ __forceinline__ void dw_stage_weights16(const DwParams& p, f16* s_w) {
    const int taps = p.kT * p.kH * p.kW;
    for (int i = threadIdx.x; i < taps * p.Cw; i += SF_THREADS) {
        const int cw = i / taps, tap = i % taps;
        s_w[tap * p.Cw + cw] = cw < p.Cwreal ? (f16)p.w[i] : (f16)0;
    }
}

__device__ __forceinline__ bool dw_decode(const DwParams& p, uint32_t row, uint32_t& n, int& t, int& h, int& w) {
    // returns true for the cls row of a sample
    uint32_t r, q, ww, hh, tt;
    fd_divmod(row, p.fdRow, n, r);
    if (p.cls && r == 0) { t = h = w = 0; return true; }
    fd_divmod(r - (uint32_t)p.cls, p.fdW, q, ww);
    fd_divmod(q, p.fdH, tt, hh);
    t = (int)tt; h = (int)hh; w = (int)ww;
    return false;
}

// WSZ floats of LDS for the weights: 3072 (12 KiB) covers MViT's 27 x 96; the 48 KiB image is only taken when needed
// (with it the static LDS footprint, 66 KiB with the reduction scratch, allowed two workgroups per CU)
template <int WSZ>
__global__ __launch_bounds__(SF_THREADS) void sf_dwconv_fwd_kernel(DwParams p) {
    __shared__ float s_w[WSZ];
    __shared__ float s_red[SF_THREADS][17];
    dw_stage_weights(p, s_w);
    __syncthreads();
    const int bx = dw_block_id(p);
    int gcol, r0, r1, rstep;
    const bool active = p.rt.init(gcol, r0, r1, rstep, bx);
    const int c = gcol * 8;
    const int cw = c % p.Cw;
    const int64_t Si = (int64_t)p.Ti * p.Hi * p.Wi + p.cls;
    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
    if (active) {
        for (int m = r0; m < r1; m += rstep) {
            uint32_t n;
            int to, ho, wo;
            const bool is_cls = dw_decode(p, (uint32_t)m, n, to, ho, wo);
            const f16* xb = p.x + ((int64_t)n * Si) * p.ldx + c;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            if (is_cls) {
                f16x8 v = ld16(xb);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = (float)v[e];
            } else {
                int tap = 0;
                for (int kt = 0; kt < p.kT; ++kt) {
                    const int t = to * p.sT - p.pT + kt;
                    for (int kh = 0; kh < p.kH; ++kh) {
                        const int h = ho * p.sH - p.pH + kh;
                        for (int kw = 0; kw < p.kW; ++kw, ++tap) {
                            const int w = wo * p.sW - p.pW + kw;
                            if ((unsigned)t >= (unsigned)p.Ti || (unsigned)h >= (unsigned)p.Hi ||
                                (unsigned)w >= (unsigned)p.Wi) continue;
                            f16x8 v = ld16(xb + (p.cls + ((int64_t)t * p.Hi + h) * p.Wi + w) * p.ldx);
                            const float* wt = s_w + tap * p.Cw + cw;
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e] * wt[e];
                        }
                    }
                }
            }
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = (f16)acc[e];
                ssum[e] += acc[e];
                ssq[e] += acc[e] * acc[e];
            }
            st16(p.y + (int64_t)m * p.ldy + c, o);
        }
    }
    if (p.stat_part)
        rowtile_reduce_store(p.rt, active, c, ssum, ssq, p.stat_part + (int64_t)bx * 2 * p.rt.C, s_red);
}

// dx[n, t, h, w, c] = sum_taps w[c][tap] * dy[n, (t + pT - kt)/sT, (h + pH - kh)/sH, (w + pW - kw)/sW, c]
// (terms with a non-integral or out-of-range quotient vanish); the cls row passes through.
template <int WSZ>
__global__ __launch_bounds__(SF_THREADS) void sf_dwconv_dgrad_kernel(DwParams p) {
    __shared__ float s_w[WSZ];
    dw_stage_weights(p, s_w);
    __syncthreads();
    const int bx = dw_block_id(p);
    int gcol, r0, r1, rstep;
    if (!p.rt.init(gcol, r0, r1, rstep, bx)) return;
    const int c = gcol * 8;
    const int cw = c % p.Cw;
    const int64_t So = (int64_t)p.To * p.Ho * p.Wo + p.cls;
    for (int m = r0; m < r1; m += rstep) {
        uint32_t n;
        int t, h, w;
        const bool is_cls = dw_decode(p, (uint32_t)m, n, t, h, w);
        const f16* db = p.dy + ((int64_t)n * So) * p.lddy + c;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (is_cls) {
            f16x8 v = ld16(db);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = (float)v[e];
        } else {
            // the stride divisions are hoisted per axis: with strides >= the kernel extent (MViT k/v pooling, stride
            // (1,8,8) / (1,4,4)) most positions have no contributing tap along h or w and leave after 3-6 divisions
            for (int kt = 0; kt < p.kT; ++kt) {
                const int ut = t + p.pT - kt;
                if (ut < 0) continue;
                uint32_t qt, rt;
                fd_divmod((uint32_t)ut, p.fdsT, qt, rt);
                if (rt || qt >= (uint32_t)p.To) continue;
                for (int kh = 0; kh < p.kH; ++kh) {
                    const int uh = h + p.pH - kh;
                    if (uh < 0) continue;
                    uint32_t qh, rh;
                    fd_divmod((uint32_t)uh, p.fdsH, qh, rh);
                    if (rh || qh >= (uint32_t)p.Ho) continue;
                    for (int kw = 0; kw < p.kW; ++kw) {
                        const int uw = w + p.pW - kw;
                        if (uw < 0) continue;
                        uint32_t qw, rw;
                        fd_divmod((uint32_t)uw, p.fdsW, qw, rw);
                        if (rw || qw >= (uint32_t)p.Wo) continue;
                        const int tap = (kt * p.kH + kh) * p.kW + kw;
                        f16x8 v = ld16(db + (p.cls + ((int64_t)qt * p.Ho + qh) * p.Wo + qw) * p.lddy);
                        const float* wt = s_w + tap * p.Cw + cw;
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] += (float)v[e] * wt[e];
                    }
                }
            }
        }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)acc[e];
        st16(p.y + (int64_t)m * p.ldy + c, o);
    }
}

// Per-block partial weight gradients: wpart[blk][tap][c] = sum over the block's output rows of dy * x(tap).
// blockIdx.z selects the temporal tap kt and a chunk of 9 of that plane's kH*kW taps, which are accumulated in registers
// (one chunk for the 3x3 planes of every MViTv2 / X3D config; MViTv1's stride+1 pooling kernels, 1x5x5 and 1x9x9 in
// configs/Kinetics/MVIT_B_32x3_CONV.yaml, take 3 and 9 chunks, each re-reading dy).
__global__ __launch_bounds__(SF_THREADS) void sf_dwconv_wgrad_kernel(DwParams p) {
    __shared__ float s_red[SF_THREADS][9];
    int bz;
    const int bx = dw_block_id(p, &bz);
    int gcol, r0, r1, rstep;
    const bool active = p.rt.init(gcol, r0, r1, rstep, bx);
    const int c = gcol * 8;
    const int nsp_all = p.kH * p.kW;
    const int nchunk = (nsp_all + 8) / 9;
    const int kt = bz / nchunk;
    const int i0 = (bz % nchunk) * 9;
    const int nsp = nsp_all - i0 < 9 ? nsp_all - i0 : 9;
    const int64_t Si = (int64_t)p.Ti * p.Hi * p.Wi + p.cls;
    float acc[9][8];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
    if (active) {
        for (int m = r0; m < r1; m += rstep) {
            uint32_t n;
            int to, ho, wo;
            if (dw_decode(p, (uint32_t)m, n, to, ho, wo)) continue;
            const int t = to * p.sT - p.pT + kt;
            if ((unsigned)t >= (unsigned)p.Ti) continue;
            f16x8 d = ld16(p.dy + (int64_t)m * p.lddy + c);
            const f16* xb = p.x + ((int64_t)n * Si + p.cls) * p.ldx + c;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                if (i < nsp) {
                    const int kh = (i0 + i) / p.kW, kw = (i0 + i) % p.kW;
                    const int h = ho * p.sH - p.pH + kh, w = wo * p.sW - p.pW + kw;
                    if ((unsigned)h < (unsigned)p.Hi && (unsigned)w < (unsigned)p.Wi) {
                        f16x8 v = ld16(xb + (((int64_t)t * p.Hi + h) * p.Wi + w) * p.ldx);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[i][e] += (float)d[e] * (float)v[e];
                    }
                }
            }
        }
    }
    // fold the row-lanes of the block, one tap at a time (fixed order)
    const int G = p.rt.C >> 3;
    const int TG = G < SF_THREADS ? G : SF_THREADS;
    const int rpi = SF_THREADS / TG;
    const int taps = p.kT * nsp_all;
#pragma unroll
    for (int i = 0; i < 9; ++i) {       // static register indices: a runtime-indexed acc[] would live in scratch
        if (i < nsp) {                  // block-uniform
#pragma unroll
            for (int e = 0; e < 8; ++e) s_red[threadIdx.x][e] = acc[i][e];
            __syncthreads();
            if (active && (int)threadIdx.x < TG) {
                float* o = p.wpart + ((int64_t)bx * taps + kt * nsp_all + i0 + i) * p.rt.C + c;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float a = 0.f;
                    for (int k = 0; k < rpi; ++k) a += s_red[threadIdx.x + k * TG][e];
                    o[e] = a;
                }
            }
            __syncthreads();
        }
    }
}

// ================================================================================================
// W-blocked variants for the common geometries (kW, sW) in {(3,1), (3,2), (1,1)} with pW = kW/2: a thread produces
// WB = 4 consecutive columns of one (n, t, h) line for its 8 channels, so every input column loaded for a (kt, kh)
// plane is reused by up to kW outputs and the per-tap address arithmetic is shared (2-3x fewer VALU operations per
// byte than the one-output-per-thread kernels above, which remain the fallback for other strides).
// Items iterated by the RowTile map: N*T*H*ceil(W/4) line groups of the ITERATED space, followed by N cls items.
#define SF_DW_WB 4
struct DwBlockIdx {
    FastDiv fdWG, fdH, fdT;
    int WG;
    int64_t groups;
};
__device__ __forceinline__ bool dwb_decode(const DwBlockIdx& bi, uint32_t item, uint32_t& n, int& t, int& h, int& w0) {
    if ((int64_t)item >= bi.groups) { n = (uint32_t)(item - bi.groups); t = h = w0 = 0; return true; }
    uint32_t q, wg, hh, tt;
    fd_divmod(item, bi.fdWG, q, wg);
    fd_divmod(q, bi.fdH, q, hh);
    fd_divmod(q, bi.fdT, n, tt);
    t = (int)tt; h = (int)hh; w0 = (int)wg * SF_DW_WB;
    return false;
}
// Out-of-range taps are read from a clamped (always valid) address and zeroed afterwards: the loads of a plane stay
// unconditional, so they are issued back to back instead of one branch + s_waitcnt vmcnt(0) per tap.
__device__ __forceinline__ f16x8 keep8(const f16x8& v, bool ok) { return ok ? v : zero8(); }
__device__ __forceinline__ int clampi(int x, int hi) { return x < 0 ? 0 : (x > hi ? hi : x); }
__device__ __forceinline__ void cvt8(const f16x8& v, float (&o)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
}

// ---- W-blocked forward stencil ----------------------------------------------------------------------------------------
// The first version of this kernel (rounds 1-2; history) spent ~215 VALU instructions per (kt, kh) plane on 48 packed FMAs:
// 72 fp16->fp32 converts (48 inputs + 24 weights), 24 v_cndmask zeroing out-of-range taps dword by dword, 32 v_mov copying
// the accumulators around a divergent "next valid plane" search loop, and that loop's exec-mask bookkeeping.  This one keeps
// the arithmetic (fp32 accumulate, same operand rounding) and removes the overhead:
//   * every plane of the kT x kH window is visited (a uniform loop: accumulators stay in place, no divergence);
//   * a tap outside the input is redirected to a 16-byte line of zeros in global memory by selecting the ADDRESS
//     (one 64-bit select per plane, plus one per edge column) instead of zeroing the loaded DATA;
//   * when the row length is a multiple of the 4-column block only column 0 of a group can fall outside (uniform test);
//   * weights are staged as fp32 (no per-plane weight converts; the 141 VGPRs of this kernel cap residency at 3 waves
//     per SIMD long before the larger LDS footprint does).
// Measured with the same restructuring of the data- and weight-gradient stencils (profiles/r3/r3_v11_dw_v2_ab.txt): X3D-M
// 1183-1191 -> 1198-1203 clips/s, MViTv2-S 547-549 -> 551-553; the first version is gone.
__device__ __attribute__((aligned(16))) f16 sf_dw_zero_line[8] = {};

// 8 consecutive weights of one tap as fp32 from either LDS image (two ds_read_b128, or one + 8 converts)
__device__ __forceinline__ void dw_load_w8(const float* w, float (&o)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(w), b = *reinterpret_cast<const f32x4*>(w + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
}
__device__ __forceinline__ void dw_load_w8(const f16* w, float (&o)[8]) { cvt8(ld16(w), o); }
__device__ __forceinline__ void dw_stage_weights_t(const DwParams& p, float* s_w) { dw_stage_weights(p, s_w); }
__device__ __forceinline__ void dw_stage_weights_t(const DwParams& p, f16* s_w) { dw_stage_weights16(p, s_w); }

// WT = float for the narrow layers (taps*Cw <= 3072: 12 KiB of LDS), f16 for the wide ones (24 KiB instead of 48 KiB, so
// that LDS does not cap residency below what the registers allow; costs 24 converts per plane)
template <int KW, int SW, int WSZ, typename WT>
__global__ __launch_bounds__(SF_THREADS) void sf_dwconv_fwd_blocked_kernel(DwParams p, DwBlockIdx bi) {
    constexpr int NIN = (SF_DW_WB - 1) * SW + KW;
    __shared__ __attribute__((aligned(16))) WT s_w[WSZ];        // taps*Cw weights, [tap][Cw]
    __shared__ float s_red[SF_THREADS][17];
    dw_stage_weights_t(p, s_w);
    __syncthreads();
    const int bx = dw_block_id(p);
    int gcol, r0, r1, rstep;
    const bool active = p.rt.init(gcol, r0, r1, rstep, bx);
    const int c = gcol * 8;
    const int cw = c % p.Cw;
    const int64_t Si = (int64_t)p.Ti * p.Hi * p.Wi + p.cls, So = (int64_t)p.To * p.Ho * p.Wo + p.cls;
    // columns 1 .. NIN-1 of every group are inside the row when the row is a whole number of groups (block-uniform)
    const bool edge_free = (p.Wo % SF_DW_WB) == 0 && (p.Wo - SF_DW_WB) * SW - p.pW + NIN - 1 < p.Wi;
    const f16* const zline = sf_dw_zero_line;
    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
    if (active) {
        for (int m = r0; m < r1; m += rstep) {
            uint32_t n;
            int to, ho, wo0;
            if (dwb_decode(bi, (uint32_t)m, n, to, ho, wo0)) {          // cls row: copy
                f16x8 v = ld16(p.x + (int64_t)n * Si * p.ldx + c);
                float f[8];
                cvt8(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { ssum[e] += f[e]; ssq[e] += f[e] * f[e]; }
                st16(p.y + (int64_t)n * So * p.ldy + c, v);
                continue;
            }
            const f16* xb = p.x + ((int64_t)n * Si + p.cls) * p.ldx + c;
            float acc[SF_DW_WB][8];
#pragma unroll
            for (int i = 0; i < SF_DW_WB; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
            const int wi0 = wo0 * SW - p.pW;
            const int t0 = to * p.sT - p.pT, h0 = ho * p.sH - p.pH;
            int tap = 0;
            for (int kt = 0; kt < p.kT; ++kt) {
                const int t = t0 + kt;
                const bool tv = (unsigned)t < (unsigned)p.Ti;
                for (int kh = 0; kh < p.kH; ++kh, tap += KW) {
                    const int h = h0 + kh;
                    const bool pv = tv && (unsigned)h < (unsigned)p.Hi;
                    // an invalid plane reads the zero line for every column (column stride 0)
                    const f16* line = pv ? xb + (((int64_t)t * p.Hi + h) * p.Wi) * p.ldx : zline;
                    const int64_t cstride = pv ? (int64_t)p.ldx : 0;
                    f16x8 raw[NIN];
#pragma unroll
                    for (int j = 0; j < NIN; ++j) {
                        const int col = wi0 + j;
                        const f16* a = line + (int64_t)col * cstride;
                        if (!(edge_free && j > 0)) a = (unsigned)col < (unsigned)p.Wi ? a : zline;
                        raw[j] = ld16(a);
                    }
                    const WT* wt = s_w + tap * p.Cw + cw;
#pragma unroll
                    for (int kw = 0; kw < KW; ++kw) {
                        float wv[8];
                        dw_load_w8(wt + kw * p.Cw, wv);
#pragma unroll
                        for (int i = 0; i < SF_DW_WB; ++i) {
                            const int j = i * SW + kw;      // compile-time after unrolling
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[i][e] += (float)raw[j][e] * wv[e];
                        }
                    }
                }
            }
            f16* yrow = p.y + ((int64_t)n * So + p.cls + ((int64_t)to * p.Ho + ho) * p.Wo + wo0) * p.ldy + c;
#pragma unroll
            for (int i = 0; i < SF_DW_WB; ++i) {
                if (wo0 + i < p.Wo) {
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        o[e] = (f16)acc[i][e];
                        ssum[e] += acc[i][e];
                        ssq[e] += acc[i][e] * acc[i][e];
                    }
                    st16(yrow + (int64_t)i * p.ldy, o);
                }
            }
        }
    }
    if (p.stat_part)
        rowtile_reduce_store(p.rt, active, c, ssum, ssq, p.stat_part + (int64_t)bx * 2 * p.rt.C, s_red);
}

// data gradient, blocked over 4 consecutive INPUT columns w0..w0+3 (w0 % 4 == 0).  With pW = KW/2 the output
// columns that can contribute are q0 + jj, q0 = (w0 + pW - (KW-1) + SW-1) / SW rounded as below, and the tap of
// (input i, column jj) is kw = B0 + i - jj*SW with a compile-time B0.
// (uniform plane loop and address-selected zero taps as in the forward stencil above)
// The planes that can contribute to input row (t, h) are kt = kt0 + a*sT, kh = kh0 + b*sH with kt0 = (t + pT) % sT,
// kh0 = (h + pH) % sH: a uniform loop over (a, b) visits exactly those candidates (no divisibility test, no skipped
// iterations for strided convolutions); a candidate outside the kernel or the output is read from the zero line.
template <int KW, int SW, int WSZ, typename WT>
__global__ __launch_bounds__(SF_THREADS) void sf_dwconv_dgrad_blocked_kernel(DwParams p, DwBlockIdx bi) {
    constexpr int PW = KW / 2;
    constexpr int B0 = SW == 1 ? KW - 1 : PW;                 // kw of (i = 0, jj = 0)
    constexpr int NQ = SW == 1 ? SF_DW_WB + KW - 1 : (SF_DW_WB - 1 + B0) / SW + 1;
    __shared__ __attribute__((aligned(16))) WT s_w[WSZ];
    dw_stage_weights_t(p, s_w);
    __syncthreads();
    const int bx = dw_block_id(p);
    int gcol, r0, r1, rstep;
    if (!p.rt.init(gcol, r0, r1, rstep, bx)) return;
    const int c = gcol * 8;
    const int cw = c % p.Cw;
    const int64_t Si = (int64_t)p.Ti * p.Hi * p.Wi + p.cls, So = (int64_t)p.To * p.Ho * p.Wo + p.cls;
    const int na = (p.kT + p.sT - 1) / p.sT, nb = (p.kH + p.sH - 1) / p.sH;
    // output columns q0 + jj, jj = 1 .. NQ-2, exist for every group when the input row is a whole number of groups and the
    // last group's columns stay below Wo (block-uniform); the first and the last column are always range-checked
    const int q_last0 = SW == 1 ? (p.Wi - SF_DW_WB) + PW - (KW - 1) : (p.Wi - SF_DW_WB) / 2;
    const bool edge_free = (p.Wi % SF_DW_WB) == 0 && q_last0 + NQ - 2 < p.Wo && (SW == 1 ? PW - (KW - 1) + 1 >= 0 : true);
    const f16* const zline = sf_dw_zero_line;
    for (int m = r0; m < r1; m += rstep) {
        uint32_t n;
        int t, h, w0;
        if (dwb_decode(bi, (uint32_t)m, n, t, h, w0)) {
            st16(p.y + (int64_t)n * Si * p.ldy + c, ld16(p.dy + (int64_t)n * So * p.lddy + c));
            continue;
        }
        const f16* db = p.dy + ((int64_t)n * So + p.cls) * p.lddy + c;
        float acc[SF_DW_WB][8];
#pragma unroll
        for (int i = 0; i < SF_DW_WB; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
        const int q0 = SW == 1 ? w0 + PW - (KW - 1) : w0 / 2;
        uint32_t qt0, kt0, qh0, kh0;
        fd_divmod((uint32_t)(t + p.pT), p.fdsT, qt0, kt0);      // t + pT = qt0*sT + kt0
        fd_divmod((uint32_t)(h + p.pH), p.fdsH, qh0, kh0);
        for (int a = 0; a < na; ++a) {
            const int kt = (int)kt0 + a * p.sT, qt = (int)qt0 - a;
            const bool tv = kt < p.kT && (unsigned)qt < (unsigned)p.To;
            for (int b = 0; b < nb; ++b) {
                const int kh = (int)kh0 + b * p.sH, qh = (int)qh0 - b;
                const bool pv = tv && kh < p.kH && (unsigned)qh < (unsigned)p.Ho;
                const f16* line = pv ? db + (((int64_t)qt * p.Ho + qh) * p.Wo) * p.lddy : zline;
                const int64_t cstride = pv ? (int64_t)p.lddy : 0;
                const int tap = pv ? (kt * p.kH + kh) * KW : 0;
                f16x8 raw[NQ];
#pragma unroll
                for (int jj = 0; jj < NQ; ++jj) {
                    const int q = q0 + jj;
                    const f16* aq = line + (int64_t)q * cstride;
                    if (!(edge_free && jj > 0 && jj < NQ - 1)) aq = (unsigned)q < (unsigned)p.Wo ? aq : zline;
                    raw[jj] = ld16(aq);
                }
                const WT* wt = s_w + tap * p.Cw + cw;
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) {
                    float wv[8];
                    dw_load_w8(wt + kw * p.Cw, wv);
#pragma unroll
                    for (int i = 0; i < SF_DW_WB; ++i) {
                        // kw = B0 + i - jj*SW  <=>  jj = (B0 + i - kw) / SW when divisible (compile-time)
                        const int num = B0 + i - kw;
                        if (num >= 0 && num % SW == 0 && num / SW < NQ) {
                            const int jj = num / SW;
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[i][e] += (float)raw[jj][e] * wv[e];
                        }
                    }
                }
            }
        }
        f16* xrow = p.y + ((int64_t)n * Si + p.cls + ((int64_t)t * p.Hi + h) * p.Wi + w0) * p.ldy + c;
#pragma unroll
        for (int i = 0; i < SF_DW_WB; ++i) {
            if (w0 + i < p.Wi) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (f16)acc[i][e];
                st16(xrow + (int64_t)i * p.ldy, o);
            }
        }
    }
}

// weight gradient, blocked over 4 consecutive OUTPUT columns; blockIdx.z = kt, kH*KW (<= 9) accumulators.
// Out-of-range taps are redirected to the zero line by ADDRESS (see the forward kernel) and the (kh) planes are visited uniformly.
template <int KW, int SW, int WSZ>
__global__ __launch_bounds__(SF_THREADS, 3) void sf_dwconv_wgrad_blocked_kernel(DwParams p, DwBlockIdx bi) {
    constexpr int NIN = (SF_DW_WB - 1) * SW + KW;
    __shared__ float s_red[SF_THREADS][9];
    const bool edge_free = (p.Wo % SF_DW_WB) == 0 && (p.Wo - SF_DW_WB) * SW - p.pW + NIN - 1 < p.Wi;   // block-uniform
    const f16* const zline = sf_dw_zero_line;
    int bz;
    const int bx = dw_block_id(p, &bz);
    int gcol, r0, r1, rstep;
    const bool active = p.rt.init(gcol, r0, r1, rstep, bx);
    const int c = gcol * 8;
    const int kt = bz;
    const int nsp = p.kH * KW;
    const int64_t Si = (int64_t)p.Ti * p.Hi * p.Wi + p.cls, So = (int64_t)p.To * p.Ho * p.Wo + p.cls;
    float acc[9][8];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
    if (active) {
        for (int m = r0; m < r1; m += rstep) {
            uint32_t n;
            int to, ho, wo0;
            if (dwb_decode(bi, (uint32_t)m, n, to, ho, wo0)) continue;
            const int t = to * p.sT - p.pT + kt;
            if ((unsigned)t >= (unsigned)p.Ti) continue;
            const f16* drow = p.dy + ((int64_t)n * So + p.cls + ((int64_t)to * p.Ho + ho) * p.Wo + wo0) * p.lddy + c;
            f16x8 d[SF_DW_WB];
#pragma unroll
            for (int i = 0; i < SF_DW_WB; ++i) {
                const f16* a = drow + (int64_t)i * p.lddy;
                if (!edge_free) a = wo0 + i < p.Wo ? a : zline;
                d[i] = ld16(a);
            }
            const f16* xb = p.x + ((int64_t)n * Si + p.cls) * p.ldx + c;
            const int wi0 = wo0 * SW - p.pW;
#pragma unroll
            for (int kh = 0; kh < 9 / KW; ++kh) {
                if (kh < p.kH) {
                    const int h = ho * p.sH - p.pH + kh;
                    {
                        const bool pv = (unsigned)h < (unsigned)p.Hi;
                        const f16* line = pv ? xb + (((int64_t)t * p.Hi + h) * p.Wi) * p.ldx : zline;
                        const int64_t cstride = pv ? (int64_t)p.ldx : 0;
                        f16x8 raw[NIN];
#pragma unroll
                        for (int j = 0; j < NIN; ++j) {
                            const int col = wi0 + j;
                            const f16* a = line + (int64_t)col * cstride;
                            if (!(edge_free && j > 0)) a = (unsigned)col < (unsigned)p.Wi ? a : zline;
                            raw[j] = ld16(a);
                        }
#pragma unroll
                        for (int j = 0; j < NIN; ++j) {
                            float xin[8];
                            cvt8(raw[j], xin);
#pragma unroll
                            for (int i = 0; i < SF_DW_WB; ++i) {
                                const int kw = j - i * SW;
                                if (kw >= 0 && kw < KW) {
#pragma unroll
                                    for (int e = 0; e < 8; ++e) acc[kh * KW + kw][e] += (float)d[i][e] * xin[e];
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    const int G = p.rt.C >> 3;
    const int TG = G < SF_THREADS ? G : SF_THREADS;
    const int rpi = SF_THREADS / TG;
    const int taps = p.kT * nsp;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        if (i < nsp) {
#pragma unroll
            for (int e = 0; e < 8; ++e) s_red[threadIdx.x][e] = acc[i][e];
            __syncthreads();
            if (active && (int)threadIdx.x < TG) {
                float* o = p.wpart + ((int64_t)bx * taps + kt * nsp + i) * p.rt.C + c;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float a = 0.f;
                    for (int k = 0; k < rpi; ++k) a += s_red[threadIdx.x + k * TG][e];
                    o[e] = a;
                }
            }
            __syncthreads();
        }
    }
}

// dw[cw][tap] (+)= scale * sum over blocks and over the C/Cw channel copies of wpart[blk][tap][c]
struct DwFinalizeParams {
    const float* wpart; int nblk, taps, C, Cw, Cwreal;
    float* dw; float scale; int accumulate;
    int ldc;            // row width of the table's tap rows (0: C); > C when the table holds two tensors' partials side by side
};
__global__ __launch_bounds__(SF_THREADS) void sf_dwconv_wgrad_finalize_kernel(DwFinalizeParams p) {
    // 32 outputs (tap, cw) per block x 8 segments of the (block, channel-copy) sum; fixed-order LDS fold
    __shared__ double s_acc[8][32];
    const int ox = threadIdx.x & 31, seg = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + ox;      // (tap, cw), cw fastest
    const bool ok = idx < p.taps * p.Cw;
    const int tap = ok ? idx / p.Cw : 0, cw = ok ? idx % p.Cw : 0;
    double s = 0.0;
    if (ok && cw < p.Cwreal) {
        const int copies = p.C / p.Cw;
        const int ldc = p.ldc > 0 ? p.ldc : p.C;
        const int64_t bs = (int64_t)p.taps * ldc;
        const float* src = p.wpart + (int64_t)tap * ldc + cw;
        // four independent partial sums (loads of four table rows in flight), folded in a fixed order
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int b = seg;
        for (; b + 24 < p.nblk; b += 32) {
            for (int cp = 0; cp < copies; ++cp) {
                const float v0 = src[(int64_t)b * bs + cp * p.Cw], v1 = src[(int64_t)(b + 8) * bs + cp * p.Cw];
                const float v2 = src[(int64_t)(b + 16) * bs + cp * p.Cw], v3 = src[(int64_t)(b + 24) * bs + cp * p.Cw];
                s0 += (double)v0; s1 += (double)v1; s2 += (double)v2; s3 += (double)v3;
            }
        }
        for (; b < p.nblk; b += 8)
            for (int cp = 0; cp < copies; ++cp) s0 += (double)src[(int64_t)b * bs + cp * p.Cw];
        s = (s0 + s1) + (s2 + s3);
    }
    s_acc[seg][ox] = s;
    __syncthreads();
    if (seg == 0 && ok && cw < p.Cwreal) {
        for (int k = 1; k < 8; ++k) s += s_acc[k][ox];
        float* dst = p.dw + (int64_t)cw * p.taps + tap;
        const float v = (float)(s * p.scale);
        *dst = p.accumulate ? *dst + v : v;
    }
}
