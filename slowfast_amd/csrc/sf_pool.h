// Spatial max-pool fused with the producer's BatchNorm+ReLU, layout conversion, weight repacking.
//
// Reference call sites: ResNetBasicStem conv -> bn -> relu -> MaxPool3d([1,3,3],[1,2,2],[0,1,1])
// (slowfast/models/stem_helper.py:182-201); input clips arrive NCTHW fp32 (tools/train_net.py:79-98).
#pragma once
#include "sf_common.h"

struct PoolParams {
    const f16* y; int ldy;          // fwd: raw conv output [N,T,H,W][C]
    const float* scale; const float* shift; int relu;   // producer BN (+ReLU); scale may be null
    int N, T, H, W, C;
    int Ho, Wo;
    int kH, kW, sH, sW, pH, pW;
    f16* out; int ldo;              // fwd: pooled [N,T,Ho,Wo][C]; bwd: g [N,T,H,W][C]
    uint8_t* argmax;                // fwd (optional out) / bwd (in): window-local index kh*kW+kw of the maximum,
                                    // [N,T,Ho,Wo][C] bytes
    const f16* pooled; int ldp;     // bwd: the forward's pooled output (ReLU mask: gradient flows iff pooled > 0)
    const f16* dout; int lddo;      // bwd: gradient of the pooled output
    FastDiv fdG, fdW, fdH;          // work index -> (group, w, h, rest); dims of the iterated space
    int64_t total;
    int cls;                        // token tensors: every sample's T*H*W rows are preceded by one cls row, which
    FastDiv fdT;                    // passes through the pool (attention.py:24-36); row += nt / T + 1
    FastDiv fdsH, fdsW;             // bwd: the strides (window index of a position)
};

// row of (nt = n*T + t, h, w) in a tensor whose spatial dims are (Hx, Wx)
__device__ __forceinline__ int64_t pool_row(const PoolParams& p, uint32_t nt, int Hx, int Wx, int h, int w) {
    int64_t r = ((int64_t)nt * Hx + h) * Wx + w;
    if (p.cls) r += fd_div(nt, p.fdT) + 1;
    return r;
}

__device__ __forceinline__ void bn_act8(const f16x8& v, const float (&sc)[8], const float (&sh)[8], int relu,
                                        float (&z)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float x = (float)v[e] * sc[e] + sh[e];
        if (relu) x = x > 0.f ? x : 0.f;
        z[e] = (float)(f16)x;  // the pooled tensor is stored in fp16: compare what is stored
    }
}

// (Round 4: an XCD-contiguous workgroup order -- what helped the depthwise stencils -- made the backward gather SLOWER here,
// 136 -> 174 us on the MViT skip pooling, profiles/r4/r4_v6_knobs_ab.txt; the plain order stays.)
// One thread per (pooled position, 8 channels).  The FIRST maximum in scan order wins (the element torch's
// max_pool3d records); its window-local index is kept for the backward pass.
__global__ __launch_bounds__(SF_THREADS) void sf_pool_fwd_kernel(PoolParams p) {
    const int64_t ncls = p.cls ? (int64_t)p.N * (p.C >> 3) : 0;
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < p.total + ncls;
         idx += (int64_t)gridDim.x * SF_THREADS) {
        if (idx >= p.total) {     // cls rows: copy
            const int64_t j = idx - p.total;
            const int G = p.C >> 3;
            const int64_t n = j / G;
            const int c = (int)(j % G) * 8;
            st16(p.out + n * ((int64_t)p.T * p.Ho * p.Wo + 1) * p.ldo + c,
                 ld16(p.y + n * ((int64_t)p.T * p.H * p.W + 1) * p.ldy + c));
            continue;
        }
        uint32_t q, gcol, wo, ho, nt;
        fd_divmod((uint32_t)idx, p.fdG, q, gcol);
        fd_divmod(q, p.fdW, q, wo);
        fd_divmod(q, p.fdH, nt, ho);
        const int c = gcol * 8;
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
        if (p.scale) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = p.scale[c + e]; sh[e] = p.shift[c + e]; }
        }
        float best[8];
        uint32_t arg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; arg[e] = 0; }
        for (int kh = 0; kh < p.kH; ++kh) {
            const int h = (int)ho * p.sH - p.pH + kh;
            if ((unsigned)h >= (unsigned)p.H) continue;
            for (int kw = 0; kw < p.kW; ++kw) {
                const int w = (int)wo * p.sW - p.pW + kw;
                if ((unsigned)w >= (unsigned)p.W) continue;
                f16x8 v = ld16(p.y + pool_row(p, nt, p.H, p.W, h, w) * p.ldy + c);
                float z[8];
                bn_act8(v, sc, sh, p.relu, z);
                const uint32_t code = (uint32_t)(kh * p.kW + kw);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool gt = z[e] > best[e];
                    best[e] = gt ? z[e] : best[e];
                    arg[e] = gt ? code : arg[e];
                }
            }
        }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)best[e];
        const int64_t orow = pool_row(p, nt, p.Ho, p.Wo, (int)ho, (int)wo);
        st16(p.out + orow * p.ldo + c, o);
        if (p.argmax) {
            // ReLU in front of the pool: a window whose maximum is not positive passes no gradient -- recorded HERE as code 0xFF
            // (no tap has it), so that the backward gathers read argmax + gradient only, not the pooled tensor (round 6)
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) arg[e] = best[e] > 0.f ? arg[e] : 0xFFu;
            }
            u32x2 pk;
            pk.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
            pk.y = arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24);
            *reinterpret_cast<u32x2*>(p.argmax + orow * p.C + c) = pk;
        }
    }
}

// Gather form of the max-pool(+ReLU) backward: one thread per (input position, 8 channels) visits the (at most
// ceil(kH/sH)*ceil(kW/sW)) windows that cover it and takes a window's gradient iff the recorded argmax is this
// position (a window the ReLU in front of the pool did not pass carries code 0xFF: sf_pool_fwd_kernel).
__global__ __launch_bounds__(SF_THREADS) void sf_pool_bwd_kernel(PoolParams p) {
    const int64_t ncls = p.cls ? (int64_t)p.N * (p.C >> 3) : 0;
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < p.total + ncls;
         idx += (int64_t)gridDim.x * SF_THREADS) {
        if (idx >= p.total) {     // cls rows: the gradient passes through
            const int64_t j = idx - p.total;
            const int G = p.C >> 3;
            const int64_t n = j / G;
            const int c = (int)(j % G) * 8;
            st16(p.out + n * ((int64_t)p.T * p.H * p.W + 1) * p.ldo + c,
                 ld16(p.dout + n * ((int64_t)p.T * p.Ho * p.Wo + 1) * p.lddo + c));
            continue;
        }
        uint32_t q, gcol, w, h, nt;
        fd_divmod((uint32_t)idx, p.fdG, q, gcol);
        fd_divmod(q, p.fdW, q, w);
        fd_divmod(q, p.fdH, nt, h);
        const int c = gcol * 8;
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = 0.f;
        // windows ho with ho*sH - pH <= h <= ho*sH - pH + kH - 1
        int ho_lo = ((int)h + p.pH - p.kH + p.sH) / p.sH; if ((int)h + p.pH - p.kH + 1 <= 0) ho_lo = 0;
        int ho_hi = ((int)h + p.pH) / p.sH; if (ho_hi > p.Ho - 1) ho_hi = p.Ho - 1;
        int wo_lo = ((int)w + p.pW - p.kW + p.sW) / p.sW; if ((int)w + p.pW - p.kW + 1 <= 0) wo_lo = 0;
        int wo_hi = ((int)w + p.pW) / p.sW; if (wo_hi > p.Wo - 1) wo_hi = p.Wo - 1;
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            const int khs = (int)h - (ho * p.sH - p.pH);
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                const int kws = (int)w - (wo * p.sW - p.pW);
                const uint32_t me = (uint32_t)(khs * p.kW + kws);
                const int64_t orow = pool_row(p, nt, p.Ho, p.Wo, ho, wo);
                const u32x2 pk = *reinterpret_cast<const u32x2*>(p.argmax + orow * p.C + c);
                const f16x8 d = ld16(p.dout + orow * p.lddo + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t a = ((e < 4 ? pk.x : pk.y) >> (8 * (e & 3))) & 0xffu;
                    g[e] += a == me ? (float)d[e] : 0.f;
                }
            }
        }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)g[e];
        st16(p.out + pool_row(p, nt, p.H, p.W, (int)h, (int)w) * p.ldo + c, o);
    }
}

// The same gather when at most TWO windows per dimension cover a position (kH <= 2 sH and kW <= 2 sW: every pooling layer of the
// model zoo -- 3x3 / stride 2 stems and skip paths).  The four candidate windows' operands (argmax bytes, gradient, pooled value)
// are requested TOGETHER, with clamped coordinates and validity as predicates: 8 loads in flight per thread (round 6: the pooled
// value is no longer read -- 40 % of the bytes a thread requested; the ReLU test lives in the argmax code).  The nested loops
// above have runtime bounds, so every window was its own load -> wait -> combine trip, behind two runtime divisions (the stem
// pool of SlowFast: 434 us for ~670 MB, round 5).  Same summation order (ho, then wo, ascending): bit-identical.
__global__ __launch_bounds__(SF_THREADS) void sf_pool_bwd4_kernel(PoolParams p) {
    const int64_t ncls = p.cls ? (int64_t)p.N * (p.C >> 3) : 0;
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < p.total + ncls;
         idx += (int64_t)gridDim.x * SF_THREADS) {
        if (idx >= p.total) {     // cls rows: the gradient passes through
            const int64_t j = idx - p.total;
            const int G = p.C >> 3;
            const int64_t n = j / G;
            const int c = (int)(j % G) * 8;
            st16(p.out + n * ((int64_t)p.T * p.H * p.W + 1) * p.ldo + c,
                 ld16(p.dout + n * ((int64_t)p.T * p.Ho * p.Wo + 1) * p.lddo + c));
            continue;
        }
        uint32_t q, gcol, w, h, nt;
        fd_divmod((uint32_t)idx, p.fdG, q, gcol);
        fd_divmod(q, p.fdW, q, w);
        fd_divmod(q, p.fdH, nt, h);
        const int c = gcol * 8;
        // candidate windows per dimension: the last one that starts at or before the position, and the one before it
        const int hh = (int)fd_div(h + (uint32_t)p.pH, p.fdsH), wh = (int)fd_div(w + (uint32_t)p.pW, p.fdsW);
        int hoc[2], woc[2], khs[2], kws[2];
        bool hv[2], wv[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int ho = hh - 1 + a, wo = wh - 1 + a;
            khs[a] = (int)h - (ho * p.sH - p.pH);
            kws[a] = (int)w - (wo * p.sW - p.pW);
            hv[a] = ho >= 0 && ho < p.Ho && khs[a] < p.kH;
            wv[a] = wo >= 0 && wo < p.Wo && kws[a] < p.kW;
            hoc[a] = ho < 0 ? 0 : (ho > p.Ho - 1 ? p.Ho - 1 : ho);
            woc[a] = wo < 0 ? 0 : (wo > p.Wo - 1 ? p.Wo - 1 : wo);
        }
        u32x2 pk[2][2];
        f16x8 d[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int64_t orow = pool_row(p, nt, p.Ho, p.Wo, hoc[a], woc[b]);
                pk[a][b] = *reinterpret_cast<const u32x2*>(p.argmax + orow * p.C + c);
                d[a][b] = ld16(p.dout + orow * p.lddo + c);
            }
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const bool on = hv[a] && wv[b];
                const uint32_t me = (uint32_t)(khs[a] * p.kW + kws[b]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t am = ((e < 4 ? pk[a][b].x : pk[a][b].y) >> (8 * (e & 3))) & 0xffu;
                    g[e] += on && am == me ? (float)d[a][b][e] : 0.f;
                }
            }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)g[e];
        st16(p.out + pool_row(p, nt, p.H, p.W, (int)h, (int)w) * p.ldo + c, o);
    }
}

// ------------------------------------------------------------------------------------------------
// MaxPool3d with kernel = stride, no padding (non-overlapping windows): nn.MaxPool3d(pool_size, pool_size, 0) in front
// of conv_phi / conv_g of the Nonlocal block (slowfast/models/nonlocal_helper.py:96-114).
struct Pool3dParams {
    int N, T, H, W, C, kT, kH, kW, To, Ho, Wo;
    const f16* x; int ldx;
    f16* out; int ldo;              // fwd: pooled; bwd: dx
    uint8_t* argmax;                // [N,To,Ho,Wo][C] window-local index (kt*kH + kh)*kW + kw of the first maximum
    const f16* dout; int lddo;
    FastDiv fdG, fdW, fdH, fdT;     // dims of the iterated space (fwd: output, bwd: input)
    int64_t total;
};
__global__ __launch_bounds__(SF_THREADS) void sf_pool3d_fwd_kernel(Pool3dParams p) {
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < p.total;
         idx += (int64_t)gridDim.x * SF_THREADS) {
        uint32_t q, gcol, wo, ho, to, n;
        fd_divmod((uint32_t)idx, p.fdG, q, gcol);
        fd_divmod(q, p.fdW, q, wo);
        fd_divmod(q, p.fdH, q, ho);
        fd_divmod(q, p.fdT, n, to);
        const int c = gcol * 8;
        float best[8];
        uint32_t arg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; arg[e] = 0; }
        uint32_t code = 0;
        for (int kt = 0; kt < p.kT; ++kt)
            for (int kh = 0; kh < p.kH; ++kh)
                for (int kw = 0; kw < p.kW; ++kw, ++code) {
                    const int64_t row = (((int64_t)n * p.T + to * p.kT + kt) * p.H + ho * p.kH + kh) * p.W + wo * p.kW + kw;
                    f16x8 v = ld16(p.x + row * p.ldx + c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float z = (float)v[e];
                        const bool gt = z > best[e];
                        best[e] = gt ? z : best[e];
                        arg[e] = gt ? code : arg[e];
                    }
                }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)best[e];
        const int64_t orow = (((int64_t)n * p.To + to) * p.Ho + ho) * p.Wo + wo;
        st16(p.out + orow * p.ldo + c, o);
        u32x2 pk;
        pk.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
        pk.y = arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24);
        *reinterpret_cast<u32x2*>(p.argmax + orow * p.C + c) = pk;
    }
}
__global__ __launch_bounds__(SF_THREADS) void sf_pool3d_bwd_kernel(Pool3dParams p) {
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < p.total;
         idx += (int64_t)gridDim.x * SF_THREADS) {
        uint32_t q, gcol, w, h, t, n;
        fd_divmod((uint32_t)idx, p.fdG, q, gcol);
        fd_divmod(q, p.fdW, q, w);
        fd_divmod(q, p.fdH, q, h);
        fd_divmod(q, p.fdT, n, t);
        const int c = gcol * 8;
        const int to = (int)t / p.kT, ho = (int)h / p.kH, wo = (int)w / p.kW;
        f16x8 o = zero8();
        if (to < p.To && ho < p.Ho && wo < p.Wo) {     // positions beyond the last full window receive no gradient
            const uint32_t me = (uint32_t)((((int)t - to * p.kT) * p.kH + ((int)h - ho * p.kH)) * p.kW + ((int)w - wo * p.kW));
            const int64_t orow = (((int64_t)n * p.To + to) * p.Ho + ho) * p.Wo + wo;
            const u32x2 pk = *reinterpret_cast<const u32x2*>(p.argmax + orow * p.C + c);
            const f16x8 d = ld16(p.dout + orow * p.lddo + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t a = ((e < 4 ? pk.x : pk.y) >> (8 * (e & 3))) & 0xffu;
                o[e] = a == me ? d[e] : (f16)0;
            }
        }
        st16(p.out + ((((int64_t)n * p.T + t) * p.H + h) * p.W + w) * p.ldo + c, o);
    }
}

// ------------------------------------------------------------------------------------------------
// NCTHW fp32 -> channels-last fp16 with the channel count zero-padded to Cp (multiple of 8)
__global__ __launch_bounds__(SF_THREADS) void sf_ncthw_to_cl_kernel(const float* x, f16* out, int N, int C, int64_t S,
                                                                    int Cp) {
    const int64_t total = (int64_t)N * S;
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * SF_THREADS) {
        const int64_t n = idx / S, s = idx - n * S;
        if (Cp == 4) {   // 3-channel clips for the W-pair-folded stem convolutions: 8 bytes per position
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = e < C ? (f16)x[(n * C + e) * S + s] : (f16)0;
            *reinterpret_cast<f16x4*>(out + idx * 4) = o;
            continue;
        }
        for (int cg = 0; cg < Cp; cg += 8) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = cg + e;
                o[e] = c < C ? (f16)x[(n * C + c) * S + s] : (f16)0;
            }
            st16(out + idx * Cp + cg, o);
        }
    }
}

// The same conversion for clips of at most 4 channels (every RGB input) when S is a multiple of 4: a lane owns FOUR consecutive
// positions -- one 16-byte load per channel plane (a wave reads 1 KB of a plane per instruction instead of 256 B) and 32 (CP = 4)
// or 64 (CP = 8) contiguous bytes of output.  Bit-identical to the kernel above (one fp32 -> 16-bit rounding per element).
template <int CP>
__global__ __launch_bounds__(SF_THREADS) void sf_ncthw_to_cl_quad_kernel(const float* x, f16* out, int N, int C, int64_t S4) {
    const int64_t total = (int64_t)N * S4;
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * SF_THREADS) {
        const int64_t n = idx / S4, q = idx - n * S4;
        const f32x4* src = reinterpret_cast<const f32x4*>(x) + (n * C) * S4 + q;
        f32x4 v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = e < C ? src[e * S4] : (f32x4){0.f, 0.f, 0.f, 0.f};
        f16* dst = out + idx * (4 * CP);
        if (CP == 4) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (f16)v[e][2 * h]; o[4 + e] = (f16)v[e][2 * h + 1]; }
                st16(dst + 8 * h, o);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f16x8 o = zero8();
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (f16)v[e][j];
                st16(dst + 8 * j, o);
            }
        }
    }
}

// channels-last fp16 (row pitch ld) -> NCTHW fp32
__global__ __launch_bounds__(SF_THREADS) void sf_cl_to_ncthw_kernel(const f16* x, int ld, float* out, int N, int C,
                                                                    int64_t S) {
    const int64_t total = (int64_t)N * C * S;
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * SF_THREADS) {
        const int64_t s = idx % S;
        const int64_t nc = idx / S;
        const int64_t c = nc % C, n = nc / C;
        out[idx] = (float)x[(n * S + s) * ld + c];
    }
}

// ------------------------------------------------------------------------------------------------
// Conv3d weight [Co][Cw][taps] fp32 -> forward operand  wf[Co][ldf]  with k = tap*Cp + ci
//                                   -> dgrad operand    wd[Cp][ldd]  with k = tap*Co + co
// (fp16, zero padded: ci >= Cw, k >= Ktot).
struct PrepParams {          // == sf_prep_item (include/sfamd.h): items of the batched launch live in device memory
    const float* w;
    f16* wf;
    f16* wd;
    int Co, Cow, Cw, Cp, taps;   // Cow = rows of w (real output channels), Co - Cow zero rows
    int ldf, ldd;
    int pad;                     // batched launch: output channels per brick (sf_prep_tile_co), 0 = element-wise blocks
};

__device__ __forceinline__ void prep_element(const PrepParams& p, int64_t idx, int64_t nf) {
    if (idx < nf) {
        const int co = (int)(idx / p.ldf), k = (int)(idx % p.ldf);
        const int tap = k / p.Cp, ci = k % p.Cp;
        float v = 0.f;
        if (tap < p.taps && ci < p.Cw && co < p.Cow) v = p.w[((int64_t)co * p.Cw + ci) * p.taps + tap];
        p.wf[idx] = (f16)v;
    } else {
        const int64_t j = idx - nf;
        const int ci = (int)(j / p.ldd), k = (int)(j % p.ldd);
        const int tap = k / p.Co, co = k % p.Co;
        float v = 0.f;
        if (tap < p.taps && ci < p.Cw && co < p.Cow) v = p.w[((int64_t)co * p.Cw + ci) * p.taps + tap];
        p.wd[j] = (f16)v;
    }
}

__global__ __launch_bounds__(SF_THREADS) void sf_prep_weights_kernel(PrepParams p) {
    const int64_t nf = (int64_t)p.Co * p.ldf;
    const int64_t nd = p.wd ? (int64_t)p.Cp * p.ldd : 0;
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < nf + nd;
         idx += (int64_t)gridDim.x * SF_THREADS)
        prep_element(p, idx, nf);
}

// All weights of a model in ONE launch (the per-layer launches of a training step are ~110 x 6 us of launch latency for
// ~50 us of memory traffic).  Workgroup b handles tile blk_off[b] of item blk_item[b]: a (tco output channels) x (32 input
// channels) x (all taps) brick of the fp32 weight is read with coalesced loads into LDS (fp16) and written out twice --
// tap-major rows of the forward operand (32 consecutive halfs per run) and of the data-gradient operand (tco consecutive
// halfs per run).  The element-wise form above reads the fp32 weight with a stride of taps (wf) or Cw*taps (wd) elements
// per lane: 16-32x over-fetch, 0.7 TB/s when batched.  Items whose taps do not fit the brick (item.pad == 0) fall back to
// 4096-element blocks of the element-wise form.
#define SF_PREP_BLOCK_ELEMS (SF_THREADS * 16)
#define SF_PREP_TILE_CI 32
#define SF_PREP_LDS_HALFS 4096
// output channels per brick (power of two <= 32; 0: brick does not fit, element-wise fallback)
static inline int sf_prep_tile_co(int taps) {
    const int row = SF_PREP_TILE_CI * taps + 2;
    int tco = 32;
    while (tco >= 1 && tco * row > SF_PREP_LDS_HALFS) tco >>= 1;
    return tco;
}

__global__ __launch_bounds__(SF_THREADS) void sf_prep_weights_batch_kernel(const PrepParams* items, const int32_t* blk_item,
                                                                          const int32_t* blk_off) {
    __shared__ f16 s_w[SF_PREP_LDS_HALFS];
    const PrepParams p = items[blk_item[blockIdx.x]];
    const int tid = threadIdx.x;
    if (p.pad == 0) {                       // element-wise fallback
        const int64_t nf = (int64_t)p.Co * p.ldf;
        const int64_t n = nf + (p.wd ? (int64_t)p.Cp * p.ldd : 0);
        const int64_t base = (int64_t)blk_off[blockIdx.x] * SF_PREP_BLOCK_ELEMS;
        for (int e = 0; e < 16; ++e) {
            const int64_t idx = base + tid + (int64_t)e * SF_THREADS;
            if (idx < n) prep_element(p, idx, nf);
        }
        return;
    }
    const int tco = p.pad, taps = p.taps;
    const int n_ci_tiles = (p.Cp + SF_PREP_TILE_CI - 1) / SF_PREP_TILE_CI;
    const int tile = blk_off[blockIdx.x];
    const int cot = tile / n_ci_tiles, cit = tile % n_ci_tiles;
    const int co0 = cot * tco, ci0 = cit * SF_PREP_TILE_CI;
    const int seg = SF_PREP_TILE_CI * taps;             // fp32 elements of one output channel inside the brick
    const int LD = seg + 2;
    int ci_real = p.Cw - ci0;                           // real input channels of this brick
    ci_real = ci_real < 0 ? 0 : (ci_real > SF_PREP_TILE_CI ? SF_PREP_TILE_CI : ci_real);
    int ci_buf = p.Cp - ci0;                            // channels of the operand buffers (zero padded)
    ci_buf = ci_buf > SF_PREP_TILE_CI ? SF_PREP_TILE_CI : ci_buf;
    const int L = ci_real * taps;
    // no divisions in the copy loops (the element-wise form spends its time in them): a 32-lane group walks one output channel
    const int lane32 = tid & 31, grp = tid >> 5;        // 8 groups of 32 lanes
    for (int r = grp; r < tco; r += SF_THREADS / 32) {
        const int co = co0 + r;
        const float* src = p.w + ((int64_t)co * p.Cw + ci0) * taps;
        const bool real = co < p.Cow;
        for (int j = lane32; j < seg; j += 32) s_w[r * LD + j] = (f16)((real && j < L) ? src[j] : 0.f);
    }
    __syncthreads();
    // forward operand: wf[co][tap * Cp + ci], 32 consecutive input channels per run
    if (lane32 < ci_buf)
        for (int r = grp; r < tco; r += SF_THREADS / 32) {
            const int co = co0 + r;
            if (co >= p.Co) break;
            f16* dst = p.wf + (int64_t)co * p.ldf + ci0 + lane32;
            const f16* srow = s_w + r * LD + lane32 * taps;
            for (int tap = 0; tap < taps; ++tap) dst[tap * p.Cp] = srow[tap];
        }
    if (cit == 0) {                                     // zero columns [taps * Cp, ldf)
        const int padf = p.ldf - taps * p.Cp;
        if (lane32 < padf)
            for (int r = grp; r < tco; r += SF_THREADS / 32)
                if (co0 + r < p.Co) p.wf[(int64_t)(co0 + r) * p.ldf + taps * p.Cp + lane32] = (f16)0.f;
    }
    if (!p.wd) return;
    // data-gradient operand: wd[ci][tap * Co + co], tco consecutive output channels per run
    {
        const int r = tid & (tco - 1), g = tid / tco, ng = SF_THREADS / tco;     // tco is a power of two
        const int co = co0 + r;
        if (co < p.Co)
            for (int cil = g; cil < ci_buf; cil += ng) {
                f16* dst = p.wd + (int64_t)(ci0 + cil) * p.ldd + co;
                const f16* srow = s_w + r * LD + cil * taps;
                for (int tap = 0; tap < taps; ++tap) dst[tap * p.Co] = srow[tap];
            }
    }
    if (cot == 0) {                                     // zero columns [taps * Co, ldd)
        const int padd = p.ldd - taps * p.Co;
        if (lane32 < padd)
            for (int c = grp; c < ci_buf; c += SF_THREADS / 32) p.wd[(int64_t)(ci0 + c) * p.ldd + taps * p.Co + lane32] = (f16)0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// Input side of the path: decoded uint8 frames -> normalised fp16 clip in the stem's layout, one
// launch per pathway.  Replaces tensor_normalize (slowfast/datasets/utils.py:278-297: x/255, - mean, / std, in that
// order, fp32) + the THWC -> CTHW permute (datasets/kinetics.py:375-408) + pack_pathway_output's temporal
// index_select / channel reversal (datasets/utils.py:78-111):
//   out[n][to][h][w][c] = ((frames[n][t_index[to]][h][w][s] / 255) - mean[s]) / std[s],  s = c (2 - c when reversed)
// out is the N,T,H,W,4 fp16 buffer that engine.StemConvUnit reads as W pairs.
struct PackClipParams {
    const unsigned char* frames;    // [N][Tin][H][W][3]
    int N, Tin, Tout;
    int64_t HW;
    const int* t_index;             // [Tout] source frame of every output frame (null: identity)
    float mean[3], stdv[3];
    int reverse;                    // DATA.REVERSE_INPUT_CHANNEL: channel c reads source channel 2 - c
    f16* out;
    int64_t total;                  // N*Tout*HW
    FastDiv fdHW, fdT;
};
__global__ __launch_bounds__(SF_THREADS) void sf_pack_clip_u8_kernel(PackClipParams p) {
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < p.total; idx += (int64_t)gridDim.x * SF_THREADS) {
        uint32_t q, hw, n, to;
        fd_divmod((uint32_t)idx, p.fdHW, q, hw);
        fd_divmod(q, p.fdT, n, to);
        const int ts = p.t_index ? p.t_index[to] : (int)to;
        const unsigned char* src = p.frames + (((int64_t)n * p.Tin + ts) * p.HW + hw) * 3;
        f16x4 o;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int sc = p.reverse ? 2 - c : c;           // normalisation happens before the channel reversal
            const float v = (float)src[sc] / 255.0f;
            o[c] = (f16)((v - p.mean[sc]) / p.stdv[sc]);
        }
        o[3] = (f16)0;
        *reinterpret_cast<f16x4*>(p.out + idx * 4) = o;
    }
}
