// Spatial max-pool fused with the producer's BatchNorm+ReLU, layout conversion, weight repacking.
//
// Reference call sites: ResNetBasicStem conv -> bn -> relu -> MaxPool3d([1,3,3],[1,2,2],[0,1,1])
// (slowfast/models/stem_helper.py:182-201); input clips arrive NCTHW fp32 (tools/train_net.py:79-98).
#pragma once
#include "sf_common.h"

struct PoolParams {
    const f16* y; int ldy;          // raw conv output [N,T,H,W][C]
    const float* scale; const float* shift; int relu;   // producer BN (+ReLU); scale may be null
    int N, T, H, W, C;
    int Ho, Wo;
    int kH, kW, sH, sW, pH, pW;
    f16* out; int ldo;              // fwd: pooled [N,T,Ho,Wo][C]; bwd: g [N,T,H,W][C]
    const f16* dout; int lddo;      // bwd: gradient of the pooled output
    FastDiv fdG, fdW, fdH;          // work index -> (group, w, h, rest); dims of the iterated space
    int64_t total;
};

__device__ __forceinline__ void bn_act8(const f16x8& v, const float (&sc)[8], const float (&sh)[8], int relu,
                                        float (&z)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float x = (float)v[e] * sc[e] + sh[e];
        if (relu) x = x > 0.f ? x : 0.f;
        z[e] = (float)(f16)x;  // forward stores fp16: compare what the forward compared
    }
}

__global__ __launch_bounds__(SF_THREADS) void sf_pool_fwd_kernel(PoolParams p) {
    const int G = p.C >> 3;
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < p.total;
         idx += (int64_t)gridDim.x * SF_THREADS) {
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
#pragma unroll
        for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
        for (int kh = 0; kh < p.kH; ++kh) {
            const int h = (int)ho * p.sH - p.pH + kh;
            if ((unsigned)h >= (unsigned)p.H) continue;
            for (int kw = 0; kw < p.kW; ++kw) {
                const int w = (int)wo * p.sW - p.pW + kw;
                if ((unsigned)w >= (unsigned)p.W) continue;
                f16x8 v = ld16(p.y + (((int64_t)nt * p.H + h) * p.W + w) * p.ldy + c);
                float z[8];
                bn_act8(v, sc, sh, p.relu, z);
#pragma unroll
                for (int e = 0; e < 8; ++e) best[e] = z[e] > best[e] ? z[e] : best[e];
            }
        }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)best[e];
        st16(p.out + (((int64_t)nt * p.Ho + ho) * p.Wo + wo) * p.ldo + c, o);
    }
}

// Gather form of the max-pool backward: one thread per (input position, 8 channels) decides, for
// each window that covers it, whether it is that window's FIRST maximum in scan order (the element
// torch's max_pool3d records), sums the matching output gradients and applies the ReLU mask.
__global__ __launch_bounds__(SF_THREADS) void sf_pool_bwd_kernel(PoolParams p) {
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < p.total;
         idx += (int64_t)gridDim.x * SF_THREADS) {
        uint32_t q, gcol, w, h, nt;
        fd_divmod((uint32_t)idx, p.fdG, q, gcol);
        fd_divmod(q, p.fdW, q, w);
        fd_divmod(q, p.fdH, nt, h);
        const int c = gcol * 8;
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
        if (p.scale) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = p.scale[c + e]; sh[e] = p.shift[c + e]; }
        }
        const f16* ybase = p.y + (int64_t)nt * p.H * p.W * p.ldy + c;
        float zs[8], g[8];
        bn_act8(ld16(ybase + ((int64_t)h * p.W + w) * p.ldy), sc, sh, p.relu, zs);
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
                bool isarg[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) isarg[e] = true;
                for (int kh = 0; kh < p.kH; ++kh) {
                    const int hh = ho * p.sH - p.pH + kh;
                    if ((unsigned)hh >= (unsigned)p.H) continue;
                    for (int kw = 0; kw < p.kW; ++kw) {
                        const int ww = wo * p.sW - p.pW + kw;
                        if ((unsigned)ww >= (unsigned)p.W) continue;
                        if (kh == khs && kw == kws) continue;
                        float zq[8];
                        bn_act8(ld16(ybase + ((int64_t)hh * p.W + ww) * p.ldy), sc, sh, p.relu, zq);
                        const bool before = (kh < khs) || (kh == khs && kw < kws);
#pragma unroll
                        for (int e = 0; e < 8; ++e) isarg[e] = isarg[e] && (before ? (zs[e] > zq[e]) : (zs[e] >= zq[e]));
                    }
                }
                f16x8 d = ld16(p.dout + (((int64_t)nt * p.Ho + ho) * p.Wo + wo) * p.lddo + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] += isarg[e] ? (float)d[e] : 0.f;
            }
        }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)((p.relu && !(zs[e] > 0.f)) ? 0.f : g[e]);
        st16(p.out + (((int64_t)nt * p.H + h) * p.W + w) * p.ldo + c, o);
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
struct PrepParams {
    const float* w;
    int Co, Cw, Cp, taps;
    f16* wf; int ldf;
    f16* wd; int ldd;
};

__global__ __launch_bounds__(SF_THREADS) void sf_prep_weights_kernel(PrepParams p) {
    const int64_t nf = (int64_t)p.Co * p.ldf;
    const int64_t nd = p.wd ? (int64_t)p.Cp * p.ldd : 0;
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < nf + nd;
         idx += (int64_t)gridDim.x * SF_THREADS) {
        if (idx < nf) {
            const int co = (int)(idx / p.ldf), k = (int)(idx % p.ldf);
            const int tap = k / p.Cp, ci = k % p.Cp;
            float v = 0.f;
            if (tap < p.taps && ci < p.Cw) v = p.w[((int64_t)co * p.Cw + ci) * p.taps + tap];
            p.wf[idx] = (f16)v;
        } else {
            const int64_t j = idx - nf;
            const int ci = (int)(j / p.ldd), k = (int)(j % p.ldd);
            const int tap = k / p.Co, co = k % p.Co;
            float v = 0.f;
            if (tap < p.taps && ci < p.Cw) v = p.w[((int64_t)co * p.Cw + ci) * p.taps + tap];
            p.wd[j] = (f16)v;
        }
    }
}
