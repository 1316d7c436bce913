// RoI head kernels (AVA detection): temporal average pool, ROIAlign fused with the 7x7 max-pool, and their backward.
//
// Reference call sites: slowfast/models/head_helper.py:85-97 (AvgPool3d([T,1,1]) -> ROIAlign(resolution,
// spatial_scale=1/16, sampling_ratio=0, aligned) -> MaxPool2d(resolution)) and :116-133 (forward).  ROIAlign itself is
// detectron2.layers.ROIAlign (= torchvision.ops.roi_align), not vendored in the reference: the kernels follow the
// published algorithm as restated in oracle/video_ref.py:roi_align (continuous coordinates, optional -0.5 alignment,
// adaptive ceil(roi/bins) sampling grid, bilinear samples, zero outside [-1, size]).
//
// The 49 bins of a (roi, channel) are reduced to their maximum in registers, so the [R, C, 7, 7] tensor never exists;
// the backward routes the gradient to the samples of the arg-max bin only.
#pragma once
#include "sf_common.h"

// mean over T of channels-last activations: x [B][T][HW][C] fp16 -> m [B][HW][C] fp32; backward broadcasts dm / T
struct TMeanParams {
    const f16* x; int ldx;
    float* m;                       // [B*HW][C]
    const float* dm; f16* dx; int lddx;
    int T, C;
    int64_t HW;
    int64_t total;                  // B*HW*(C/8)
    FastDiv fdG, fdHW;
};
__global__ __launch_bounds__(SF_THREADS) void sf_tmean_fwd_kernel(TMeanParams p) {
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < p.total; idx += (int64_t)gridDim.x * SF_THREADS) {
        uint32_t row, g8, b, hw;
        fd_divmod((uint32_t)idx, p.fdG, row, g8);
        fd_divmod(row, p.fdHW, b, hw);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int t = 0; t < p.T; ++t) {
            const f16x8 v = ld16(p.x + (((int64_t)b * p.T + t) * p.HW + hw) * p.ldx + g8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        }
        float* o = p.m + (int64_t)row * p.C + g8 * 8;
        const float inv = 1.f / (float)p.T;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = acc[e] * inv;
    }
}
__global__ __launch_bounds__(SF_THREADS) void sf_tmean_bwd_kernel(TMeanParams p) {
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < p.total; idx += (int64_t)gridDim.x * SF_THREADS) {
        uint32_t row, g8, b, hw;
        fd_divmod((uint32_t)idx, p.fdG, row, g8);
        fd_divmod(row, p.fdHW, b, hw);
        const float* src = p.dm + (int64_t)row * p.C + g8 * 8;
        const float inv = 1.f / (float)p.T;
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)(src[e] * inv);
        for (int t = 0; t < p.T; ++t) st16(p.dx + (((int64_t)b * p.T + t) * p.HW + hw) * p.lddx + g8 * 8, o);
    }
}

struct RoiParams {
    const float* m;                 // [B][H][W][C] fp32
    float* dm;                      // backward: same shape, pre-zeroed
    const float* rois;              // [R][5] = batch index, x1, y1, x2, y2 (input pixels)
    int R, B, H, W, C;
    int res;                        // output bins per axis (7)
    float scale;                    // 1 / 16
    int aligned;
    float* out;                     // [R][ldo] max over the res*res bins
    int ldo, col0;                  // this pathway writes columns [col0, col0 + C)
    unsigned char* arg;             // [R][C] arg-max bin
    const float* dout; int lddo;    // backward: gradient w.r.t. out
};

struct RoiGeom {
    int b;
    float y1, x1, bh, bw;
    int gh, gw;
    float inv_count;
};
__device__ __forceinline__ RoiGeom roi_geom(const RoiParams& p, int r) {
    RoiGeom g;
    const float* q = p.rois + (int64_t)r * 5;
    g.b = (int)q[0];
    const float off = p.aligned ? 0.5f : 0.f;
    g.x1 = q[1] * p.scale - off;
    g.y1 = q[2] * p.scale - off;
    float rw = q[3] * p.scale - off - g.x1, rh = q[4] * p.scale - off - g.y1;
    if (!p.aligned) {
        rw = fmaxf(rw, 1.f);
        rh = fmaxf(rh, 1.f);
    }
    g.bh = rh / (float)p.res;
    g.bw = rw / (float)p.res;
    g.gh = (int)ceilf(rh / (float)p.res);
    g.gw = (int)ceilf(rw / (float)p.res);
    const int cnt = g.gh * g.gw;
    g.inv_count = 1.f / (float)(cnt > 1 ? cnt : 1);
    return g;
}
// bilinear sample position -> the four taps and weights (zero weights outside [-1, size])
struct RoiTap { int y0, y1, x0, x1; float w00, w01, w10, w11; };
__device__ __forceinline__ RoiTap roi_tap(float y, float x, int H, int W) {
    RoiTap t;
    const bool ok = !(y < -1.f || y > (float)H || x < -1.f || x > (float)W);
    y = fmaxf(y, 0.f);
    x = fmaxf(x, 0.f);
    t.y0 = (int)y;
    t.x0 = (int)x;
    if (t.y0 >= H - 1) { t.y0 = t.y1 = H - 1; y = (float)t.y0; } else t.y1 = t.y0 + 1;
    if (t.x0 >= W - 1) { t.x0 = t.x1 = W - 1; x = (float)t.x0; } else t.x1 = t.x0 + 1;
    const float ly = y - (float)t.y0, lx = x - (float)t.x0, hy = 1.f - ly, hx = 1.f - lx;
    const float k = ok ? 1.f : 0.f;
    t.w00 = hy * hx * k; t.w01 = hy * lx * k; t.w10 = ly * hx * k; t.w11 = ly * lx * k;
    return t;
}

// one thread per (roi, channel): max over the bins of the bin averages
__global__ __launch_bounds__(SF_THREADS) void sf_roi_align_max_fwd_kernel(RoiParams p) {
    const int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x;
    if (idx >= (int64_t)p.R * p.C) return;
    const int r = (int)(idx / p.C), c = (int)(idx % p.C);
    const RoiGeom g = roi_geom(p, r);
    const float* f = p.m + (int64_t)g.b * p.H * p.W * p.C + c;
    float best = -INFINITY;
    int best_bin = 0;
    for (int ph = 0; ph < p.res; ++ph)
        for (int pw = 0; pw < p.res; ++pw) {
            float acc = 0.f;
            for (int iy = 0; iy < g.gh; ++iy) {
                const float y = g.y1 + ph * g.bh + (iy + 0.5f) * g.bh / (float)g.gh;
                for (int ix = 0; ix < g.gw; ++ix) {
                    const float x = g.x1 + pw * g.bw + (ix + 0.5f) * g.bw / (float)g.gw;
                    const RoiTap t = roi_tap(y, x, p.H, p.W);
                    acc += t.w00 * f[((int64_t)t.y0 * p.W + t.x0) * p.C] + t.w01 * f[((int64_t)t.y0 * p.W + t.x1) * p.C] +
                           t.w10 * f[((int64_t)t.y1 * p.W + t.x0) * p.C] + t.w11 * f[((int64_t)t.y1 * p.W + t.x1) * p.C];
                }
            }
            acc *= g.inv_count;
            if (acc > best) { best = acc; best_bin = ph * p.res + pw; }
        }
    p.out[(int64_t)r * p.ldo + p.col0 + c] = best;
    p.arg[idx] = (unsigned char)best_bin;
}

// gradient of the arg-max bin scattered to its bilinear taps (fp32 atomics: several rois may share a pixel)
__global__ __launch_bounds__(SF_THREADS) void sf_roi_align_max_bwd_kernel(RoiParams p) {
    const int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x;
    if (idx >= (int64_t)p.R * p.C) return;
    const int r = (int)(idx / p.C), c = (int)(idx % p.C);
    const RoiGeom g = roi_geom(p, r);
    float* d = p.dm + (int64_t)g.b * p.H * p.W * p.C + c;
    const int bin = p.arg[idx], ph = bin / p.res, pw = bin % p.res;
    const float gr = p.dout[(int64_t)r * p.lddo + p.col0 + c] * g.inv_count;
    for (int iy = 0; iy < g.gh; ++iy) {
        const float y = g.y1 + ph * g.bh + (iy + 0.5f) * g.bh / (float)g.gh;
        for (int ix = 0; ix < g.gw; ++ix) {
            const float x = g.x1 + pw * g.bw + (ix + 0.5f) * g.bw / (float)g.gw;
            const RoiTap t = roi_tap(y, x, p.H, p.W);
            atomicAdd(d + ((int64_t)t.y0 * p.W + t.x0) * p.C, gr * t.w00);
            atomicAdd(d + ((int64_t)t.y0 * p.W + t.x1) * p.C, gr * t.w01);
            atomicAdd(d + ((int64_t)t.y1 * p.W + t.x0) * p.C, gr * t.w10);
            atomicAdd(d + ((int64_t)t.y1 * p.W + t.x1) * p.C, gr * t.w11);
        }
    }
}
