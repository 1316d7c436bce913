// Depthwise TEMPORAL convolution (kT, 1, 1), stride 1, padding kT / 2 on channels-last rows (round 6): one pass over the tensor.
//
// Reference call site: X3DStem.conv (slowfast/models/stem_helper.py:262-285): Conv3d(dim_out, dim_out, [5, 1, 1], padding [2, 0, 0],
// groups = dim_out) behind the (1, 3, 3) spatial convolution, on 24 channels x 16 frames x 112 x 112 positions -- 1.23 GB in +
// out per pass at batch 64, the largest single tensor of X3D-M.  Forward, data gradient, weight gradient.
//
// The W-blocked stencils of sf_dwconv.h (written for 3 x 3 x 3 windows) re-read every input frame kT times through the
// vector-memory path and decode a row into (n, t, h, w) per output: 524 / 438 / 921 us for the three directions of the X3D-M
// stem against a 270 us stream (profiles/r6_v20_x3d_kernel_stats.md).  A temporal window has no spatial halo, so here
//   * a thread owns ONE (sample, position, 8-channel group) column and walks it along t with the kT frames of its window in
//     registers: every input element is loaded once (16 bytes per lane, the lanes of a wave contiguous in memory), two frames
//     ahead of its first use; fp32 weights (the nn.Conv3d parameter itself) are loop invariants in registers;
//   * the data gradient is the same walk with the taps mirrored; the weight gradient walks x and dy together into kT x 8 fp32
//     accumulators per thread;
//   * BatchNorm partial sums (forward) and the weight-gradient partials are folded over the threads of a workgroup through LDS
//     in a fixed order: one partial row per workgroup, summed by the existing finalize kernels.
#pragma once
#include "sf_common.h"

struct DwTempParams {
    const f16* a; int lda;              // x (forward, weight gradient) / dy (data gradient): rows (n, t, hw), C channels
    const f16* b; int ldb;              // weight gradient: dy
    f16* dst; int ldd;                  // y / dx
    const float* w;                     // [Cwreal][kT] fp32
    float* part;                        // forward: optional [blocks][2][C] sum / sum of squares; weight gradient: [blocks][kT][C]
    int N, C, Cw, Cwreal, T, HW;
    int G, PP, iters;                   // 8-channel groups per position, positions per workgroup pass, passes per workgroup
    int cols;                           // N * HW
    int flip;                           // 1: taps mirrored (data gradient)
    FastDiv fdHW, fdG;
};

// ordered fold of per-thread 8-float vectors over the PP positions of a workgroup: out[g * 8 + e] = sum_pp v[pp * G + g][e]
__device__ __forceinline__ void dwt_fold8(const float (&v)[8], float (*s_red)[9], float* out, int G, int PP, bool sync_first) {
    if (sync_first) __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) s_red[threadIdx.x][e] = v[e];
    __syncthreads();
    for (int i = threadIdx.x; i < G * 8; i += SF_THREADS) {
        const int g = i >> 3, e = i & 7;
        float s = 0.f;
        for (int pp = 0; pp < PP; ++pp) s += s_red[pp * G + g][e];
        out[i] = s;
    }
}

// MODE 0: forward / data gradient (STATS: BatchNorm partial sums of the fp32 results); MODE 2: weight gradient
template <int KT, int MODE, bool STATS>
__global__ __launch_bounds__(SF_THREADS) void sf_dwtemporal_kernel(DwTempParams p) {
    constexpr int PT = KT / 2, LA = 2;
    __shared__ float s_red[SF_THREADS][9];
    const int tid = threadIdx.x;
    uint32_t upp, ug;
    fd_divmod((uint32_t)tid, p.fdG, upp, ug);
    const int pp = (int)upp, g = (int)ug;
    const bool lane_on = pp < p.PP;
    const int c = g * 8;
    const int64_t fsa = (int64_t)p.HW * p.lda;                         // frame stride of a (elements)

    float wr[MODE == 2 ? 1 : KT][8];
    if constexpr (MODE != 2) {
        const int cw = c % p.Cw;
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                wr[k][e] = (lane_on && cw + e < p.Cwreal) ? p.w[(int64_t)(cw + e) * KT + (p.flip ? KT - 1 - k : k)] : 0.f;
    }
    float ssum[8], ssq[8];
    float wacc[MODE == 2 ? KT : 1][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
    if constexpr (MODE == 2) {
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) wacc[k][e] = 0.f;
    }

    for (int it = 0; it < p.iters; ++it) {
        const int q = (blockIdx.x * p.iters + it) * p.PP + pp;
        if (!lane_on || q >= p.cols) continue;                          // no barrier inside the walk
        uint32_t n, pos;
        fd_divmod((uint32_t)q, p.fdHW, n, pos);
        const int64_t row0 = (int64_t)n * p.T * p.HW + pos;             // row of frame 0
        const f16* ap = p.a + row0 * p.lda + c;
        // after the shift + insert at the top of step t: win[k] = x[t - PT + k] (frames < 0 and >= T are zeros).  Before the
        // loop slot PT + 1 + j therefore holds frame j < PT, the queue holds frames PT, PT + 1 (requested LA steps ahead)
        f16x8 win[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) win[k] = zero8();
#pragma unroll
        for (int j = 0; j < PT; ++j)
            if (j < p.T) win[PT + 1 + j] = ld16(ap + (int64_t)j * fsa);
        f16x8 qa[LA];
#pragma unroll
        for (int j = 0; j < LA; ++j) qa[j] = (PT + j < p.T) ? ld16(ap + (int64_t)(PT + j) * fsa) : zero8();
        const f16* bp = nullptr;
        f16x8 qb[LA];
        if constexpr (MODE == 2) {
            bp = p.b + row0 * p.ldb + c;
#pragma unroll
            for (int j = 0; j < LA; ++j) qb[j] = (j < p.T) ? ld16(bp + (int64_t)j * (int64_t)p.HW * p.ldb) : zero8();
        }
        f16* dp = MODE == 2 ? nullptr : p.dst + row0 * p.ldd + c;
        for (int t = 0; t < p.T; ++t) {
#pragma unroll
            for (int k = 0; k < KT - 1; ++k) win[k] = win[k + 1];
            win[KT - 1] = qa[0];
#pragma unroll
            for (int j = 0; j < LA - 1; ++j) qa[j] = qa[j + 1];
            const int tn = t + PT + LA;
            qa[LA - 1] = tn < p.T ? ld16(ap + (int64_t)tn * fsa) : zero8();
            if constexpr (MODE == 2) {
                const f16x8 d = qb[0];
#pragma unroll
                for (int j = 0; j < LA - 1; ++j) qb[j] = qb[j + 1];
                const int tb = t + LA;
                qb[LA - 1] = tb < p.T ? ld16(bp + (int64_t)tb * (int64_t)p.HW * p.ldb) : zero8();
#pragma unroll
                for (int k = 0; k < KT; ++k)
#pragma unroll
                    for (int e = 0; e < 8; ++e) wacc[k][e] += (float)d[e] * (float)win[k][e];
            } else {
                float acc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
                for (int k = 0; k < KT; ++k)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += (float)win[k][e] * wr[k][e];
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o[e] = (f16)acc[e];
                    if constexpr (STATS) { ssum[e] += acc[e]; ssq[e] += acc[e] * acc[e]; }
                }
                st16(dp + (int64_t)t * (int64_t)p.HW * p.ldd, o);
            }
        }
    }

    if constexpr (MODE == 2) {
        float* out = p.part + (int64_t)blockIdx.x * KT * p.C;
#pragma unroll
        for (int k = 0; k < KT; ++k) dwt_fold8(wacc[k], s_red, out + (int64_t)k * p.C, p.G, p.PP, k > 0);
    } else if constexpr (STATS) {
        float* out = p.part + (int64_t)blockIdx.x * 2 * p.C;
        dwt_fold8(ssum, s_red, out, p.G, p.PP, false);
        dwt_fold8(ssq, s_red, out + p.C, p.G, p.PP, true);
    }
}
