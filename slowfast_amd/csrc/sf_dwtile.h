// Depthwise 3x3x3 convolution on channels-last token / feature rows, LDS-tiled plane sweep (round 4).
//
// Reference call sites: the attention pooling convolutions pool_q / pool_k / pool_v of MViT (slowfast/models/attention.py:13-45,
// 227-266: Conv3d(head_dim, head_dim, 3x3x3, stride (1,s,s), padding 1, groups=head_dim) per head with the cls token routed
// around it) and their autograd backward.
//
// Why: the W-blocked stencils of sf_dwconv.h read every input element 13.5 times through the vector-memory path and convert it
// from fp16 each time -- 60 VALU lane-operations per output element, 129 VALU instructions per (kt, kh) plane for 48 packed
// FMAs (profiles/r4_v3_pmc_tokens.md, r4_v4: 57 us for a 15 us stream even with the window re-reads served by one L2).
// Here a workgroup sweeps the T frames of one (sample, row tile, 32-channel chunk):
//   * every input plane tile (with its halo, zero-filled outside the image) is fetched ONCE, converted to fp32 ONCE and kept
//     in LDS; the 9 (kh, kw) neighbours of an output position are ds_read_b128 pairs at compile-time-constant offsets -- no
//     per-tap address arithmetic, no range checks, no selects;
//   * an input plane contributes to the three output planes t = tin - kt + 1 at once: three accumulator sets live in registers
//     and rotate as the sweep advances, so the plane is read from LDS once for all three temporal taps;
//   * the arithmetic is unchanged: fp16 operands, fp32 products and accumulation (packed fp32 FMAs), one rounding at the store.
// The same kernel is the data gradient: for stride 1 it is the correlation with the flipped weights; for stride 2 the output
// gradient is staged zero-upsampled (values on the even grid positions), which turns the strided transpose convolution into the
// same stride-1 sweep -- three quarters of its products are zeros, still ~2x fewer instructions than the gather stencil.
#pragma once
#include "sf_common.h"

#define SF_DWT_CC 32                            // channels per workgroup
#define SF_DWT_G (SF_DWT_CC / 8)                // 16-byte channel groups per position
#define SF_DWT_PT (SF_THREADS / SF_DWT_G)       // position threads per workgroup
#define SF_DWT_PP 48                            // floats per staged position: 32 channels as [half][channel group][4] + 16 of
                                                // padding -- 192 B, so that the 16 lanes of every ds_read_b128 group (positions
                                                // p, p+3, p+5, p+6 x 4 channel groups) hit 16 distinct 16-byte bank slots
#define SF_DWT_PLANE 12288                      // floats of LDS for one plane tile (48 KiB)
#define SF_DWT_VPT 6                            // 16-byte vectors a thread stages per plane (rows * cols * G <= 256 * VPT)
#define SF_DWT_NPMAX 4                          // output positions per thread

struct DwTileParams {
    const f16* src; int ld_src;         // staged operand: x (forward) / dy (data gradient), rows (n, [cls], t, h, w)
    f16* dst; int ld_dst;               // y / dx
    const float* w;                     // [Cwreal][27] fp32 (the nn.Conv3d parameter)
    int N, C, Cw, Cwreal, cls, T;
    int Hs, Ws;                         // source plane extents
    int Hd, Wd;                         // destination plane extents
    int Hg, Wg;                         // extents of the staged grid (== source; zero-upsampled mode: == destination)
    int ups;                            // 1: grid (a, b) holds src(a / 2, b / 2) on even (a, b), zero elsewhere
    int s;                              // read stride of the sweep over the grid (forward stride; 1 for data gradients)
    int flip;                           // 1: taps are used mirrored (data gradient)
    int TH, RT, CT;                     // output rows per tile; staged rows = TH * s + 2, staged columns = Wg + 2
    int tiles_h, nchunks;
    FastDiv fdCT, fdWd, fdG;
};

__device__ __forceinline__ void dwt_cvt8(const f16x8& v, float* o) {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
}

template <int NP>
__global__ __launch_bounds__(SF_THREADS) void sf_dwtile_kernel(DwTileParams p) {
    __shared__ __attribute__((aligned(16))) float s_plane[SF_DWT_PLANE];
    __shared__ __attribute__((aligned(16))) float s_w[27 * SF_DWT_CC];
    const int tid = threadIdx.x;
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int chunk = (int)(bid % (uint32_t)p.nchunks);
    const int tile = (int)((bid / (uint32_t)p.nchunks) % (uint32_t)p.tiles_h);
    const int n = (int)(bid / (uint32_t)(p.nchunks * p.tiles_h));
    const int c0 = chunk * SF_DWT_CC;
    const int r0 = tile * p.TH;                                 // first destination row of the tile
    const int64_t Ss = (int64_t)p.T * p.Hs * p.Ws + p.cls, Sd = (int64_t)p.T * p.Hd * p.Wd + p.cls;

    // weights of this chunk, [tap][channel], mirrored for the data gradient; channel c uses weight row c % Cw
    for (int i = tid; i < 27 * SF_DWT_CC; i += SF_THREADS) {
        const int tap = i / SF_DWT_CC, c = i % SF_DWT_CC;
        const int cw = (c0 + c) % p.Cw;
        s_w[i] = cw < p.Cwreal ? p.w[cw * 27 + (p.flip ? 26 - tap : tap)] : 0.f;
    }
    if (p.cls && tile == 0 && tid < SF_DWT_G)                  // the cls row passes through
        st16(p.dst + (int64_t)n * Sd * p.ld_dst + c0 + tid * 8, ld16(p.src + (int64_t)n * Ss * p.ld_src + c0 + tid * 8));

    // ---- staging map: vector v = tid + 256 * u of a plane tile -> (tile row i, tile column j, channel group)
    const int V = p.RT * p.CT * SF_DWT_G;
    int st_off[SF_DWT_VPT];                 // element offset inside a source plane (row-major positions x ld), -1: zero
    int st_lds[SF_DWT_VPT];                 // float offset in s_plane, -1: no such vector
#pragma unroll
    for (int u = 0; u < SF_DWT_VPT; ++u) {
        const int v = tid + SF_THREADS * u;
        st_off[u] = -1;
        st_lds[u] = -1;
        if (v < V) {
            uint32_t pos, cgv, i, j;
            fd_divmod((uint32_t)v, p.fdG, pos, cgv);
            fd_divmod(pos, p.fdCT, i, j);
            st_lds[u] = (int)pos * SF_DWT_PP + (int)cgv * 4;
            const int gr = r0 * p.s - 1 + (int)i, gc = (int)j - 1;
            bool ok = (unsigned)gr < (unsigned)p.Hg && (unsigned)gc < (unsigned)p.Wg;
            int sr = gr, sc = gc;
            if (p.ups) {
                ok = ok && !((gr | gc) & 1);
                sr = gr >> 1;
                sc = gc >> 1;
                ok = ok && sr < p.Hs && sc < p.Ws;
            }
            if (ok) st_off[u] = (sr * p.Ws + sc) * p.ld_src + c0 + (int)cgv * 8;
        }
    }
    const f16* const src_n = p.src + ((int64_t)n * Ss + p.cls) * p.ld_src;
    const int64_t plane_src = (int64_t)p.Hs * p.Ws * p.ld_src;
    f16x8 pre[SF_DWT_VPT];
    auto prefetch = [&](int t) {
        const f16* base = src_n + (int64_t)t * plane_src;
#pragma unroll
        for (int u = 0; u < SF_DWT_VPT; ++u) pre[u] = st_off[u] >= 0 ? ld16(base + st_off[u]) : zero8();
    };

    // ---- compute map: thread = (position thread pt, channel group cg); positions q = pt + PT * k of the TH x Wd tile
    const int cg = tid % SF_DWT_G, pt = tid / SF_DWT_G;
    const int P = p.TH * p.Wd;
    int lbase[NP], doff[NP];
    bool pok[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int q = pt + SF_DWT_PT * k;
        uint32_t r, wq;
        fd_divmod((uint32_t)q, p.fdWd, r, wq);
        pok[k] = q < P && r0 + (int)r < p.Hd;
        if (!pok[k]) { r = 0; wq = 0; }
        lbase[k] = ((int)r * p.s * p.CT + (int)wq * p.s) * SF_DWT_PP + cg * 4;
        doff[k] = ((r0 + (int)r) * p.Wd + (int)wq) * p.ld_dst + c0 + cg * 8;
    }
    f16* const dst_n = p.dst + ((int64_t)n * Sd + p.cls) * p.ld_dst;
    const int64_t plane_dst = (int64_t)p.Hd * p.Wd * p.ld_dst;

    float acc[3][NP][8];                    // [0]: output plane tin - 1, [1]: tin, [2]: tin + 1
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[a][k][e] = 0.f;
    auto emit = [&](int t) {                // acc[0] -> destination plane t
        f16* base = dst_n + (int64_t)t * plane_dst;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if (!pok[k]) continue;
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (f16)acc[0][k][e];
            st16(base + doff[k], o);
        }
    };

    prefetch(0);
    const int rowf = p.CT * SF_DWT_PP;      // floats per staged row
    for (int tin = 0; tin < p.T; ++tin) {
        __syncthreads();                    // every thread is done reading plane tin - 1 (and the weights are staged)
#pragma unroll
        for (int u = 0; u < SF_DWT_VPT; ++u) {
            if (st_lds[u] >= 0) {
                float f[8];
                dwt_cvt8(pre[u], f);
                *reinterpret_cast<f32x4*>(s_plane + st_lds[u]) = (f32x4){f[0], f[1], f[2], f[3]};
                *reinterpret_cast<f32x4*>(s_plane + st_lds[u] + 16) = (f32x4){f[4], f[5], f[6], f[7]};
            }
        }
        __syncthreads();
        if (tin + 1 < p.T) prefetch(tin + 1);          // in flight under the sweep of plane tin
        // One (kh, kw) tap per iteration of a ROLLED loop: the neighbour of every position (NP x 8 floats) and the three
        // temporal weights of the tap (24 floats) are read from LDS, 3 x NP x 4 packed FMAs follow.  Unrolled, hipcc hoists all
        // 27 x 8 weights and 9 x NP x 8 neighbours of a plane above the FMAs (256 VGPRs + 102 AGPRs at NP = 1: one wave per
        // SIMD); the weights are also loop-invariant over the planes, so their address is laundered once per plane.
        int wz = cg * 8;
        SF_CONSUME_V(wz);
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - 3 * kh;
            const int doffs = kh * rowf + kw * SF_DWT_PP;
            float d[NP][8];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const float* src = s_plane + lbase[k] + doffs;
                const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) { d[k][e] = a[e]; d[k][4 + e] = b[e]; }
            }
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                const float* wp = s_w + wz + (kt * 9 + tap) * SF_DWT_CC;
                const f32x4 wa = *reinterpret_cast<const f32x4*>(wp), wb = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
                for (int k = 0; k < NP; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[2 - kt][k][e] += d[k][e] * wa[e];
                        acc[2 - kt][k][4 + e] += d[k][4 + e] * wb[e];
                    }
            }
        }
        if (tin >= 1) emit(tin - 1);
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc[0][k][e] = acc[1][k][e];
                acc[1][k][e] = acc[2][k][e];
                acc[2][k][e] = 0.f;
            }
    }
    emit(p.T - 1);
}
