// Depthwise 3x3x3 convolution on channels-last token / feature rows, LDS-tiled plane sweep (round 4).
//
// Reference call sites: the attention pooling convolutions pool_q / pool_k / pool_v of MViT (slowfast/models/attention.py:13-45,
// 227-266: Conv3d(head_dim, head_dim, 3x3x3, stride (1,s,s), padding 1, groups=head_dim) per head with the cls token routed
// around it) and their autograd backward.
//
// Why: the W-blocked stencils of sf_dwconv.h read every input element 13.5 times through the vector-memory path and convert it
// from fp16 each time -- 60 VALU lane-operations per output element, 129 VALU instructions per (kt, kh) plane for 48 packed
// FMAs (profiles/r4/r4_v3_pmc_tokens.md, r4_v4: 57 us for a 15 us stream even with the window re-reads served by one L2).
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
                f32x4* const q = reinterpret_cast<f32x4*>(s_plane) + (st_lds[u] >> 2);     // indexed as vectors: the compiler cannot
                q[0] = (f32x4){f[0], f[1], f[2], f[3]};                                     // see that a float offset is a multiple of
                q[4] = (f32x4){f[4], f[5], f[6], f[7]};                                     // 4 and splits the store into ds_write2_b32
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

// ------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same convolutions on the same staging (round 4): dw[c][kt][kh][kw] = sum over (n, t, h, w) of
// dy[n][t][h][w][c] * x[n][t + kt - 1][h*s + kh - 1][w*s + kw - 1][c].
//
// The W-blocked stencil (sf_dwconv_wgrad_blocked_kernel) runs one grid slice per kt, reads every dy element three times and
// every x element 13.5 times through the vector-memory path with an address select per load, and at the 14-wide planes of
// MViT stage 3 its 4-column blocks do not divide the row (edge selects on every load, one eighth of the lanes idle).
// Here a workgroup sweeps the T frames of one (sample, row tile, 32-channel chunk) like the forward kernel above:
//   * the input plane tile is staged once as fp32 (halo zero-filled), the output-gradient planes t - 1, t, t + 1 it pairs with
//     sit in LDS as fp16 (three rotating slots, 64 B apart modulo the bank span);
//   * thread = (row-segment subset, (kt, kh) role, 8-channel group): 3 x 8 accumulators (the three kw taps), walking its row
//     segments (the tile's rows cut into equal pieces so that the seven subsets are evenly loaded) with a sliding window over
//     the staged row -- per position one new 32-byte x read (two at stride 2), one
//     16-byte dy read, 24 FMAs; lanes that differ only in kt read the same x address, lanes that differ only in kh the same
//     dy address (LDS broadcast), the rest of a 16-lane group falls on distinct bank slots;
//   * no range checks or address selects in the loop: out-of-image taps are zeros in the staged tile, output rows beyond the
//     image are zeros in the staged dy.
// The workgroup folds its subsets in a fixed order and writes ONE partial row [27][32 channels of C]; the existing
// sf_dwconv_wgrad_finalize_kernel sums the rows (and the head copies of a shared weight).
// The staged x tile keeps the forward kernel's 48-float position pitch (conflict-free staging stores) and every staged row
// carries 64 B of padding, which puts the three kh rows of a column on different bank slots.
// Three LDS size classes (static arrays; XF floats of x tile, DP halves per dy plane slot incl. 64 B of bank offset between the
// slots): <7168, 3168> = 47 KiB, three workgroups per CU (a 9 x 16 staged x tile: seven 14-wide output rows at stride 1);
// <12288, 3168> = 67 KiB, two per CU (a whole 14 x 14 input plane with its 7 x 7 stride-2 output); <12288, 6304> = 85 KiB, one
// per CU.  Planes wider than 16 stay on the stencil: with a 32-float pitch (2-way conflicts in the staging stores) and rows cut
// into short segments they fit three workgroups per CU but measured no faster than the stencil (28 x 28 stride 2: 67 against
// 61 us, 56 x 56: 220 = 220 us) and slowed the 14-wide ones (52 -> 56, 30 -> 38 us; profiles/r4/r4_v19_dw_bench.txt).
#define SF_DWW_PP SF_DWT_PP
#define SF_DWW_XF_S 7168
#define SF_DWW_DP_S (98 * SF_DWT_CC + 32)
#define SF_DWW_XF_L SF_DWT_PLANE
#define SF_DWW_DP_L (196 * SF_DWT_CC + 32)
#define SF_DWW_VPT 4                            // 16-byte dy vectors a thread stages per plane
#define SF_DWW_SUBMAX 7                         // 7 subsets x 9 roles x 4 channel groups = 252 threads

struct DwTileWgradParams {
    const f16* x; int ldx;
    const f16* dy; int lddy;
    float* wpart;                       // [N * tiles_h][27][C]
    int N, C, cls, T;
    int Hi, Wi, Ho, Wo, s;
    int TH, RT, CT, rowf;               // output rows per tile; staged x rows = (TH - 1) * s + 3, columns = Wi + 2; floats per staged row
    int tiles_h, nchunks, nsub, nseg, SL, nitems;
    FastDiv fdCT, fdG, fdWo, fdSeg;
};

template <int S, int XF, int DP>
__global__ __launch_bounds__(SF_THREADS, 3) void sf_dwtile_wgrad_kernel(DwTileWgradParams p) {
    // (launch bounds: at most 168 registers, so that the 47 KiB class keeps its third workgroup per CU)
    __shared__ __attribute__((aligned(16))) float s_x[XF];
    __shared__ __attribute__((aligned(16))) f16 s_dy[3 * DP];
    const int tid = threadIdx.x;
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int chunk = (int)(bid % (uint32_t)p.nchunks);
    const int tile = (int)((bid / (uint32_t)p.nchunks) % (uint32_t)p.tiles_h);
    const int n = (int)(bid / (uint32_t)(p.nchunks * p.tiles_h));
    const int c0 = chunk * SF_DWT_CC;
    const int r0 = tile * p.TH;                                 // first output row of the tile
    const int64_t Si = (int64_t)p.T * p.Hi * p.Wi + p.cls, So = (int64_t)p.T * p.Ho * p.Wo + p.cls;

    // ---- staging maps (as in the forward kernel; the x rows carry 64 B of padding so that the kh rows of a column sit on
    // different bank slots)
    const int VX = p.RT * p.CT * SF_DWT_G;
    int x_off[SF_DWT_VPT], x_lds[SF_DWT_VPT];
#pragma unroll
    for (int u = 0; u < SF_DWT_VPT; ++u) {
        const int v = tid + SF_THREADS * u;
        x_off[u] = -1;
        x_lds[u] = -1;
        if (v < VX) {
            uint32_t pos, cgv, i, j;
            fd_divmod((uint32_t)v, p.fdG, pos, cgv);
            fd_divmod(pos, p.fdCT, i, j);
            x_lds[u] = (int)i * p.rowf + (int)j * SF_DWW_PP + (int)cgv * 4;
            const int gr = r0 * p.s - 1 + (int)i, gc = (int)j - 1;
            if ((unsigned)gr < (unsigned)p.Hi && (unsigned)gc < (unsigned)p.Wi) x_off[u] = (gr * p.Wi + gc) * p.ldx + c0 + (int)cgv * 8;
        }
    }
    const int VD = p.TH * p.Wo * SF_DWT_G;
    int d_off[SF_DWW_VPT], d_lds[SF_DWW_VPT];
#pragma unroll
    for (int u = 0; u < SF_DWW_VPT; ++u) {
        const int v = tid + SF_THREADS * u;
        d_off[u] = -1;
        d_lds[u] = -1;
        if (v < VD) {
            uint32_t pos, cgv, r, w;
            fd_divmod((uint32_t)v, p.fdG, pos, cgv);
            fd_divmod(pos, p.fdWo, r, w);
            d_lds[u] = (int)pos * SF_DWT_CC + (int)cgv * 8;
            if (r0 + (int)r < p.Ho) d_off[u] = ((r0 + (int)r) * p.Wo + (int)w) * p.lddy + c0 + (int)cgv * 8;
        }
    }
    const f16* const x_n = p.x + ((int64_t)n * Si + p.cls) * p.ldx;
    const f16* const dy_n = p.dy + ((int64_t)n * So + p.cls) * p.lddy;
    const int64_t plane_x = (int64_t)p.Hi * p.Wi * p.ldx, plane_dy = (int64_t)p.Ho * p.Wo * p.lddy;
    f16x8 prex[SF_DWT_VPT], pred[SF_DWW_VPT];
    auto prefetch_x = [&](int t) {
        const f16* base = x_n + (int64_t)t * plane_x;
#pragma unroll
        for (int u = 0; u < SF_DWT_VPT; ++u) prex[u] = x_off[u] >= 0 ? ld16(base + x_off[u]) : zero8();
    };
    auto prefetch_dy = [&](int t) {
        const f16* base = dy_n + (int64_t)t * plane_dy;
#pragma unroll
        for (int u = 0; u < SF_DWW_VPT; ++u) pred[u] = d_off[u] >= 0 ? ld16(base + d_off[u]) : zero8();
    };
    auto store_dy = [&](int t) {
        f16* slot = s_dy + (t % 3) * DP;
#pragma unroll
        for (int u = 0; u < SF_DWW_VPT; ++u)
            if (d_lds[u] >= 0) st16(slot + d_lds[u], pred[u]);
    };

    // ---- compute map
    const int cg = tid % SF_DWT_G, role = (tid / SF_DWT_G) % 9, sub = tid / (SF_DWT_G * 9);
    const int kt = role / 3, kh = role - 3 * kt;
    const bool worker = sub < p.nsub;
    float acc[3][8];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
    auto X = [&](const float* row, int j, float (&o)[8]) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(row + j * SF_DWW_PP), b = *reinterpret_cast<const f32x4*>(row + j * SF_DWW_PP + 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
    };

    prefetch_dy(0);
    store_dy(0);
    prefetch_x(0);
    if (p.T > 1) prefetch_dy(1);
    for (int tin = 0; tin < p.T; ++tin) {
        __syncthreads();                    // every thread is done with x plane tin - 1 and dy plane tin - 2
#pragma unroll
        for (int u = 0; u < SF_DWT_VPT; ++u) {
            if (x_lds[u] >= 0) {
                float f[8];
                dwt_cvt8(prex[u], f);
                f32x4* const q = reinterpret_cast<f32x4*>(s_x) + (x_lds[u] >> 2);
                q[0] = (f32x4){f[0], f[1], f[2], f[3]};
                q[4] = (f32x4){f[4], f[5], f[6], f[7]};
            }
        }
        if (tin + 1 < p.T) store_dy(tin + 1);
        __syncthreads();
        if (tin + 1 < p.T) prefetch_x(tin + 1);        // in flight under the sweep of plane tin
        if (tin + 2 < p.T) prefetch_dy(tin + 2);
        const int t = tin - kt + 1;                     // the output plane this role pairs with input plane tin
        if (worker && (unsigned)t < (unsigned)p.T) {
            const f16* const dyp = s_dy + (t % 3) * DP + cg * 8;
            for (int it = sub; it < p.nitems; it += p.nsub) {
                uint32_t r, g;
                fd_divmod((uint32_t)it, p.fdSeg, r, g);
                const int w0 = (int)g * p.SL;
                int w1 = w0 + p.SL;
                if (w1 > p.Wo) w1 = p.Wo;
                const float* const xrow = s_x + ((int)r * S + kh) * p.rowf + cg * 4;
                const f16* const dyr = dyp + (int)r * p.Wo * SF_DWT_CC;
                // sliding window over the staged row, unrolled over its rotation (3 positions at stride 1, 2 at stride 2): the
                // window registers change roles instead of being copied (16 v_mov per position in the rolled form, of ~40
                // VALU instructions per position on a kernel that is ~60 % VALU-bound; profiles/r4/r4_v23_pmc2_dw_insts.md)
                float xa[8], xb[8], xc[8];
                auto fma3 = [&](const float (&a)[8], const float (&b)[8], const float (&c)[8], int w) {
                    float d[8];
                    dwt_cvt8(ld16(dyr + w * SF_DWT_CC), d);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        acc[0][e] += d[e] * a[e];
                        acc[1][e] += d[e] * b[e];
                        acc[2][e] += d[e] * c[e];
                    }
                };
                int w = w0;
                if (S == 1) {
                    X(xrow, w0, xa);
                    X(xrow, w0 + 1, xb);
                    for (; w + 2 < w1; w += 3) {
                        X(xrow, w + 2, xc);
                        fma3(xa, xb, xc, w);
                        X(xrow, w + 3, xa);
                        fma3(xb, xc, xa, w + 1);
                        X(xrow, w + 4, xb);
                        fma3(xc, xa, xb, w + 2);
                    }
                    if (w < w1) {
                        X(xrow, w + 2, xc);
                        fma3(xa, xb, xc, w);
                        if (w + 1 < w1) {
                            X(xrow, w + 3, xa);
                            fma3(xb, xc, xa, w + 1);
                        }
                    }
                } else {
                    X(xrow, 2 * w0, xa);
                    for (; w + 1 < w1; w += 2) {
                        X(xrow, 2 * w + 1, xb);
                        X(xrow, 2 * w + 2, xc);
                        fma3(xa, xb, xc, w);
                        X(xrow, 2 * w + 3, xb);
                        X(xrow, 2 * w + 4, xa);
                        fma3(xc, xb, xa, w + 1);
                    }
                    if (w < w1) {
                        X(xrow, 2 * w + 1, xb);
                        X(xrow, 2 * w + 2, xc);
                        fma3(xa, xb, xc, w);
                    }
                }
            }
        }
    }
    // ---- fold the subsets (fixed order) and write the workgroup's partial row
    __syncthreads();
    if (worker) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) s_x[((sub * 27) + role * 3 + k) * SF_DWT_CC + cg * 8 + e] = acc[k][e];
    }
    __syncthreads();
    float* const out = p.wpart + ((int64_t)(n * p.tiles_h + tile) * 27) * p.C + c0;
    for (int idx = tid; idx < 27 * SF_DWT_CC; idx += SF_THREADS) {
        const int tap = idx / SF_DWT_CC, c = idx - tap * SF_DWT_CC;
        float a = 0.f;
        for (int k = 0; k < p.nsub; ++k) a += s_x[(k * 27 + tap) * SF_DWT_CC + c];
        out[(int64_t)tap * p.C + c] = a;
    }
}
