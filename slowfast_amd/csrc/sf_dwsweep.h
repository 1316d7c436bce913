// Depthwise 3x3x3 convolution (padding 1, stride (1, s, s), s = 1 | 2) on channels-last rows: ring-buffered plane sweep with the
// CHANNELS on the lanes (round 6).
//
// Reference call sites: MViT pooling convolutions pool_q / pool_k / pool_v (slowfast/models/attention.py:13-45, 227-266) and the
// X3D bottleneck's channelwise 3x3x3 (slowfast/models/resnet_helper.py:214-224), forward, data gradient and weight gradient.
//
// Why a third kernel family.  The W-blocked stencils (sf_dwconv.h) read every input element 13.5 times through the vector-memory
// path; the LDS plane sweep of round 4 (sf_dwtile.h) fetches it once but keeps positions on the lanes: the 27 x 8 weights of a
// lane then have to come out of LDS again for every plane (6 ds_read_b128 of weights beside 4 of data per 48 FMAs -- LDS-bound,
// 36 us for a 15 us stream) and the staged tile is fp32 (48 KiB per plane).  Here
//   * a lane owns FOUR CHANNELS (one 8-byte LDS word per position) and one row of the tile: its 27 x 4 fp32 weights are loop
//     invariants in registers (the nn.Conv3d parameter itself, no rounding, no staging), read once per workgroup;
//   * the 64 lanes of a wave are 8 rows x 8 channel quads and slide along W together: per output only the NEW column of the
//     3 x 3 x 3 window is read (9 ds_read_b64 per 108 FMAs at stride 1, 18 at stride 2), the products are v_fma_mix (fp16 operand
//     from the LDS word, fp32 weight, fp32 accumulate: no conversions); the row pitch of the staged tile is chosen so that the
//     four rows of a 32-lane read group fall into four different 64-byte bank windows;
//   * planes stay fp16 in LDS (a whole 16 x 16 x 32-channel plane is 17 KiB) and travel global -> LDS directly
//     (global_load_lds_dwordx4, zero line for the halo) into a ring of four slots: planes t-1, t, t+1 are read while t+2 lands; one
//     raw barrier per plane, the wait for a plane's copies is a COUNTED vmcnt that leaves the output stores issued behind them in
//     flight; the temporal halo planes -1 and T are zero copies through the same stream (no special cases in the loop).
// Three bodies on this skeleton:
//   MODE 0  forward (s = 1 | 2), and the stride-1 data gradient as the correlation with the mirrored weights;
//   MODE 1  stride-2 data gradient: a lane owns a 2 x 2 block of dx and the 2 x 2 window of dy it depends on (9 of the 27 taps
//           per position on average: no zero-upsampled products, the four parity classes are straight-line code);
//   MODE 2  weight gradient: the same window as MODE 0 multiplied by the lane's dy word into 27 x 4 accumulators; the 32 lanes
//           that share a channel quad are folded through LDS in a fixed order at the end, one partial row per workgroup
//           (summed by sf_dwconv_wgrad_finalize_kernel).
// BatchNorm partial sums (X3D) ride on MODE 0 as 8 more registers per lane.
#pragma once
#include "sf_common.h"

#define SF_DWS_LDS 81920                        // static LDS of a workgroup: two per CU
#define SF_DWS_NR 4                             // ring slots of the staged operand
#define SF_DWS_NRB 3                            // ring slots of dy (weight gradient)
#define SF_DWS_MAXVPT 5                         // copy instructions per wave and plane (1 KiB each)
#define SF_DWS_MAXVPTB 4

struct DwSweepParams {
    const f16* a; int lda;              // staged operand: x (forward, weight gradient) / dy (data gradient); rows (n, [cls], t, h, w)
    const f16* b; int ldb;              // weight gradient: dy
    f16* dst; int ldd;                  // y / dx
    const float* w;                     // [Cwreal][27] fp32 (the nn.Conv3d parameter)
    float* part;                        // MODE 0: optional [rows][2][C] BatchNorm partial sums; MODE 2: [rows][27][C]
    int N, C, Cw, Cwreal, cls, T;
    int Ha, Wa, Hb, Wb, Hd, Wd;         // plane extents of a, b, dst
    int Hit, Wit;                       // extents of the iterated space (outputs; MODE 1: 2 x 2 blocks of dx = positions of dy)
    int flip;                           // 1: taps mirrored (stride-1 data gradient)
    int TH, TW, tiles_h, tiles_w, nchunks;
    int RA, CA, RP, slotb, vpt;         // staged tile of a: rows, columns, row pitch and ring-slot size in bytes, copies per wave
    int RB, CBt, RPB, slotbB, vptB;     // staged tile of b
    int ngrp, nseg, SL;                 // 8-row groups of the tile, column segments per group, columns per segment
    int nr;                             // sf_dwrot_kernel: ring slots of the staged operand (dy: nr - 1)
    int gs;                             // sf_dwrot_kernel<., 3, ...>: the convolution's stride (>= 3, windows do not overlap)
    // PAIR mode (csplit > 0; round 6): ONE launch for two convolutions of the same geometry on two tensors of csplit channels each
    // (MViT pool_k / pool_v of a block: attention.py:227-266) -- channel chunks >= csplit take the second operand set, with
    // channel offsets counted from csplit; the partial tables are [rows][.][2 * csplit] (Cpart), first tensor first
    const f16* a2; const f16* b2; f16* dst2; const float* w2;
    int csplit, Cpart;
    FastDiv fdRP, fdRPB, fdSeg;
};

typedef f16 dws_x4 __attribute__((ext_vector_type(4)));
typedef uint32_t dws_w2 __attribute__((ext_vector_type(2)));           // four channels of one position: two 32-bit LDS words

__device__ __forceinline__ dws_w2 dws_ld(const char* smem, int off) { return *reinterpret_cast<const dws_w2*>(smem + off); }

// 16-bit element e (0 | 1) of a packed word as fp32 (portable form: the bf16 library and the host simulator)
__device__ __forceinline__ float dws_half(uint32_t x, int e) {
#ifdef SF_ACT_BF16
    return __builtin_bit_cast(float, e ? (x & 0xffff0000u) : (x << 16));
#else
    return (float)__builtin_bit_cast(f16, (uint16_t)(e ? x >> 16 : x & 0xffffu));
#endif
}
// acc + x[E] * w and acc + x[EX] * y[EY] on v_fma_mix_f32 (fp16 operands picked out of the packed words by op_sel, fp32 accumulate:
// no conversion instructions).  Written as asm: left to itself hipcc's SLP vectoriser turns the 108 FMAs of a step into
// v_cvt_f32_f16 + v_pk_fma_f32 pairs -- twice the VALU issue slots (MI355X_MICROARCH.md: packed fp32 is no faster than two FMAs).
#if defined(SF_HOSTSIM) || defined(SF_ACT_BF16)
template <int E>
__device__ __forceinline__ float dws_fma_xw(uint32_t x, float w, float acc) { return __builtin_fmaf(dws_half(x, E), w, acc); }
template <int EX, int EY>
__device__ __forceinline__ float dws_fma_xy(uint32_t x, uint32_t y, float acc) { return __builtin_fmaf(dws_half(x, EX), dws_half(y, EY), acc); }
#else
template <int E>
__device__ __forceinline__ float dws_fma_xw(uint32_t x, float w, float acc) {
    if constexpr (E == 0) asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(x), "v"(w));
    else asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(x), "v"(w));
    return acc;
}
template <int EX, int EY>
__device__ __forceinline__ float dws_fma_xy(uint32_t x, uint32_t y, float acc) {
    if constexpr (EX == 0 && EY == 0) asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(x), "v"(y));
    else if constexpr (EX == 1 && EY == 0) asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(x), "v"(y));
    else if constexpr (EX == 0 && EY == 1) asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(x), "v"(y));
    else asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(x), "v"(y));
    return acc;
}
#endif
// the four channels of a word pair against four fp32 weights / against the four channels of another pair
__device__ __forceinline__ void dws_fma4_w(const dws_w2& x, const float (&w)[4], float (&acc)[4]) {
    acc[0] = dws_fma_xw<0>(x.x, w[0], acc[0]);
    acc[1] = dws_fma_xw<1>(x.x, w[1], acc[1]);
    acc[2] = dws_fma_xw<0>(x.y, w[2], acc[2]);
    acc[3] = dws_fma_xw<1>(x.y, w[3], acc[3]);
}
__device__ __forceinline__ void dws_fma4_y(const dws_w2& x, const dws_w2& y, float (&acc)[4]) {
    acc[0] = dws_fma_xy<0, 0>(x.x, y.x, acc[0]);
    acc[1] = dws_fma_xy<1, 1>(x.x, y.x, acc[1]);
    acc[2] = dws_fma_xy<0, 0>(x.y, y.y, acc[2]);
    acc[3] = dws_fma_xy<1, 1>(x.y, y.y, acc[3]);
}

// wait until this wave's copies are done while up to `keep` (wave-uniform) younger stores stay in flight
__device__ __forceinline__ void dws_wait_copies(int keep) {
    if (keep >= 8) SF_WAIT_VMEM_N(8);
    else if (keep >= 4) SF_WAIT_VMEM_N(4);
    else if (keep >= 2) SF_WAIT_VMEM_N(2);
    else SF_WAIT_VMEM_N(0);
}

template <int MODE, int S, bool STATS>
__global__ __launch_bounds__(SF_THREADS, 2) void sf_dwsweep_kernel(DwSweepParams p) {
    __shared__ __attribute__((aligned(1024))) char smem[SF_DWS_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rs = lane >> 3, q = lane & 7;
    uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int chunk = (int)(bid % (uint32_t)p.nchunks);
    bid /= (uint32_t)p.nchunks;
    const int tw = (int)(bid % (uint32_t)p.tiles_w);
    bid /= (uint32_t)p.tiles_w;
    const int th = (int)(bid % (uint32_t)p.tiles_h);
    const int n = (int)(bid / (uint32_t)p.tiles_h);
    int c0 = chunk * 32;                                        // channel offset inside the operand tensors ...
    const int cpart = c0, cpw = p.csplit > 0 ? p.Cpart : p.C;   // ... and inside the partial tables (row width cpw)
    if (p.csplit > 0 && c0 >= p.csplit) {                       // PAIR mode, second tensor (workgroup-uniform)
        c0 -= p.csplit;
        p.a = p.a2; p.b = p.b2; p.dst = p.dst2; p.w = p.w2;
    }
    const int cq = (p.C - c0) >= 32 ? 8 : (p.C - c0) >> 2;      // channel quads of this chunk
    const bool qok = q < cq;
    const int r0 = th * p.TH, q0 = tw * p.TW;                   // tile origin in the iterated space
    const int ar0 = MODE == 1 ? r0 : r0 * S - 1, ac0 = MODE == 1 ? q0 : q0 * S - 1;     // origin of the staged tile of a
    const f16* const zline = reinterpret_cast<const f16*>(sf_zero_line);

    // ---- weights of the lane's four channels (channel c uses weight row c % Cw; rows >= Cwreal are padding)
    float wr[MODE == 2 ? 1 : 27][4];
    if constexpr (MODE != 2) {
        const int cw = (c0 + 4 * (qok ? q : 0)) % p.Cw;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = qok && cw + e < p.Cwreal;
            const float* wp = p.w + (ok ? cw + e : 0) * 27;
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                const float v = wp[p.flip ? 26 - tap : tap];
                wr[tap][e] = ok ? v : 0.f;
            }
        }
    }

    // ---- copy maps: instruction u of this wave fills bytes [(4u + wave) KiB, +1 KiB) of a ring slot, 16 bytes per lane
    const int64_t Sa = (int64_t)p.T * p.Ha * p.Wa + p.cls;
    const f16* const a_n = p.a + ((int64_t)n * Sa + p.cls) * p.lda + c0;
    const int64_t planeA = (int64_t)p.Ha * p.Wa * p.lda;
    int aoff[SF_DWS_MAXVPT];
#pragma unroll
    for (int u = 0; u < SF_DWS_MAXVPT; ++u) {
        aoff[u] = -1;
        if (u < p.vpt && (u * 4 + wave) * 1024 < p.slotb) {
            const uint32_t o = (uint32_t)(((u * 4 + wave) * 64 + lane) * 16);
            uint32_t i, rem;
            fd_divmod(o, p.fdRP, i, rem);
            const int j = (int)(rem >> 6), sub = (int)(rem & 63);
            const int gr = ar0 + (int)i, gc = ac0 + j;
            if ((int)i < p.RA && j < p.CA && (sub >> 3) < cq && (unsigned)gr < (unsigned)p.Ha && (unsigned)gc < (unsigned)p.Wa)
                aoff[u] = (gr * p.Wa + gc) * p.lda + (sub >> 1);
        }
    }
    auto issue_a = [&](int t) {
        const bool tv = (unsigned)t < (unsigned)p.T;
        const f16* base = a_n + (int64_t)t * planeA;
        char* slot = smem + (t & (SF_DWS_NR - 1)) * p.slotb + wave * 1024;
#pragma unroll
        for (int u = 0; u < SF_DWS_MAXVPT; ++u)
            if (u < p.vpt && (u * 4 + wave) * 1024 < p.slotb) {
                const f16* g = (tv && aoff[u] >= 0) ? base + aoff[u] : zline;
                SF_GLOBAL_LOAD_LDS16_ASM(g, slot + u * 4096);
            }
    };
    // dy planes of the weight gradient: [TH rows][TW columns][64 B], no halo; rows / columns beyond the plane are zeros
    char* const smemB = smem + SF_DWS_NR * p.slotb;
    const int64_t Sb = (int64_t)p.T * p.Hb * p.Wb + p.cls;
    const f16* const b_n = MODE == 2 ? p.b + ((int64_t)n * Sb + p.cls) * p.ldb + c0 : nullptr;
    const int64_t planeB = (int64_t)p.Hb * p.Wb * p.ldb;
    int boff[MODE == 2 ? SF_DWS_MAXVPTB : 1];
    if constexpr (MODE == 2) {
#pragma unroll
        for (int u = 0; u < SF_DWS_MAXVPTB; ++u) {
            boff[u] = -1;
            if (u < p.vptB && (u * 4 + wave) * 1024 < p.slotbB) {
                const uint32_t o = (uint32_t)(((u * 4 + wave) * 64 + lane) * 16);
                uint32_t i, rem;
                fd_divmod(o, p.fdRPB, i, rem);
                const int j = (int)(rem >> 6), sub = (int)(rem & 63);
                const int gr = r0 + (int)i, gc = q0 + j;
                if ((int)i < p.RB && j < p.CBt && (sub >> 3) < cq && gr < p.Hb && gc < p.Wb) boff[u] = (gr * p.Wb + gc) * p.ldb + (sub >> 1);
            }
        }
    }
    auto issue_b = [&](int t) {
        if constexpr (MODE == 2) {
            const f16* base = b_n + (int64_t)t * planeB;
            char* slot = smemB + (t % SF_DWS_NRB) * p.slotbB + wave * 1024;
#pragma unroll
            for (int u = 0; u < SF_DWS_MAXVPTB; ++u)
                if (u < p.vptB && (u * 4 + wave) * 1024 < p.slotbB) {
                    const f16* g = boff[u] >= 0 ? base + boff[u] : zline;
                    SF_GLOBAL_LOAD_LDS16_ASM(g, slot + u * 4096);
                }
        }
    };

    // ---- destination
    const int64_t Sd = (int64_t)p.T * p.Hd * p.Wd + p.cls;
    f16* const dst_n = MODE == 2 ? nullptr : p.dst + ((int64_t)n * Sd + p.cls) * p.ldd + c0 + 4 * q;
    if (MODE != 2 && p.cls && th == 0 && tw == 0 && tid < (cq >> 1))      // the cls row passes through
        st16(p.dst + (int64_t)n * Sd * p.ldd + c0 + tid * 8, ld16(p.a + (int64_t)n * Sa * p.lda + c0 + tid * 8));

    const int ntask = p.ngrp * p.nseg;
    // output stores this wave issues per plane (>= is enough: the counted wait may only under-estimate)
    int nst = 0;
    if constexpr (MODE != 2) {
        for (int k = wave; k < ntask; k += 4) {
            uint32_t g, sg;
            fd_divmod((uint32_t)k, p.fdSeg, g, sg);
            int ce = (int)sg * p.SL + p.SL;
            if (ce > p.TW) ce = p.TW;
            if (q0 + ce > p.Wit) ce = p.Wit - q0;
            const int len = ce - (int)sg * p.SL;
            // (only the stores that are certain: a group whose first row is inside the tile, one store per step)
            if (len > 0 && (int)g * 8 < p.TH && r0 + (int)g * 8 < p.Hit) nst += len;
        }
    }

    float ssum[STATS ? 4 : 1], ssq[STATS ? 4 : 1];
    if constexpr (STATS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
    }
    float wacc[MODE == 2 ? 27 : 1][4];
    if constexpr (MODE == 2) {
#pragma unroll
        for (int tap = 0; tap < 27; ++tap)
#pragma unroll
            for (int e = 0; e < 4; ++e) wacc[tap][e] = 0.f;
    }

    // ---- column of the window: the 3 (planes) x 3 (rows) words of staged column j; base[] = per-plane lane addresses
    auto load_col = [&](dws_w2 (&X)[3][3], const int (&base)[3], int j) {
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) X[kt][kh] = dws_ld(smem, base[kt] + kh * p.RP + j * 64);
    };

    issue_a(-1);
    issue_a(0);
    issue_a(1);
    issue_b(0);
    if (p.T > 1) issue_b(1);

    for (int t = 0; t < p.T; ++t) {
        // planes <= t + 1 were issued before the previous plane's stores: retire them, keep the stores in flight
        if (t == 0) SF_WAIT_VMEM_N(0);
        else dws_wait_copies(nst);
        SF_BARRIER_KEEP_VMEM();             // copies of every wave visible; everyone is done with plane t - 2
        if (t + 2 <= p.T) issue_a(t + 2);
        if (MODE == 2 && t + 2 < p.T) issue_b(t + 2);

        int pbase[3];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) pbase[kt] = ((t - 1 + kt) & (SF_DWS_NR - 1)) * p.slotb;
        for (int k = wave; k < ntask; k += 4) {
            uint32_t g, sg;
            fd_divmod((uint32_t)k, p.fdSeg, g, sg);
            const int r = (int)g * 8 + rs;
            const bool rok = r < p.TH && r0 + r < p.Hit;
            const int cs = (int)sg * p.SL;
            int ce = cs + p.SL;
            if (ce > p.TW) ce = p.TW;
            if (q0 + ce > p.Wit) ce = p.Wit - q0;
            if (ce <= cs) continue;
            const int rr = rok ? r : 0, qq = qok ? q : 0;
            const bool lok = rok && qok;

            if constexpr (MODE == 0 || MODE == 2) {
                int base[3];
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) base[kt] = pbase[kt] + rr * S * p.RP + qq * 8;
                f16* drow = nullptr;
                const char* brow = nullptr;
                if constexpr (MODE == 0) drow = dst_n + ((int64_t)(t * p.Hd + r0 + r) * p.Wd + q0) * p.ldd;
                else brow = smemB + (t % SF_DWS_NRB) * p.slotbB + rr * p.RPB + qq * 8;
                dws_w2 X0[3][3], X1[3][3], X2[3][3];
                // one output column: A, B, C hold staged columns c*S, c*S + 1, c*S + 2
                auto body = [&](const dws_w2 (&A)[3][3], const dws_w2 (&B)[3][3], const dws_w2 (&Cc)[3][3], int c) {
                    if constexpr (MODE == 0) {
                        // column by column (kw outer): the window slot of kw = 0 is dead after its 36 FMAs
                        float a0[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                                for (int kh = 0; kh < 3; ++kh) {
                                    const dws_w2& x = kw == 0 ? A[kt][kh] : kw == 1 ? B[kt][kh] : Cc[kt][kh];
                                    dws_fma4_w(x, wr[(kt * 3 + kh) * 3 + kw], a0);
                                }
                        dws_x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = a0[e];
                            o[e] = (f16)v;
                            if constexpr (STATS) {
                                if (lok) { ssum[e] += v; ssq[e] += v * v; }
                            }
                        }
                        if (lok) *reinterpret_cast<dws_x4*>(drow + (int64_t)c * p.ldd) = o;
                    } else {
                        dws_w2 d = dws_ld(brow, c * 64);
                        if (!lok) d = (dws_w2){0u, 0u};
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                                for (int kh = 0; kh < 3; ++kh) {
                                    const dws_w2& x = kw == 0 ? A[kt][kh] : kw == 1 ? B[kt][kh] : Cc[kt][kh];
                                    dws_fma4_y(x, d, wacc[(kt * 3 + kh) * 3 + kw]);
                                }
                    }
                };
                if constexpr (S == 1) {
                    // window (A, B, C) = columns (c, c+1, c+2); the next output needs (B, C, new)
                    load_col(X0, base, cs);
                    load_col(X1, base, cs + 1);
                    for (int c = cs; c < ce; c += 3) {
                        load_col(X2, base, c + 2);
                        body(X0, X1, X2, c);
                        if (c + 1 < ce) {
                            load_col(X0, base, c + 3);
                            body(X1, X2, X0, c + 1);
                        }
                        if (c + 2 < ce) {
                            load_col(X1, base, c + 4);
                            body(X2, X0, X1, c + 2);
                        }
                    }
                } else {
                    // window (A, B, C) = columns (2c, 2c+1, 2c+2); the next output needs (C, new, new)
                    load_col(X0, base, 2 * cs);
                    for (int c = cs; c < ce; c += 3) {
                        load_col(X1, base, 2 * c + 1);
                        load_col(X2, base, 2 * c + 2);
                        body(X0, X1, X2, c);
                        if (c + 1 < ce) {
                            load_col(X0, base, 2 * c + 3);
                            load_col(X1, base, 2 * c + 4);
                            body(X2, X0, X1, c + 1);
                        }
                        if (c + 2 < ce) {
                            load_col(X2, base, 2 * c + 5);
                            load_col(X0, base, 2 * c + 6);
                            body(X1, X2, X0, c + 2);
                        }
                    }
                }
            } else {
                // ---- MODE 1: lane = block row a = r (dx rows 2a, 2a + 1), columns b = c (dx columns 2b, 2b + 1); the staged tile
                // holds dy rows r0 .. r0 + TH and columns q0 .. q0 + TW (zeros beyond the plane).  dx plane t takes dy plane t + 1 - kt.
                int base[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) base[pl] = pbase[pl] + rr * p.RP + qq * 8;
                const int h0 = 2 * (r0 + r), w00 = 2 * q0;
                f16* drow = dst_n + ((int64_t)(t * p.Hd + h0) * p.Wd + w00) * p.ldd;
                const bool row1 = h0 + 1 < p.Hd;
                dws_w2 D0[3][2], D1[3][2];
                auto load2 = [&](dws_w2 (&D)[3][2], int j) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                        for (int i = 0; i < 2; ++i) D[pl][i] = dws_ld(smem, base[pl] + i * p.RP + j * 64);
                };
                auto body = [&](const dws_w2 (&L)[3][2], const dws_w2 (&R)[3][2], int c) {
                    float o00[4], o01[4], o10[4], o11[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o00[e] = 0.f; o01[e] = 0.f; o10[e] = 0.f; o11[e] = 0.f; }
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const int k9 = (2 - pl) * 9;          // kt = 2 - pl; taps (kh, kw) of that temporal slice
                        const dws_w2 &d00 = L[pl][0], &d01 = R[pl][0], &d10 = L[pl][1], &d11 = R[pl][1];
                        dws_fma4_w(d00, wr[k9 + 4], o00);
                        dws_fma4_w(d01, wr[k9 + 3], o01);
                        dws_fma4_w(d00, wr[k9 + 5], o01);
                        dws_fma4_w(d10, wr[k9 + 1], o10);
                        dws_fma4_w(d00, wr[k9 + 7], o10);
                        dws_fma4_w(d11, wr[k9 + 0], o11);
                        dws_fma4_w(d10, wr[k9 + 2], o11);
                        dws_fma4_w(d01, wr[k9 + 6], o11);
                        dws_fma4_w(d00, wr[k9 + 8], o11);
                    }
                    dws_x4 v00, v01, v10, v11;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v00[e] = (f16)o00[e]; v01[e] = (f16)o01[e]; v10[e] = (f16)o10[e]; v11[e] = (f16)o11[e]; }
                    const bool col1 = w00 + 2 * c + 1 < p.Wd;
                    f16* d = drow + (int64_t)(2 * c) * p.ldd;
                    if (lok) {
                        *reinterpret_cast<dws_x4*>(d) = v00;
                        if (col1) *reinterpret_cast<dws_x4*>(d + p.ldd) = v01;
                        if (row1) {
                            f16* d1 = d + (int64_t)p.Wd * p.ldd;
                            *reinterpret_cast<dws_x4*>(d1) = v10;
                            if (col1) *reinterpret_cast<dws_x4*>(d1 + p.ldd) = v11;
                        }
                        if constexpr (STATS) {          // column sums of dx (the bias gradient of the Linear that produced x)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float a = o00[e];
                                if (col1) a += o01[e];
                                if (row1) { a += o10[e]; if (col1) a += o11[e]; }
                                ssum[e] += a;
                            }
                        }
                    }
                };
                load2(D0, cs);
                for (int c = cs; c < ce; c += 2) {
                    load2(D1, c + 1);
                    body(D0, D1, c);
                    if (c + 1 < ce) {
                        load2(D0, c + 2);
                        body(D1, D0, c + 1);
                    }
                }
            }
        }
    }

    // ---- epilogues: fold the 32 lanes (8 row slots x 4 waves) that share a channel quad, fixed order
    if constexpr (MODE != 2 && STATS) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[tid * 8 + e] = ssum[e];
            red[tid * 8 + 4 + e] = ssq[e];
        }
        __syncthreads();
        if (tid < 64) {
            const int st = tid >> 5, ch = tid & 31;         // (statistic, channel of the chunk)
            float acc = 0.f;
            for (int wv = 0; wv < 4; ++wv)
                for (int r = 0; r < 8; ++r) acc += red[(wv * 64 + r * 8 + (ch >> 2)) * 8 + st * 4 + (ch & 3)];
            if (p.cls && th == 0 && tw == 0 && (ch >> 2) < cq) {
                const float v = (float)p.a[(int64_t)n * Sa * p.lda + c0 + ch];
                acc += st ? v * v : v;
            }
            const int64_t prow = ((int64_t)n * p.tiles_h + th) * p.tiles_w + tw;
            if ((ch >> 2) < cq) p.part[(prow * 2 + st) * cpw + cpart + ch] = acc;
        }
    }
    if constexpr (MODE == 2) {
        float* red = reinterpret_cast<float*>(smem);
        const int64_t prow = ((int64_t)n * p.tiles_h + th) * p.tiles_w + tw;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 9; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[tid * 36 + i * 4 + e] = wacc[kt * 9 + i][e];
            __syncthreads();
            for (int o = tid; o < 9 * 32; o += SF_THREADS) {
                const int i = o >> 5, ch = o & 31;
                float acc = 0.f;
                for (int wv = 0; wv < 4; ++wv)
                    for (int r = 0; r < 8; ++r) acc += red[(wv * 64 + r * 8 + (ch >> 2)) * 36 + i * 4 + (ch & 3)];
                if ((ch >> 2) < cq) p.part[(prow * 27 + kt * 9 + i) * cpw + cpart + ch] = acc;
            }
        }
    }
}

// ========================================================================================================================
// Rotating-accumulator body (round 6, second form): forward / stride-1 data gradient (MODE 0) and weight gradient (MODE 2).
//
// Measured on MI355X (tools/ubench/valu_rate.hip, profiles/r6_v3_valu_rate.txt): EVERY VALU wave-instruction issues in ~4 cycles per
// SIMD -- v_fma_f32, v_fma_mix_f32, v_cvt_f32_f16 alike -- and v_pk_fma_f32 does TWO fp32 FMAs in ~4.6.  The 108 v_fma_mix per
// step of sf_dwsweep_kernel are therefore already the floor of that body (PMC: 7 900 VALU instructions per wave at 4 cycles each,
// two waves per SIMD).  This body halves the instruction count:
//   * a lane owns a channel PAIR (one 32-bit LDS word per position), one row and a run of SL output columns, and keeps the
//     accumulators of THREE output planes (t-1, t, t+1) for the whole run in registers (3 x SL x 2 fp32);
//   * the sweep visits one INPUT plane per iteration: every word of the lane's 3 x (SL + 2) window is read once, converted once
//     (2 v_cvt) and used as the packed operand of 9 v_pk_fma_f32 (3 kw x 3 kt: the two channels of the pair against the weight
//     pair), i.e. 27 packed FMAs + ~8 conversions per output pair instead of 54 v_fma_mix; 3.9 LDS words per output instead of 9;
//   * after input plane t the accumulators of output plane t-1 are complete: stored, zeroed, and the three sets change roles
//     (the plane loop is unrolled by three, no register moves); no temporal halo planes exist at all;
//   * only the plane being read has to be resident, every other ring slot is a plane in flight: a cold plane takes ~4 us from
//     issue to landed, so with the three slots of the first version an iteration lasted latency / 2 = 2 us whatever it computed
//     (profiles/r6_v8_probe.txt: 12 us + 2.0 - 2.3 us per plane at both strides); the ring is as deep as the LDS class allows.
// The weight gradient is the same walk with the roles swapped: the lane keeps the dy words of its run for planes t-1, t, t+1 as
// fp32 pairs (converted once per plane) and 27 packed accumulators.
#define SF_DWR_MAXNR 6
#define SF_DW_GAP_W 3072                    // 27 x Cw floats of LDS in sf_dwgap_dgrad_kernel (Cw <= 112)
typedef float dwr_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ dwr_f2 dwr_cvt2(uint32_t x) { return (dwr_f2){dws_half(x, 0), dws_half(x, 1)}; }
// c + a * b on both halves (v_pk_fma_f32).  An asm statement in the GPU builds: with the builtin, hipcc's scheduler hoists every
// LDS read and conversion of a plane above the FMAs (198 - 256 live registers for a loop that carries 96) and then spills the
// weights; volatile asm keeps the FMAs in source order, which bounds what is live to one row of window words.
__device__ __forceinline__ dwr_f2 dwr_fma(dwr_f2 a, dwr_f2 b, dwr_f2 c) {
#if defined(SF_HOSTSIM)
    return __builtin_elementwise_fma(a, b, c);
#else
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    return c;
#endif
}
// every word of a window row has arrived before the first FMA of the row (one exposed LDS round trip per row, not per word)
template <int N>
__device__ __forceinline__ void dwr_arrived(uint32_t (&w)[N]) {
#if !defined(SF_HOSTSIM)
#pragma unroll
    for (int j = 0; j < N; ++j) asm volatile("" : "+v"(w[j]));
#endif
}

// counted wait: retire everything but the youngest `keep` vector-memory operations (wave-uniform, 0 .. 31)
__device__ __forceinline__ void dwr_wait_keep(int keep) {
    if (keep >= 24) SF_WAIT_VMEM_N(24);
    else if (keep >= 20) SF_WAIT_VMEM_N(20);
    else if (keep >= 16) SF_WAIT_VMEM_N(16);
    else if (keep >= 12) SF_WAIT_VMEM_N(12);
    else if (keep >= 10) SF_WAIT_VMEM_N(10);
    else if (keep >= 8) SF_WAIT_VMEM_N(8);
    else if (keep >= 6) SF_WAIT_VMEM_N(6);
    else if (keep >= 4) SF_WAIT_VMEM_N(4);
    else if (keep >= 3) SF_WAIT_VMEM_N(3);
    else if (keep >= 2) SF_WAIT_VMEM_N(2);
    else if (keep >= 1) SF_WAIT_VMEM_N(1);
    else SF_WAIT_VMEM_N(0);
}

// static LDS: three workgroups per CU for the stride-1 sweeps (VALU-bound: 12 waves per CU), two for the stride-2 ones (a quarter of
// the arithmetic per staged byte: what they need is planes in flight)
template <int MODE, int S>
struct DwrLds { static constexpr int bytes = S == 1 ? 53248 : 81920; };      // (S == 3: strides >= 3, see the kernel)

// Tile = ngrp groups of 4 rows x nseg segments of SL columns (ngrp * nseg <= 4: one (group, segment) task per wave, kept for the
// whole sweep); DwSweepParams: TH = 4 * ngrp, TW = SL * nseg, CA = (TW - 1) * S + 3, RA = (TH - 1) * S + 3.
// Halo positions (outside the image, channels beyond C) are the same for every plane: the ring is zeroed once and the copies of
// those lanes are masked out, so a copy is `global_load_lds_dwordx4 voff, s[plane base]` with one 32-bit register per piece.
template <int MODE, int S, int SL, bool STATS>
__global__ __launch_bounds__(SF_THREADS, MODE == 2 || S == 2 || SL > 4 ? 3 : 4) void sf_dwrot_kernel(DwSweepParams p) {
    constexpr int NJ = (SL - 1) * S + 3;                        // window words per staged row
    constexpr int LDSB = DwrLds<MODE, S>::bytes;
    __shared__ __attribute__((aligned(1024))) char smem[LDSB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rs = lane >> 4, pr = lane & 15;
    uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int chunk = (int)(bid % (uint32_t)p.nchunks);
    bid /= (uint32_t)p.nchunks;
    const int tw = (int)(bid % (uint32_t)p.tiles_w);
    bid /= (uint32_t)p.tiles_w;
    const int th = (int)(bid % (uint32_t)p.tiles_h);
    const int n = (int)(bid / (uint32_t)p.tiles_h);
    int c0 = chunk * 32;                                        // channel offset inside the operand tensors ...
    const int cpart = c0, cpw = p.csplit > 0 ? p.Cpart : p.C;   // ... and inside the partial tables (row width cpw)
    if (p.csplit > 0 && c0 >= p.csplit) {                       // PAIR mode, second tensor (workgroup-uniform)
        c0 -= p.csplit;
        p.a = p.a2; p.b = p.b2; p.dst = p.dst2; p.w = p.w2;
    }
    const int cq = (p.C - c0) >= 32 ? 8 : (p.C - c0) >> 2;      // channel quads of this chunk (pairs: 2 * cq)
    const bool pok = pr < 2 * cq;
    const int r0 = th * p.TH, q0 = tw * p.TW;
    const int ar0 = r0 * S - 1, ac0 = q0 * S - 1;
    // S == 3 stands for every stride >= 3 (p.gs): the 3 x 3 windows of neighbouring outputs are disjoint, so only the rows and
    // columns that are used are staged, packed -- staged row 3 * r + kh = input row (r0 + r) * gs - 1 + kh, columns alike -- and the
    // walk over the staged tile is the one of a stride-3 convolution

    // ---- zero the rings (the halo never changes), then the copies may start
    {
        const int total = p.nr * p.slotb + (MODE == 2 ? (p.nr - 1) * p.slotbB : 0);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid * 16; i < total; i += SF_THREADS * 16) *reinterpret_cast<f32x4*>(smem + i) = z;
    }

    // ---- the wave's task
    const int ntask = p.ngrp * p.nseg;
    const bool has_task = wave < ntask;
    const int g = has_task ? wave / p.nseg : 0, sg = has_task ? wave - g * p.nseg : 0;
    const int r = g * 4 + rs;                                   // tile row of the lane
    const int cs = sg * SL;                                     // first tile column of the run
    const bool rok = has_task && r < p.TH && r0 + r < p.Hit;
    const bool lok = rok && pok;
    const bool wave_rows = has_task && g * 4 < p.TH && r0 + g * 4 < p.Hit;       // the wave has a valid row (wave-uniform)
    int ncol = p.Wit - (q0 + cs);                               // valid columns of the run (wave-uniform)
    if (ncol > SL) ncol = SL;
    if (ncol < 0 || !wave_rows) ncol = 0;

    // ---- weights of the lane's channel pair: w2[tap] = (w[c][tap], w[c + 1][tap])
    dwr_f2 w2[MODE == 2 ? 1 : 27];
    if constexpr (MODE != 2) {
        const int cw = (c0 + 2 * (pok ? pr : 0)) % p.Cw;
        const bool ok0 = pok && cw < p.Cwreal, ok1 = pok && cw + 1 < p.Cwreal;
        const float* wp0 = p.w + (ok0 ? cw : 0) * 27;
        const float* wp1 = p.w + (ok1 ? cw + 1 : 0) * 27;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int tt = p.flip ? 26 - tap : tap;
            const float v0 = wp0[tt], v1 = wp1[tt];
            w2[tap] = (dwr_f2){ok0 ? v0 : 0.f, ok1 ? v1 : 0.f};
        }
    }

    // ---- copy maps: instruction u of this wave fills bytes [(4u + wave) KiB, +1 KiB) of a ring slot, 16 bytes per lane;
    // aoff = byte offset of the lane's piece inside a plane (-1: halo, masked out); amask = instructions with any live lane
    const int64_t Sa = (int64_t)p.T * p.Ha * p.Wa + p.cls;
    const f16* const a_n = p.a + ((int64_t)n * Sa + p.cls) * p.lda + c0;
    const int64_t planeA = (int64_t)p.Ha * p.Wa * p.lda;
    int aoff[SF_DWS_MAXVPT];
    int amask = 0, cnt_a = 0;
#pragma unroll
    for (int u = 0; u < SF_DWS_MAXVPT; ++u) {
        aoff[u] = -1;
        if (u < p.vpt && (u * 4 + wave) * 1024 < p.slotb) {
            const uint32_t o = (uint32_t)(((u * 4 + wave) * 64 + lane) * 16);
            uint32_t i, rem;
            fd_divmod(o, p.fdRP, i, rem);
            const int j = (int)(rem >> 6), sub = (int)(rem & 63);
            int gr = ar0 + (int)i, gc = ac0 + j;
            if constexpr (S == 3) {
                gr = (r0 + (int)i / 3) * p.gs - 1 + (int)i % 3;
                gc = (q0 + j / 3) * p.gs - 1 + j % 3;
            }
            if ((int)i < p.RA && j < p.CA && (sub >> 3) < cq && (unsigned)gr < (unsigned)p.Ha && (unsigned)gc < (unsigned)p.Wa)
                aoff[u] = ((gr * p.Wa + gc) * p.lda + (sub >> 1)) * 2;
            if (__any(aoff[u] >= 0)) { amask |= 1 << u; ++cnt_a; }
        }
    }
    amask = __builtin_amdgcn_readfirstlane(amask);
    cnt_a = __builtin_amdgcn_readfirstlane(cnt_a);
    auto issue_a = [&](int t, int slot_a) {
        const f16* base = a_n + (int64_t)t * planeA;
        char* slot = smem + slot_a * p.slotb + wave * 1024;
#pragma unroll
        for (int u = 0; u < SF_DWS_MAXVPT; ++u)
            if (amask & (1 << u)) {
                SF_GLOBAL_LOAD_LDS16_SADDR_IF(aoff[u] >= 0, base, aoff[u], slot + u * 4096);
            }
    };
    char* const smemB = smem + p.nr * p.slotb;
    const int64_t Sb = (int64_t)p.T * p.Hb * p.Wb + p.cls;
    const f16* const b_n = MODE == 2 ? p.b + ((int64_t)n * Sb + p.cls) * p.ldb + c0 : nullptr;
    const int64_t planeB = (int64_t)p.Hb * p.Wb * p.ldb;
    int boff[MODE == 2 ? SF_DWS_MAXVPTB : 1];
    int bmask = 0, cnt_b = 0;
    if constexpr (MODE == 2) {
#pragma unroll
        for (int u = 0; u < SF_DWS_MAXVPTB; ++u) {
            boff[u] = -1;
            if (u < p.vptB && (u * 4 + wave) * 1024 < p.slotbB) {
                const uint32_t o = (uint32_t)(((u * 4 + wave) * 64 + lane) * 16);
                uint32_t i, rem;
                fd_divmod(o, p.fdRPB, i, rem);
                const int j = (int)(rem >> 6), sub = (int)(rem & 63);
                const int gr = r0 + (int)i, gc = q0 + j;
                if ((int)i < p.RB && j < p.CBt && (sub >> 3) < cq && gr < p.Hb && gc < p.Wb)
                    boff[u] = ((gr * p.Wb + gc) * p.ldb + (sub >> 1)) * 2;
                if (__any(boff[u] >= 0)) { bmask |= 1 << u; ++cnt_b; }
            }
        }
        bmask = __builtin_amdgcn_readfirstlane(bmask);
        cnt_b = __builtin_amdgcn_readfirstlane(cnt_b);
    }
    auto issue_b = [&](int t, int slot_b) {
        if constexpr (MODE == 2) {
            const f16* base = b_n + (int64_t)t * planeB;
            char* slot = smemB + slot_b * p.slotbB + wave * 1024;
#pragma unroll
            for (int u = 0; u < SF_DWS_MAXVPTB; ++u)
                if (bmask & (1 << u)) {
                    SF_GLOBAL_LOAD_LDS16_SADDR_IF(boff[u] >= 0, base, boff[u], slot + u * 4096);
                }
        }
    };

    // ---- destination: wave-uniform plane base + the lane's 32-bit byte offset
    const int64_t Sd = (int64_t)p.T * p.Hd * p.Wd + p.cls;
    char* const dst_n = MODE == 2 ? nullptr : reinterpret_cast<char*>(p.dst + ((int64_t)n * Sd + p.cls) * p.ldd + c0);
    const uint32_t dvo = MODE == 2 ? 0u : (uint32_t)((((r0 + r) * p.Wd + q0 + cs) * p.ldd + 2 * pr) * 2);
    const int64_t plane_d = (int64_t)p.Hd * p.Wd * p.ldd * 2;
    if (MODE != 2 && p.cls && th == 0 && tw == 0 && tid < (cq >> 1))      // the cls row passes through
        st16(p.dst + (int64_t)n * Sd * p.ldd + c0 + tid * 8, ld16(p.a + (int64_t)n * Sa * p.lda + c0 + tid * 8));

    dwr_f2 ssum = {0.f, 0.f}, ssq = {0.f, 0.f};
    const dwr_f2 zero2 = {0.f, 0.f};
    // MODE 0: A*[o] = accumulators of an output plane; MODE 2: A*[o] = dy of a plane as fp32 pairs, wacc = 27 packed accumulators
    dwr_f2 A0[SL], A1[SL], A2[SL];
#pragma unroll
    for (int o = 0; o < SL; ++o) { A0[o] = zero2; A1[o] = zero2; A2[o] = zero2; }
    dwr_f2 wacc[MODE == 2 ? 27 : 1];
    if constexpr (MODE == 2) {
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) wacc[tap] = zero2;
    }
    const int lane_off = (rok ? r : 0) * S * p.RP + cs * S * 64 + (pok ? pr : 0) * 4;
    const int lane_offB = (rok ? r : 0) * p.RPB + cs * 64 + (pok ? pr : 0) * 4;

    auto emit = [&](dwr_f2 (&A)[SL], int t) {           // MODE 0: store output plane t, clear the set
        char* d = dst_n + (int64_t)t * plane_d;
#pragma unroll
        for (int o = 0; o < SL; ++o) {
            if (o < ncol) {
                const dwr_f2 v = A[o];
                if constexpr (STATS) {
                    if (lok) { ssum += v; ssq = __builtin_elementwise_fma(v, v, ssq); }
                }
                f16x2 h = {(f16)v.x, (f16)v.y};
                if (lok) *reinterpret_cast<f16x2*>(d + (uint64_t)(dvo + (uint32_t)(o * p.ldd * 2))) = h;
            }
            A[o] = zero2;
        }
    };
    auto load_dy = [&](dwr_f2 (&A)[SL], int t, int slot_b) {        // MODE 2: dy plane t of the lane's run as fp32 pairs (zeros outside)
        if ((unsigned)t < (unsigned)p.T) {
            const char* src = smemB + slot_b * p.slotbB + lane_offB;
            uint32_t wd[SL];
#pragma unroll
            for (int o = 0; o < SL; ++o) wd[o] = *reinterpret_cast<const uint32_t*>(src + o * 64);
#pragma unroll
            for (int o = 0; o < SL; ++o) A[o] = lok ? dwr_cvt2(wd[o]) : zero2;
        } else {
#pragma unroll
            for (int o = 0; o < SL; ++o) A[o] = zero2;
        }
    };
    // one input plane: P0 / P1 / P2 = the sets of planes t-1 / t / t+1 (taps kt = 2 / 1 / 0)
    auto plane = [&](dwr_f2 (&P0)[SL], dwr_f2 (&P1)[SL], dwr_f2 (&P2)[SL], int slot_a) {
        const char* src = smem + slot_a * p.slotb + lane_off;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const char* row = src + kh * p.RP;
            uint32_t wd[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) wd[j] = *reinterpret_cast<const uint32_t*>(row + j * 64);
            dwr_arrived(wd);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const dwr_f2 x = dwr_cvt2(wd[j]);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    if ((j - kw) % S != 0) continue;
                    const int o = (j - kw) / S;
                    if (j - kw < 0 || o >= SL) continue;
                    if constexpr (MODE == 0) {
                        P0[o] = dwr_fma(x, w2[(2 * 3 + kh) * 3 + kw], P0[o]);
                        P1[o] = dwr_fma(x, w2[(1 * 3 + kh) * 3 + kw], P1[o]);
                        P2[o] = dwr_fma(x, w2[(0 * 3 + kh) * 3 + kw], P2[o]);
                    } else {
                        wacc[(2 * 3 + kh) * 3 + kw] = dwr_fma(x, P0[o], wacc[(2 * 3 + kh) * 3 + kw]);
                        wacc[(1 * 3 + kh) * 3 + kw] = dwr_fma(x, P1[o], wacc[(1 * 3 + kh) * 3 + kw]);
                        wacc[(0 * 3 + kh) * 3 + kw] = dwr_fma(x, P2[o], wacc[(0 * 3 + kh) * 3 + kw]);
                    }
                }
            }
        }
    };

    // ---- sweep.  D = nr - 1 planes in flight.  Order of this wave's vector-memory operations (C = copies of a plane, S = stores
    // of an output plane; the weight gradient issues the dy plane B(k) in front of C(k) and stores nothing):
    //   C(0) .. C(D-1) | it 0: C(D) | it 1: C(D+1) S(0) | it k: C(k+D) S(k-1) ...
    // Before plane k is read everything up to C(k) -- and B(k+1) -- must have retired; what may stay in flight is counted.
    __syncthreads();                                    // the rings are zero
    const int D = p.nr - 1;
    for (int k = 0; k < D && k < p.T; ++k) {
        if constexpr (MODE == 2) issue_b(k, k % D);
        issue_a(k, k);
    }
    const int nst = MODE == 0 ? ncol : 0;
    int sa = 0, sa_next = D % p.nr;                     // slots of plane t and of plane t + D
    int sb1 = D > 1 ? 1 : 0, sb_next = 0;               // dy slots of plane t + 1 and of plane t + D
    auto step = [&](dwr_f2 (&P0)[SL], dwr_f2 (&P1)[SL], dwr_f2 (&P2)[SL], int t) {
        int last = t + D - 1;                           // youngest plane issued so far
        if (last > p.T - 1) last = p.T - 1;
        int keep;
        if constexpr (MODE == 0) {
            int lo = t - D - 1;                         // stores behind C(t): S(max(0, t-D-1)) .. S(t-2)
            if (lo < 0) lo = 0;
            const int ns = t - 2 - lo + 1;
            keep = (last - t) * cnt_a + (ns > 0 ? ns * nst : 0);
        } else {
            // B(t+1) must be there too: C(t+1) sits right behind it, then whole planes
            keep = t + 1 <= last ? cnt_a + (last - (t + 1)) * (cnt_a + cnt_b) : 0;
        }
        dwr_wait_keep(keep);
        SF_BARRIER_KEEP_VMEM();                         // plane t of every wave visible; slot of plane t-1 is free
        if (t + D < p.T) {
            if constexpr (MODE == 2) issue_b(t + D, sb_next);
            issue_a(t + D, sa_next);
        }
        if (has_task) {
            if constexpr (MODE == 2) load_dy(P2, t + 1, sb1);
            plane(P0, P1, P2, sa);
            if constexpr (MODE == 0) {
                if (t >= 1) emit(P0, t - 1);
                else {                                  // what plane 0 sent to "output plane -1": dropped
#pragma unroll
                    for (int o = 0; o < SL; ++o) P0[o] = zero2;
                }
            }
        }
        sa = sa + 1 == p.nr ? 0 : sa + 1;
        sa_next = sa_next + 1 == p.nr ? 0 : sa_next + 1;
        sb1 = sb1 + 1 >= D ? 0 : sb1 + 1;
        sb_next = sb_next + 1 >= D ? 0 : sb_next + 1;
    };
    if constexpr (MODE == 2) {
        // dy(0) must be in LDS before the first plane reads it
        SF_WAIT_VMEM_N(0);
        SF_BARRIER_KEEP_VMEM();
        if (has_task) load_dy(A1, 0, 0);
    }
    int t = 0;
#pragma unroll 1
    for (; t + 2 < p.T; t += 3) {
        step(A0, A1, A2, t);
        step(A1, A2, A0, t + 1);
        step(A2, A0, A1, t + 2);
    }
    if (t < p.T) {
        step(A0, A1, A2, t);
        if (t + 1 < p.T) {
            step(A1, A2, A0, t + 1);
            if constexpr (MODE == 0) { if (has_task) emit(A2, t + 1); }
        } else if constexpr (MODE == 0) { if (has_task) emit(A1, t); }
    } else if constexpr (MODE == 0) { if (has_task) emit(A0, p.T - 1); }

    // ---- epilogues: fold the lanes that share a channel pair (4 row slots x 4 waves), fixed order
    if constexpr (MODE == 0 && STATS) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        red[tid * 4 + 0] = ssum.x; red[tid * 4 + 1] = ssum.y; red[tid * 4 + 2] = ssq.x; red[tid * 4 + 3] = ssq.y;
        __syncthreads();
        if (tid < 64) {
            const int st = tid >> 5, ch = tid & 31;
            float acc = 0.f;
            for (int wv = 0; wv < 4; ++wv)
                for (int rr = 0; rr < 4; ++rr) acc += red[(wv * 64 + rr * 16 + (ch >> 1)) * 4 + st * 2 + (ch & 1)];
            if (p.cls && th == 0 && tw == 0 && (ch >> 2) < cq) {
                const float v = (float)p.a[(int64_t)n * Sa * p.lda + c0 + ch];
                acc += st ? v * v : v;
            }
            const int64_t prow = ((int64_t)n * p.tiles_h + th) * p.tiles_w + tw;
            if ((ch >> 2) < cq) p.part[(prow * 2 + st) * cpw + cpart + ch] = acc;
        }
    }
    if constexpr (MODE == 2) {
        float* red = reinterpret_cast<float*>(smem);
        const int64_t prow = ((int64_t)n * p.tiles_h + th) * p.tiles_w + tw;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                red[tid * 18 + i * 2 + 0] = wacc[kt * 9 + i].x;
                red[tid * 18 + i * 2 + 1] = wacc[kt * 9 + i].y;
            }
            __syncthreads();
            for (int o = tid; o < 9 * 32; o += SF_THREADS) {
                const int i = o >> 5, ch = o & 31;
                float acc = 0.f;
                for (int wv = 0; wv < 4; ++wv)
                    for (int rr = 0; rr < 4; ++rr) acc += red[(wv * 64 + rr * 16 + (ch >> 1)) * 18 + i * 2 + (ch & 1)];
                if ((ch >> 2) < cq) p.part[(prow * 27 + kt * 9 + i) * cpw + cpart + ch] = acc;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// Data gradient of the same convolution at strides >= 3 (MViT k / v pooling of stages 1 and 2: stride 8 and 4): the windows of
// neighbouring outputs are disjoint, so an input position receives at most ONE (kh, kw) tap -- (h + 1) = qh * s + kh with
// kh <= 2 -- and three temporal ones; every other position of dx is zero.  One thread = one position x 8 channels: the quotients
// come from two magic divisions, no tap loop (sf_dwconv_dgrad_kernel walks all 27 taps with a division each: 173 us for a
// 154 MB write at the block-0 shape), weights as fp32 from LDS.
struct DwGapParams {
    const f16* dy; int lddy;
    f16* dx; int lddx;
    const float* w;                     // [Cwreal][27]
    int N, C, Cw, Cwreal, cls, T, Hi, Wi, Ho, Wo, s;
    int64_t rows;                       // N * (T * Hi * Wi + cls)
    FastDiv fdRow, fdW, fdH, fdS, fdG;
};
__global__ __launch_bounds__(SF_THREADS) void sf_dwgap_dgrad_kernel(DwGapParams p) {
    __shared__ float s_w[SF_DW_GAP_W];
    for (int i = threadIdx.x; i < 27 * p.Cw; i += SF_THREADS) {
        const int cw = i / 27, tap = i - cw * 27;
        s_w[tap * p.Cw + cw] = cw < p.Cwreal ? p.w[i] : 0.f;
    }
    __syncthreads();
    const int G = p.C >> 3;
    const int64_t So = (int64_t)p.T * p.Ho * p.Wo + p.cls;
    const int64_t total = p.rows * G;
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * SF_THREADS) {
        uint32_t row, g, n, rr, q, w, t, h;
        fd_divmod((uint32_t)idx, p.fdG, row, g);
        fd_divmod(row, p.fdRow, n, rr);
        const int c = (int)g * 8;
        f16* out = p.dx + (int64_t)row * p.lddx + c;
        if (p.cls && rr == 0) {                                 // the cls row passes through
            st16(out, ld16(p.dy + (int64_t)n * So * p.lddy + c));
            continue;
        }
        fd_divmod(rr - (uint32_t)p.cls, p.fdW, q, w);
        fd_divmod(q, p.fdH, t, h);
        uint32_t qh, kh, qw, kw;
        fd_divmod(h + 1u, p.fdS, qh, kh);
        fd_divmod(w + 1u, p.fdS, qw, kw);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (kh <= 2u && kw <= 2u && qh < (uint32_t)p.Ho && qw < (uint32_t)p.Wo) {
            const f16* src = p.dy + ((int64_t)n * So + p.cls + (int64_t)qh * p.Wo + qw) * p.lddy + c;
            const int cw = c % p.Cw;
            // dx plane t takes dy plane t + 1 - kt
            f16x8 v[3];
            bool ok[3];
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                const int to = (int)t + 1 - kt;
                ok[kt] = (unsigned)to < (unsigned)p.T;
                v[kt] = ld16(src + (int64_t)(ok[kt] ? to : 0) * p.Ho * p.Wo * p.lddy);
            }
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                const float* wt = s_w + ((kt * 3 + (int)kh) * 3 + (int)kw) * p.Cw + cw;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += ok[kt] ? (float)v[kt][e] * wt[e] : 0.f;
            }
        }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)acc[e];
        st16(out, o);
    }
}
