// Implicit-GEMM convolution kernels (forward / data-gradient / weight-gradient) for gfx950.
//
// One gather-GEMM skeleton serves every Conv3d geometry of the reference's ResNet/SlowFast path
// (reference call sites: slowfast/models/resnet_helper.py:331-369 BottleneckTransform a/b/c,
// :485-493 ResBlock.branch1, stem_helper.py:182-189 ResNetBasicStem.conv,
// video_model_builder.py:147-154 FuseFastToSlow.conv_f2s):
//   fwd  : Y[m, co]   = sum_k A(m, k) * Wf[co, k]     A = gathered input,   k = (tap, ci)
//   dgrad: dX[p, ci]  = sum_k A(p, k) * Wd[ci, k]     A = gathered dY,      k = (tap, co)
//   wgrad: dW[co, k]  = sum_m dY[m, co] * A(m, k)     (reduction over positions, LDS transpose read)
// Tiles: 128 x BN x 32, 4 waves, v_mfma_f32_16x16x32_f16, fp32 accumulate, register-staged double
// buffered LDS (global loads of step s+1 are in flight under the MFMAs of step s).
#pragma once
#include "sf_common.h"

struct IgemmParams {
    GatherSide g;
    int M;              // rows of the output
    const f16* wmat;    // [Nout][ldw] fp16, K contiguous, zero padded to ldw
    int ldw;
    int Nout;
    int ksteps;         // ceil(Ktot / 32)
    f16* y;
    int ldy;
    const float* bias;  // optional [Nout]
    const f16* resid;   // optional [M][ldr] added in the epilogue (dgrad accumulation)
    int ldr;
    float* stat_part;   // optional [mtiles][2][Nout] per-tile column sum / sum of squares (BatchNorm)
    int ntiles_n;
    // batched GEMMs (attention): blockIdx.y = b*bh + j, operands advance by (b, j) strides (elements)
    int bh;
    int64_t sa_b, sa_h, sw_b, sw_h, sy_b, sy_h, sr_b, sr_h;
    // fused activation epilogues of the Mlp (common.py:25-34): act_mode 1 writes act_aux = gelu(y) next to y (fc1),
    // act_mode 2 multiplies y by gelu'(act_aux) (data gradient of fc2 -> gradient of fc1's output);
    // act_mode 3 = ReLU on the stored value (eval-mode convolutions with BatchNorm folded into weights + bias)
    int act_mode;
    f16* act_aux;
    int ld_aux;
    int resid_row0;     // the residual is added to rows >= resid_row0 only (pooled attention: not to the cls row)
    const uint8_t* resid_bits;  // optional [M][Nout/8] bit mask: residual element (m, c) counts only when its bit is set
    float alpha;        // accumulators are scaled by alpha before bias / residual (0 means 1)
    // Fused BatchNorm-backward reduction (data gradients of a convolution whose INPUT was relu(bn(bnb_y)), engine.ResBlockFn):
    // the tile this workgroup stores IS the gradient dz w.r.t. that activation, so the per-channel sums sf_bn_bwd_reduce would
    // make in a separate pass over dz and y -- sum g and sum g * y with g = dz masked by (y * scale + shift > 0) -- are taken
    // here from the stored (fp16-rounded) values: one read of the y tile instead of a pass over dz AND y, one launch less.
    // bnb_part[mt][2][Nout] gets one row per M tile (fixed summation order: deterministic); nullptr = off.
    const f16* bnb_y; int bnb_ld;
    const float* bnb_scale; const float* bnb_shift;
    float* bnb_part;
    const uint8_t* bnb_bits;        // optional [rows][Nout/8] bit mask replacing the recomputed one (block-output ReLU)
    F32Rows f32;                    // fp32 side rows of the output (token residual sums; sf_common.h), f32.out == nullptr: off
    int linear;                     // host-side hint: a plain matrix product (nn.Linear / attention GEMM), not a convolution
};

// g = dz masked by the producer's ReLU (same expression as masked_grad8 / sf_bn_bwd_apply use), accumulated per channel.
// use_bits: the mask is bit e of `bits` (the 1-bit image of a block output, sf_bn_act) instead of the recomputed one.
__device__ __forceinline__ void bnb_accumulate(const f16x8& dz, const f16x8& yv, const float (&sc)[8], const float (&sh)[8],
                                               float (&sg)[8], float (&sgy)[8], bool use_bits, uint32_t bits) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bool open = use_bits ? ((bits >> e) & 1u) != 0u : ((float)yv[e] * sc[e] + sh[e] > 0.f);
        const float g = open ? (float)dz[e] : 0.f;
        sg[e] += g;
        sgy[e] += g * (float)yv[e];
    }
}

// Everything the store loop of the implicit-GEMM epilogues reads from global memory for ONE 8-column group of one row.  The
// loop is run in chunks: all loads of a chunk are issued first (no store in between, so they are in flight together), then the
// chunk is combined and stored -- the row-at-a-time form paid one full memory latency per row and operand
// (round 3: the fused BatchNorm-backward reduction added a third operand and cost as much as the pass it replaced).
struct EpiLoads {
    f16x8 r, y;                 // residual, BatchNorm-backward operand (the rarer operands -- GELU input, second BatchNorm -- are
    uint32_t rbits, bbits;      // read in the combine phase: 10 registers per row in flight, the budget the accumulators leave)
};

// Workgroup reduction of the per-thread 8-channel sums of the store loop (thread t keeps column group t % CG): butterfly over
// the lanes of a wave that share a group, waves through LDS in a fixed order, one partial-table row [2][Nout] per M tile.
template <int NW, int CG>
__device__ __forceinline__ void bnb_reduce_store(float (&sg)[8], float (&sgy)[8], float* red, float* prow, int n0, int Nout,
                                                 float* sgy2 = nullptr, float* prow2 = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int e = 0; e < 8; ++e)
        for (int mask = CG; mask < 64; mask <<= 1) {
            sg[e] += __shfl_xor(sg[e], mask);
            sgy[e] += __shfl_xor(sgy[e], mask);
            if (prow2) sgy2[e] += __shfl_xor(sgy2[e], mask);
        }
    __syncthreads();                                   // every thread is done with the staging buffer `red` overlays
    if (lane < CG) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[(wave * CG + lane) * 24 + e] = sg[e];
            red[(wave * CG + lane) * 24 + 8 + e] = sgy[e];
            if (prow2) red[(wave * CG + lane) * 24 + 16 + e] = sgy2[e];
        }
    }
    __syncthreads();
    if (tid < CG * 8) {
        const int cgi = tid >> 3, e = tid & 7;
        const int col = n0 + cgi * 8 + e;
        if (col < Nout) {
            float s = 0.f, q = 0.f, q2 = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                s += red[(w * CG + cgi) * 24 + e];
                q += red[(w * CG + cgi) * 24 + 8 + e];
                if (prow2) q2 += red[(w * CG + cgi) * 24 + 16 + e];
            }
            prow[col] = s;
            prow[Nout + col] = q;
            if (prow2) { prow2[col] = s; prow2[Nout + col] = q2; }
        }
    }
}

// GL = plain-GEMM operands (1x1x1 unit-stride convolutions without a fused input BatchNorm, linear layers, attention
// products; K a multiple of 32) are copied global -> LDS directly (global_load_lds_dwordx4): no staging registers, no
// ds_write pass, no per-element address arithmetic in the K loop.  The LDS image is lane-linear (16 rows x 64 B per
// wave instruction), so the XOR swizzle of lds_tile_off() is applied on the SOURCE side: lane l of a 16-row chunk
// fetches the logical 16-byte slot that belongs at physical slot l & 3 of row l >> 2.  Rows beyond M / Nout are clamped
// to the last valid row (their results are never stored).
// (A second register stage, "PF2", was measured and removed: +90 VGPRs, one resident workgroup, SlowFast 383 vs 507
// clips/s -- profiles/r1_visit7_*_pf2.json.)
// Two loader experiments were measured and removed: an incremental tap iterator for the register-staged loader (3 % slower than
// the dividing gather, profiles/r1/r1_visit19_ab.txt) and a three-stage ring for the direct-to-LDS path (inline-asm copies, raw
// barrier, counted waits: no gain on any pointwise layer, -1 % on MViTv2-S at three workgroups per CU,
// profiles/r3/r3_v10_gl3_ab.txt -- these layers are not latency-bound).
// F32R: the fp32 side rows of the output (IgemmParams::f32) are compiled in -- a separate instantiation, because even the dead
// branch costs the 128-VGPR variants 20 spilled registers (hipcc -Rpass-analysis=kernel-resource-usage, round 4).
template <int BN, int WM, int WN, bool PW, bool GL = false, bool OCC4 = false, bool F32R = false>
__global__ __launch_bounds__(SF_THREADS, OCC4 ? 4 : 1) void sf_igemm_kernel(IgemmParams p) {
    constexpr int BM = 128, BK = 32;
    constexpr int NST = 2;
    constexpr int WAVES_N = BN / WN, WAVES_M = BM / WM;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK;
    constexpr int STG_LD = BN + 8;
    constexpr int SMEM_MAIN = NST * (A_ELEMS + B_ELEMS), SMEM_STG = BM * STG_LD;
    constexpr int SMEM = SMEM_MAIN > SMEM_STG ? SMEM_MAIN : SMEM_STG;
    constexpr int NB = (BN * 4 + SF_THREADS - 1) / SF_THREADS;

    // ONE LDS object (hipcc serialises direct-to-LDS copies against ds_reads of any OTHER __shared__ object):
    // [operand stages | epilogue staging] [BatchNorm scale/shift tables (register-staged variant only)] [stat partials]
    constexpr int TF_BYTES = GL ? 0 : 2 * 512 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[SMEM * 2 + TF_BYTES + WAVES_M * 2 * BN * 4];
    f16* const smem = reinterpret_cast<f16*>(lds_raw);
    float* const s_scale = reinterpret_cast<float*>(lds_raw + SMEM * 2);
    float* const s_shift = s_scale + 512;
    float (*const s_red)[2][BN] = reinterpret_cast<float (*)[2][BN]>(lds_raw + SMEM * 2 + TF_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tile = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int nt = tile % p.ntiles_n, mt = tile / p.ntiles_n;
    const int m0 = mt * BM, n0 = nt * BN;
    const GatherSide& g = p.g;
    const bool has_tf = g.scale != nullptr;
    const f16* a_src = g.src;
    const f16* wmat = p.wmat;
    f16* yout = p.y;
    const f16* resid = p.resid;
    if (p.bh > 0) {
        const int zb = blockIdx.y / p.bh, zj = blockIdx.y % p.bh;
        a_src += zb * p.sa_b + zj * p.sa_h;
        wmat += zb * p.sw_b + zj * p.sw_h;
        yout += zb * p.sy_b + zj * p.sy_h;
        if (resid) resid += zb * p.sr_b + zj * p.sr_h;
    }

    if constexpr (!GL) {
        if (has_tf) {
            for (int c = tid; c < g.C; c += SF_THREADS) {
                s_scale[c] = g.scale[c];
                s_shift[c] = g.shift[c];
            }
        }
    }

    // loader assignment: A rows (tid>>2) and (tid>>2)+64, 16-byte slot tid&3
    const int kq = tid & 3;
    RowPos rp[2];
    if constexpr (!GL) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int row = m0 + (tid >> 2) + 64 * j;
            rp[j] = PW ? decode_row_pw(g, (uint32_t)row, row < p.M) : decode_row(g, (uint32_t)row, row < p.M);
        }
    }
    const float act_lo = g.relu ? 0.f : -INFINITY;
    // GL: wave w copies the 16-row chunks w and w + 4 of the A tile and chunks w, w + 4, ... of the B tile
    constexpr int NBC = (BN / 16 + 3) / 4;
    const f16* ga[2];
    const f16* gb[NBC];
    if constexpr (GL) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (wave + 4 * j) * 16 + (lane >> 2);
            const int slot = ((lane & 3) - 2 * ((row >> 2) & 3)) & 3;
            int m = m0 + row;
            if (m >= p.M) m = p.M - 1;
            ga[j] = a_src + (int64_t)m * g.ld + slot * 8;
        }
#pragma unroll
        for (int j = 0; j < NBC; ++j) {
            const int row = (wave + 4 * j) * 16 + (lane >> 2);
            const int slot = ((lane & 3) - 2 * ((row >> 2) & 3)) & 3;
            int co = n0 + row;
            if (co >= p.Nout) co = p.Nout - 1;
            gb[j] = wmat + (int64_t)co * p.ldw + slot * 8;
        }
    }
    auto issue_tile = [&](int ks, int buf) {
        f16* As = smem + buf * (A_ELEMS + B_ELEMS);
        f16* Bs = As + A_ELEMS;
#pragma unroll
        for (int j = 0; j < 2; ++j) SF_GLOBAL_LOAD_LDS16(ga[j] + ks * BK, As + (wave + 4 * j) * 16 * BK);
#pragma unroll
        for (int j = 0; j < NBC; ++j)
            if ((wave + 4 * j) * 16 < BN) SF_GLOBAL_LOAD_LDS16(gb[j] + ks * BK, Bs + (wave + 4 * j) * 16 * BK);
    };

    struct Stage {
        f16x8 ra[2], rb[NB];
        bool ok[2];
        uint32_t c0;
    };
    Stage st0;

    auto load_tile = [&](int ks, Stage& st) {
        const uint32_t k0 = (uint32_t)(ks * BK + kq * 8);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int64_t off;
            uint32_t c0 = 0;
            // masked lanes issue no request (measured: an unconditional clamped load + select is 8-13 % SLOWER here,
            // profiles/r1/r1_visit9_*; the four loads of a stage are in flight together either way)
            bool ok = PW ? gather_offset_pw(g, rp[j], k0, off, c0) : gather_offset(g, rp[j], k0, off, c0);
            st.ra[j] = ok ? ld16(a_src + off) : zero8();
            st.ok[j] = ok;
            if (ok) st.c0 = c0;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int idx = tid + SF_THREADS * j;
            int brow = idx >> 2;
            int co = n0 + brow;
            bool ok = (idx < BN * 4) && (co < p.Nout) && (k0 < (uint32_t)g.Ktot);
            st.rb[j] = ok ? ld16(wmat + (int64_t)co * p.ldw + k0) : zero8();
        }
    };
    auto store_tile = [&](int buf, const Stage& st) {
        f16* As = smem + buf * (A_ELEMS + B_ELEMS);
        f16* Bs = As + A_ELEMS;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f16x8 v = st.ra[j];
            if (has_tf && st.ok[j]) v = bn_act8(v, s_scale + st.c0, s_shift + st.c0, act_lo);
            st16(As + lds_tile_off((tid >> 2) + 64 * j, kq), v);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int idx = tid + SF_THREADS * j;
            if (idx < BN * 4) st16(Bs + lds_tile_off(idx >> 2, kq), st.rb[j]);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const f16* As = smem + buf * (A_ELEMS + B_ELEMS);
        const f16* Bs = As + A_ELEMS;
        f16x8 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = ld16(As + lds_tile_off(wm * WM + i * 16 + (lane & 15), lane >> 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = ld16(Bs + lds_tile_off(wn * WN + j * 16 + (lane & 15), lane >> 4));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = SF_MFMA16(af[i], bf[j], acc[i][j]);
    };

    if constexpr (GL) {
        issue_tile(0, 0);
        SF_WAIT_VMEM();
        __syncthreads();
        for (int ks = 0; ks < p.ksteps; ++ks) {
            if (ks + 1 < p.ksteps) issue_tile(ks + 1, (ks + 1) & 1);
            compute(ks & 1);
            SF_WAIT_VMEM();
            __syncthreads();
        }
    } else {
        if (has_tf) __syncthreads();  // scale/shift tables visible before the first store_tile
        load_tile(0, st0);
        store_tile(0, st0);
        __syncthreads();
        for (int ks = 0; ks < p.ksteps; ++ks) {
            const bool more = ks + 1 < p.ksteps;
            if (more) load_tile(ks + 1, st0);
            compute(ks & 1);
            if (more) store_tile((ks + 1) & 1, st0);
            __syncthreads();
        }
    }

    // ---------------- epilogue: scale, bias, BatchNorm partial statistics (fp32, from the accumulators)
    {
        const float alpha = p.alpha != 0.f ? p.alpha : 1.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int colj = n0 + wn * WN + j * 16 + (lane & 15);
            const float b = (p.bias && colj < p.Nout) ? p.bias[colj] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = acc[i][j][r] * alpha + b;
        }
    }
    if constexpr (F32R) f32_rows_epilogue<TM, TN>(acc, p.f32, m0 + wm * WM, n0 + wn * WN, p.M, p.Nout, resid, p.ldr, p.resid_row0);
    if (p.stat_part) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // rows beyond M carry only the bias: keep them out of the statistics
                    const bool rowok = m0 + wm * WM + i * 16 + 4 * (lane >> 4) + r < p.M;
                    const float v = rowok ? acc[i][j][r] : 0.f;
                    s += v;
                    q += v * v;
                }
            s = wave_sum_over_row_groups(s);
            q = wave_sum_over_row_groups(q);
            if (lane < 16) {
                s_red[wm][0][wn * WN + j * 16 + lane] = s;
                s_red[wm][1][wn * WN + j * 16 + lane] = q;
            }
        }
    }
    // ---------------- stage the tile through LDS so that global stores are 16-byte, row-contiguous
    f16* stg = smem;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * WN + j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * WM + i * 16 + 4 * (lane >> 4) + r;
                stg[row * STG_LD + col] = (f16)acc[i][j][r];
            }
        }
    __syncthreads();
    if (p.stat_part && tid < BN) {
        const int col = n0 + tid;
        if (col < p.Nout) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES_M; ++w) {
                s += s_red[w][0][tid];
                q += s_red[w][1][tid];
            }
            p.stat_part[((int64_t)mt * 2 + 0) * p.Nout + col] = s;
            p.stat_part[((int64_t)mt * 2 + 1) * p.Nout + col] = q;
        }
    }
    constexpr int CG = BN / 8;
    static_assert(SF_THREADS % CG == 0 && 64 % CG == 0, "a thread keeps one column group over the whole store loop");
    const bool bnb = p.bnb_part != nullptr;
    float bsg[8], bsgy[8], bsc[8], bsh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bsg[e] = 0.f; bsgy[e] = 0.f; bsc[e] = 1.f; bsh[e] = 0.f; }
    if (bnb && p.bnb_y && !p.bnb_bits && n0 + (tid % CG) * 8 < p.Nout) { load8f(p.bnb_scale + n0 + (tid % CG) * 8, bsc); load8f(p.bnb_shift + n0 + (tid % CG) * 8, bsh); }
    constexpr int ITER = BM * CG / SF_THREADS > 0 ? BM * CG / SF_THREADS : 1;
    static_assert(BM * CG % SF_THREADS == 0 || BM * CG < SF_THREADS, "whole store iterations");
    // rows in flight per thread: 2 where the register cap leaves room (the 128-VGPR variants hold their accumulators in VGPRs,
    // dead by now); the others keep accumulators in AGPRs and have no spare VGPRs without losing a resident wave
    constexpr int CH = (BN >= 128 && OCC4) ? 2 : 1;
    static_assert(ITER % CH == 0, "whole chunks");
    const int ecg = tid % CG, ecol = n0 + ecg * 8;
    for (int it0 = 0; it0 < ITER; it0 += CH) {
        EpiLoads L[CH];
        bool ok[CH], rok[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int idx = tid + (it0 + u) * SF_THREADS;
            const int row = idx / CG, m = m0 + row;
            ok[u] = idx < BM * CG && m < p.M && ecol < p.Nout;
            rok[u] = ok[u] && resid && m >= p.resid_row0;
            if constexpr (F32R) {
                uint32_t srow;
                if (f32_row(p.f32, m, srow)) rok[u] = false;               // residual already inside the staged value
            }
            L[u].rbits = 0xffu; L[u].bbits = 0u;
            if (rok[u]) {
                L[u].r = ld16(resid + (int64_t)m * p.ldr + ecol);
                if (p.resid_bits) L[u].rbits = p.resid_bits[(int64_t)m * (p.Nout >> 3) + (ecol >> 3)];
            }
            if (ok[u] && bnb && p.bnb_y) {
                L[u].y = ld16(p.bnb_y + (int64_t)m * p.bnb_ld + ecol);
                if (p.bnb_bits) L[u].bbits = p.bnb_bits[(int64_t)m * (p.Nout >> 3) + (ecol >> 3)];
            }
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (!ok[u]) continue;
            const int idx = tid + (it0 + u) * SF_THREADS;
            const int row = idx / CG, m = m0 + row;
            f16x8 v = ld16(stg + row * STG_LD + ecg * 8);
            if (rok[u]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const f16 r = ((L[u].rbits >> e) & 1u) ? L[u].r[e] : (f16)0.f;
                    v[e] = (f16)((float)v[e] + (float)r);
                }
            }
            if (p.act_mode == 2) {
                const f16x8 h = ld16(p.act_aux + (int64_t)m * p.ld_aux + ecol);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (f16)((float)v[e] * gelu_df((float)h[e]));
            }
            if (p.act_mode == 3) {          // inference-fused convolution: ReLU after bias (+ residual)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > (f16)0.f ? v[e] : (f16)0.f;
            }
            st16(yout + (int64_t)m * p.ldy + ecol, v);
            if (p.act_mode == 1) {
                f16x8 a;
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = (f16)gelu_f((float)v[e]);
                st16(p.act_aux + (int64_t)m * p.ld_aux + ecol, a);
            }
            if (bnb && p.bnb_y) bnb_accumulate(v, L[u].y, bsc, bsh, bsg, bsgy, p.bnb_bits != nullptr, L[u].bbits);
            else if (bnb) {                 // plain column sums of the stored tile (bias gradient of the consumer Linear)
#pragma unroll
                for (int e = 0; e < 8; ++e) bsg[e] += (float)v[e];
            }
        }
    }
    if (bnb) bnb_reduce_store<4, CG>(bsg, bsgy, reinterpret_cast<float*>(smem), p.bnb_part + (int64_t)mt * 2 * p.Nout, n0, p.Nout);
}

// ---------------------------------------------------------------------------------------------
struct WgradParams {
    GatherSide g;       // gathered forward input (mode 0), columns k = tap*g.C + ci
    const f16* dy;      // [M][ldy]
    int ldy;
    int Co;
    int M;
    float* ws;          // split partials [splits][Co_pad][Kpad] fp32 (plain stores, reduced by sf_wgrad_reduce_kernel)
    int Co_pad, Kpad;
    int nchunks;        // ceil(M / 32)
    int chunks_per_split;
    int tiles_k, tiles_c;   // grid decomposition (1-D launch, XCD-remapped)
    // batched "TN" GEMM mode (attention dV / dK): blockIdx.z = b*bh + j selects the operands, one split, and the
    // tile is written directly as fp16: out[co][k] = out_scale * sum_m dy[m][co] * x[m][k]
    int bh;
    int64_t sp_b, sp_h, sx_b, sx_h, so_b, so_h;
    f16* out16; int ldo; float out_scale;
};

// dw[Co][Cw][taps] (PyTorch Conv3d weight layout) (+)= out_scale * sum_splits ws[s][co][tap*C + ci]
struct WgradReduceParams {
    const float* ws;
    int splits, Co, Co_pad, Kpad, Ktot;   // Co = rows of dw (real output channels)
    FastDiv fdC;
    float* dw;
    int Cw, taps;
    float out_scale;
    int accumulate;
    int lanes;          // split lanes per output element (power of two, 1..32)
    int slice4;         // 1: kcol = (slice*4 + chunk)*C + ci with three real taps per slice (sf_stem.h on 8-channel 3-wide kernels)
};

// 256 threads = E element quads x L split lanes (L = p.lanes, a power of two <= 32): lane z of a quad sums splits
// z, z+L, ... of four consecutive kcol (one 16-byte load per slab, four loads in flight), an LDS tree folds the L lanes
// (fixed order: the result does not depend on scheduling).  Consecutive quads are contiguous in every slab, so the
// E threads of one lane read E*16 contiguous bytes (a full 128-byte line for E >= 8).
__global__ __launch_bounds__(SF_THREADS) void sf_wgrad_reduce_kernel(WgradReduceParams p) {
    __shared__ f32x4 s_acc[SF_THREADS];
    const int L = p.lanes, E = SF_THREADS / L;
    const int e = threadIdx.x % E, z0 = threadIdx.x / E;
    const int64_t total = (int64_t)p.Co * p.Kpad;           // Kpad is a multiple of 128
    const int64_t slab = (int64_t)p.Co_pad * p.Kpad;
    const int64_t idx = ((int64_t)blockIdx.x * E + e) * 4;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    if (idx < total) {
        const float* src = p.ws + idx;          // element (co, kcol) sits at co*Kpad + kcol = idx in every slab
        int z = z0;
        for (; z + 3 * L < p.splits; z += 4 * L) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (int64_t)z * slab);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + (int64_t)(z + L) * slab);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + (int64_t)(z + 2 * L) * slab);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(src + (int64_t)(z + 3 * L) * slab);
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; z < p.splits; z += L) a0 += *reinterpret_cast<const f32x4*>(src + (int64_t)z * slab);
    }
    s_acc[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    for (int half = L >> 1; half >= 1; half >>= 1) {
        if (z0 < half) s_acc[threadIdx.x] += s_acc[threadIdx.x + half * E];
        __syncthreads();
    }
    if (z0 == 0 && idx < total) {
        const int co = (int)(idx / p.Kpad), kcol0 = (int)(idx % p.Kpad);
        const f32x4 r = s_acc[threadIdx.x];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kcol = kcol0 + j;
            if (kcol < p.Ktot) {
                uint32_t tap, ci;
                fd_divmod((uint32_t)kcol, p.fdC, tap, ci);
                if (p.slice4) {
                    if ((tap & 3u) == 3u) continue;
                    tap = (tap >> 2) * 3u + (tap & 3u);
                }
                if (ci < (uint32_t)p.Cw) {
                    float* dst = p.dw + ((int64_t)co * p.Cw + ci) * p.taps + tap;
                    const float v = r[j] * p.out_scale;
                    *dst = p.accumulate ? *dst + v : v;
                }
            }
        }
    }
}

template <class V>
__device__ __forceinline__ f16x4 as_f16x4(V v) {
    f16x4 o;
    __builtin_memcpy(&o, &v, 8);
    return o;
}

// Tap decomposition of one K-column group, hoisted out of the position loop of the weight-gradient kernel
// (a loader thread keeps its column group for the whole kernel).
struct TapPos {
    int dt, dh, dw;     // tap offsets in source coordinates (kt*dilT, kh*dilH, kw*dilW)
    uint32_t c0;        // first channel of the 8-channel group
    bool valid;         // column group < Ktot
};
__device__ __forceinline__ TapPos decode_tap(const GatherSide& g, uint32_t k0) {
    TapPos t;
    t.valid = k0 < (uint32_t)g.Ktot;
    uint32_t tap, kt, kh, kw, q;
    fd_divmod(t.valid ? k0 : 0u, g.fdC, tap, t.c0);
    fd_divmod(tap, g.fdkW, q, kw);
    fd_divmod(q, g.fdkH, kt, kh);
    t.dt = (int)kt * g.dilT; t.dh = (int)kh * g.dilH; t.dw = (int)kw * g.dilW;
    return t;
}
// mode-0 gather (rows are conv outputs) of a pre-decoded tap
__device__ __forceinline__ bool gather_offset_tap(const GatherSide& g, const RowPos& r, const TapPos& tp, int64_t& off) {
    if (!r.valid || !tp.valid) return false;
    const int t = r.bt + tp.dt, h = r.bh + tp.dh, w = r.bw + tp.dw;
    if ((unsigned)t >= (unsigned)g.sT || (unsigned)h >= (unsigned)g.sH || (unsigned)w >= (unsigned)g.sW) return false;
    off = ((((int64_t)r.n * g.sT + t) * g.sH + h) * g.sW + w) * (int64_t)g.ld + tp.c0;
    return true;
}

// Weight gradient: dW[co][k] = sum_m dY[m][co] * A(m, k), one BMW x 128 tile of dW per workgroup, reduction over
// the positions of one split.  A stage stages KS*32 positions of both operands in LDS.
//   KS == 1: two LDS buffers, global loads of stage s+1 in flight under the MFMAs of stage s (MFMA-bound tiles);
//   KS  > 1: small-Co tiles are HBM-bound and latency-limited -- a long stage (128 positions) amortises the
//            barrier and the load latency, one LDS buffer + register staging keeps 3 workgroups per CU.
// MFMA fragments come from ds_read_b64_tr_b16 (hardware transpose read).
template <int BMW, int WM, int WN, int KS>
__global__ __launch_bounds__(SF_THREADS) void sf_wgrad_kernel(WgradParams p) {
    constexpr int BNW = 128, BKM = 32, ROWS = BKM * KS;
    constexpr int WAVES_N = BNW / WN, WAVES_M = BMW / WM;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int LDA = BMW + 16, LDB = BNW + 16;
    constexpr int BUF = ROWS * (LDA + LDB);
    constexpr int NBUF = KS == 1 ? 2 : 1;
    constexpr int NA = (ROWS * (BMW / 8) + SF_THREADS - 1) / SF_THREADS;
    constexpr int NX = 2 * KS;

    __shared__ __attribute__((aligned(16))) f16 smem[NBUF * BUF];
    __shared__ __attribute__((aligned(16))) float s_scale[512];
    __shared__ __attribute__((aligned(16))) float s_shift[512];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // 1-D grid over (tile_k fastest, tile_c, split | batch), XCD-remapped: the tiles of one split share dy / x panels
    const uint32_t wg = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = (int)(wg % (uint32_t)p.tiles_k);
    const int by = (int)((wg / (uint32_t)p.tiles_k) % (uint32_t)p.tiles_c);
    const int bz = (int)(wg / ((uint32_t)p.tiles_k * (uint32_t)p.tiles_c));
    const int n0 = bx * BNW, c0 = by * BMW;
    const GatherSide& g = p.g;
    const bool has_tf = g.scale != nullptr;
    if (has_tf) {
        for (int c = tid; c < g.C; c += SF_THREADS) {
            s_scale[c] = g.scale[c];
            s_shift[c] = g.shift[c];
        }
    }
    const f16* x_src = g.src;
    const f16* dy_src = p.dy;
    f16* out16 = p.out16;
    const int nstages = (p.nchunks + KS - 1) / KS;
    // this split's stages (chunks_per_split counts 32-position chunks and is a multiple of KS)
    int sb = bz * (p.chunks_per_split / KS);
    int se = sb + p.chunks_per_split / KS;
    if (p.bh > 0) {
        const int zb = bz / p.bh, zj = bz % p.bh;
        dy_src += zb * p.sp_b + zj * p.sp_h;
        x_src += zb * p.sx_b + zj * p.sx_h;
        out16 += zb * p.so_b + zj * p.so_h;
        sb = 0;
        se = nstages;
    }
    if (se > nstages) se = nstages;

    f16x8 ra[NA], rb[NX];
    bool rb_ok[NX];
    const TapPos tp = decode_tap(g, (uint32_t)(n0 + (tid & 15) * 8));

    auto load_tile = [&](int stage) {
        const int mbase = stage * ROWS;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            int idx = tid + SF_THREADS * j;
            int ml = idx / (BMW / 8), cg = idx % (BMW / 8);
            int m = mbase + ml, co = c0 + cg * 8;
            bool ok = (idx < ROWS * (BMW / 8)) && (m < p.M) && (co < p.Co);
            ra[j] = ok ? ld16(dy_src + (int64_t)m * p.ldy + co) : zero8();
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            int m = mbase + (tid >> 4) + 16 * j;
            RowPos rp = decode_row(g, (uint32_t)m, m < p.M);
            int64_t off;
            bool ok = gather_offset_tap(g, rp, tp, off);
            rb[j] = ok ? ld16(x_src + off) : zero8();
            rb_ok[j] = ok;
        }
    };
    auto store_tile = [&](int buf) {
        f16* Ys = smem + buf * BUF;
        f16* Xs = Ys + ROWS * LDA;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            int idx = tid + SF_THREADS * j;
            if (idx < ROWS * (BMW / 8)) st16(Ys + (idx / (BMW / 8)) * LDA + (idx % (BMW / 8)) * 8, ra[j]);
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            f16x8 v = rb[j];
            if (has_tf && rb_ok[j]) v = bn_act8(v, s_scale + tp.c0, s_shift + tp.c0, g.relu ? 0.f : -INFINITY);
            st16(Xs + ((tid >> 4) + 16 * j) * LDB + (tid & 15) * 8, v);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int pl = lane & 15, g4 = lane >> 4;
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16* Ys = smem + buf * BUF + ks * BKM * LDA;
            const f16* Xs = smem + buf * BUF + ROWS * LDA + ks * BKM * LDB;
            f16x8 af[TM], bf[TN];
            {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f16* ptr = Ys + (8 * g4 + 4 * h + (pl >> 2)) * LDA + wm * WM + i * 16 + 4 * (pl & 3);
                        f16x4 t = as_f16x4(SF_LDS_TR16(ptr));
                        af[i][4 * h + 0] = t[0]; af[i][4 * h + 1] = t[1]; af[i][4 * h + 2] = t[2]; af[i][4 * h + 3] = t[3];
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f16* ptr = Xs + (8 * g4 + 4 * h + (pl >> 2)) * LDB + wn * WN + j * 16 + 4 * (pl & 3);
                        f16x4 t = as_f16x4(SF_LDS_TR16(ptr));
                        bf[j][4 * h + 0] = t[0]; bf[j][4 * h + 1] = t[1]; bf[j][4 * h + 2] = t[2]; bf[j][4 * h + 3] = t[3];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = SF_MFMA16(af[i], bf[j], acc[i][j]);
        }
    };

    if (has_tf) __syncthreads();
    if (sb < se) {
        load_tile(sb);
        store_tile(0);
    }
    __syncthreads();
    for (int st = sb; st < se; ++st) {
        const bool more = st + 1 < se;
        if (more) load_tile(st + 1);
        if constexpr (NBUF == 2) {
            compute((st - sb) & 1);
            if (more) store_tile((st - sb + 1) & 1);
            __syncthreads();
        } else {
            compute(0);
            __syncthreads();            // every wave is done reading the stage
            if (more) store_tile(0);
            __syncthreads();
        }
    }
    if (p.bh > 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int kcol = n0 + wn * WN + j * 16 + pl;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = c0 + wm * WM + i * 16 + 4 * g4 + r;
                    if (co < p.Co && kcol < g.Ktot) out16[(int64_t)co * p.ldo + kcol] = (f16)(acc[i][j][r] * p.out_scale);
                }
        }
        return;
    }
    // every split owns its slab: plain (non-atomic) stores, also when it had no rows to reduce (zeros)
    float* slab = p.ws + (int64_t)bz * p.Co_pad * p.Kpad;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int kcol = n0 + wn * WN + j * 16 + pl;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = c0 + wm * WM + i * 16 + 4 * g4 + r;
                slab[(int64_t)co * p.Kpad + kcol] = acc[i][j][r];
            }
    }
}
