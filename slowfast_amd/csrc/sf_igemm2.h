// Implicit-GEMM convolution, second generation (gfx950): every operand travels global -> LDS directly.
//
// Same contraction as sf_igemm.h (reference call sites: slowfast/models/resnet_helper.py:331-369 BottleneckTransform a/b/c,
// :485-493 ResBlock.branch1, video_model_builder.py:147-154 FuseFastToSlow.conv_f2s; nn.Linear of attention.py / common.py):
//   Y[m, n] = sum_{tap, c} SRC[pos(m) + delta(tap)][c] * W[n][wcol(tap) + c]
// What changed against the first kernel (profiles/r2/r2_*: its loaders were VALU-bound -- three magic-number divisions, bounds
// checks and a BatchNorm transform per 16-byte operand group and K step -- and every 32-deep K step ended in vmcnt(0) + barrier):
//   * a K step never straddles a tap (C % BK == 0), so the tap of a step is WAVE-UNIFORM: one scalar table lookup per step,
//     no per-lane tap decomposition.  A lane keeps, for the rows it copies, the element offset of the row's base position and
//     a bit mask "tap t stays inside the source" (decoded once, before the loop).
//   * operands are copied with global_load_lds_dwordx4 (per-lane source address, lane-linear LDS image): padding taps read a
//     16-byte line of zeros instead of the tensor.  No staging registers, no ds_write pass.  The XOR bank swizzle of the
//     ds_read_b128 fragment reads is applied on the SOURCE side (which 16-byte K slot a lane fetches).
//   * BK = 64 (one 128-byte line per row and step), a ring of THREE stages and counted s_waitcnt vmcnt: two stages are in
//     flight while one is multiplied; ONE raw s_barrier per step.
//   * 256 x BN tiles, 8 waves (2 per SIMD): (256 + BN) * 64 * 2 bytes per 256 * BN * 64 MACs -- 48 B/clk/CU at BN = 128,
//     under what the L2 delivers (~56 B/clk/CU); the 128 x 128 x 32 tile needed 64 B/clk/CU.
// The producer's BatchNorm + ReLU is NOT applied here: engine.ResBlockFn materialises relu(bn(y)) once (engine.MATERIALIZE).
#pragma once
#include "sf_common.h"

#define SF_I2_MAXTAPS 32


// padding taps read sf_zero_line (sf_common.h)

struct Igemm2Tap {
    int32_t dlin;       // linear source-position offset of the tap: (dt*sH + dh)*sW + dw
    int32_t wcol;       // first column of the tap in the weight matrix
};

struct Igemm2Params {
    const f16* src;     // gathered operand, rows of C channels at pitch ld
    int ld, C;
    int sT, sH, sW;     // source extents
    FastDiv fdrW, fdrH, fdrT;               // row -> (n, a, b, c) over the row space
    int mulT, mulH, mulW, offT, offH, offW; // base source coordinate of a row: a*mulT + offT, ...
    int ntaps;
    int8_t dt[SF_I2_MAXTAPS], dh[SF_I2_MAXTAPS], dw[SF_I2_MAXTAPS];    // tap displacement (source coordinates)
    Igemm2Tap taps[SF_I2_MAXTAPS];
    int M;
    const f16* wmat;    // [Nout][ldw]
    int ldw, Nout;
    f16* y;
    int ldy;
    const float* bias;
    const f16* resid;
    int ldr;
    float* stat_part;   // [ceil(M / 128)][2][Nout]
    int ntiles_n;
    int act_mode;
    f16* act_aux;
    int ld_aux;
    int resid_row0;
    float alpha;
    const uint8_t* resid_bits;  // optional [rows][Nout/8] bit mask of the residual (see IgemmParams)
    // output-row map (strided data gradients run one launch per stride-residue class of input positions, see
    // launch_igemm2_strided_dgrad): row (n, a, b, c) of the class is stored at position
    // ((n*oT + a*omT + ooT)*oH + b*omH + ooH)*oW + c*omW + ooW of y / resid.  omap == 0: rows are stored densely.
    int omap;
    int oT, oH, oW, omT, omH, omW, ooT, ooH, ooW;
    // fused BatchNorm-backward reduction (data gradients only; see IgemmParams): per-tile column sums of g and g * bnb_y,
    // g = stored output masked by (bnb_y * bnb_scale + bnb_shift > 0), into bnb_part[mt][2][Nout]
    const f16* bnb_y; int bnb_ld;
    const float* bnb_scale; const float* bnb_shift;
    float* bnb_part;
    const uint8_t* bnb_bits;        // optional [rows][Nout/8] bit mask replacing the recomputed one (block-output ReLU)
    // DIAGNOSTIC (SF_IGEMM2_ABLATE, tools/microbench.py only; results are garbage): bit 0 no copies inside the K loop, bit 1 no
    // LDS reads / MFMAs, bit 2 LDS reads but no MFMAs, bit 3 return before the epilogue, bit 4 no barrier inside the K loop
    int ablate;
    F32Rows f32;        // fp32 side rows of the output (token residual sums; sf_common.h), f32.out == nullptr: off
    int linear;         // host-side hint: a plain matrix product (tile choice, sf_api.hip)
};

// LDS operand tile [rows][BK] fp16; the 16-byte K slot of a row is XOR-swizzled so that the 16 lanes one ds_read_b128 phase
// serves hit 16 distinct bank slots (MI355X_MICROARCH.md, LDS table; checked for every lane group by tools/lds_swizzle_check.py)
template <int BK>
__device__ __forceinline__ int i2_lds_off(int row, int kslot) {
    if constexpr (BK == 64) return row * 64 + ((kslot ^ (row & 7)) << 3);
    else return lds_tile_off(row, kslot);
}

// Epilogue of the second-generation kernel (sf_igemm2_kernel; a function of its own since the round-5 sf_igemm3 experiment shared it): alpha / bias, fp32 side rows, BatchNorm
// partial statistics from the accumulators, the tile staged through LDS (the operand stages are dead by now; the caller has
// passed the workgroup barrier that ends its K loop) and stored in 16-byte row-contiguous pieces with residual (+ bit mask), GELU
// epilogues, the fused BatchNorm-backward reduction.  acc[i][j] = rows wm * WM + i * 16 .., columns wn * WN + j * 16 .. of the tile.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool F32R>
__device__ __forceinline__ void i2_epilogue(const Igemm2Params& p, f32x4 (&acc)[BM / WAVES_M / 16][BN / WAVES_N / 16], f16* smem,
                                            float (*s_red)[2][BN], int* s_orow, int mt, int nt) {
    constexpr int NW = WAVES_M * WAVES_N, NT = 64 * NW;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int STG_LD = BN + 8;
    constexpr int HALVES = BM / 128, WPH = WAVES_M / HALVES;
    static_assert(WAVES_M % HALVES == 0, "a wave row belongs to one 128-row group");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int m0 = mt * BM, n0 = nt * BN;
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
    if constexpr (F32R) f32_rows_epilogue<TM, TN>(acc, p.f32, m0 + wm * WM, n0 + wn * WN, p.M, p.Nout, p.resid, p.ldr, p.resid_row0);
    if (p.stat_part) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
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
    if (p.stat_part && tid < HALVES * BN) {
        const int half = tid / BN, c = tid % BN;
        const int col = n0 + c;
        const int prow = mt * HALVES + half;                        // statistics rows are 128 positions each
        if (col < p.Nout && (int64_t)prow * 128 < p.M) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WPH; ++w) {
                s += s_red[half * WPH + w][0][c];
                q += s_red[half * WPH + w][1][c];
            }
            p.stat_part[((int64_t)prow * 2 + 0) * p.Nout + col] = s;
            p.stat_part[((int64_t)prow * 2 + 1) * p.Nout + col] = q;
        }
    }
    constexpr int CG = BN / 8;
    static_assert(NT % CG == 0 && 64 % CG == 0, "a thread keeps one column group over the whole store loop");
    const bool bnb = p.bnb_part != nullptr;
    float bsg[8], bsgy[8], bsc[8], bsh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bsg[e] = 0.f; bsgy[e] = 0.f; bsc[e] = 1.f; bsh[e] = 0.f; }
    if (bnb && p.bnb_y && !p.bnb_bits && n0 + (tid % CG) * 8 < p.Nout) { load8f(p.bnb_scale + n0 + (tid % CG) * 8, bsc); load8f(p.bnb_shift + n0 + (tid % CG) * 8, bsh); }
    constexpr int ITER = BM * CG / NT;
    static_assert(BM * CG % NT == 0 && ITER >= 1, "whole store iterations");
    constexpr int CH = ITER < 2 ? ITER : 2;
    static_assert(ITER % CH == 0, "whole chunks");
    const int ecg = tid % CG, ecol = n0 + ecg * 8;
    for (int it0 = 0; it0 < ITER; it0 += CH) {
        EpiLoads L[CH];
        bool ok[CH], rok[CH];
        int mo[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int idx = tid + (it0 + u) * NT;
            const int row = idx / CG, mr = m0 + row;
            ok[u] = mr < p.M && ecol < p.Nout;
            const int m = ok[u] ? (p.omap ? s_orow[row] : mr) : 0;
            mo[u] = m;
            rok[u] = ok[u] && p.resid && m >= p.resid_row0;
            if constexpr (F32R) {
                uint32_t srow;
                if (f32_row(p.f32, m, srow)) rok[u] = false;               // residual already inside the staged value
            }
            L[u].rbits = 0xffu; L[u].bbits = 0u;
            if (rok[u]) {
                L[u].r = ld16(p.resid + (int64_t)m * p.ldr + ecol);
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
            const int idx = tid + (it0 + u) * NT;
            const int row = idx / CG, m = mo[u];
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
            if (p.act_mode == 3) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > (f16)0.f ? v[e] : (f16)0.f;
            }
            st16(p.y + (int64_t)m * p.ldy + ecol, v);
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
    if (bnb) bnb_reduce_store<NW, CG>(bsg, bsgy, reinterpret_cast<float*>(smem), p.bnb_part + (int64_t)mt * 2 * p.Nout, n0, p.Nout);
}

// Tried in round 3 and removed again (the code is in the history, commits c9ff12f / 6f38f0a; evidence under profiles/):
//  * a STRIP variant -- ONE staged strip of source rows per channel chunk serves every tap, a tap is a row offset into it,
//    padding is masked in the fragment: 6x fewer gathered bytes on 3x3 layers, and 15-30 % SLOWER on every eligible layer
//    (61 VALU instructions per K step against 21; profiles/r3/r3_v5_strip_ab.txt);
//  * issuing the next stage's copies between the two MFMA halves of a stage in half of the waves: a wash
//    (profiles/r3/r3_v7_stagger_ab.txt).  profiles/r3/r3_v6_igemm2_ablation.md shows what bounds the loop instead: the copy stream
//    alone and the MFMA stream alone each take 75-80 % of the kernel's time and overlap only partly.
template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int NST, bool F32R = false>   // F32R: see sf_igemm_kernel
// register cap: the 32-deep variant must fit TWO workgroups per CU (4 waves per SIMD -> 128 VGPRs), the 64-deep one runs alone
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (BK == 32 ? 2 : 1) * (WAVES_M * WAVES_N) / 4) void sf_igemm2_kernel(Igemm2Params p) {
    constexpr int NW = WAVES_M * WAVES_N, NT = 64 * NW;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int KSL = BK / 8;                 // 16-byte slots per row
    constexpr int RPI = 64 / KSL;               // rows one wave instruction copies (8 at BK = 64, 16 at BK = 32)
    constexpr int NA = BM / RPI / NW;           // A copies per wave and stage
    constexpr int NBI = BN / RPI;               // B copy instructions per stage (all waves together)
    constexpr int NB = (NBI + NW - 1) / NW;
    static_assert(BM % (RPI * NW) == 0, "A tile: whole copy instructions per wave");
    static_assert(NBI % NW == 0 || NBI < NW, "B tile: uniform copy count per wave or one partial round");
    static_assert(BM % 128 == 0, "statistics are kept per 128 rows");
    constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK, STAGE = A_ELEMS + B_ELEMS;
    constexpr int STG_LD = BN + 8;
    constexpr int SMEM_MAIN = NST * STAGE, SMEM_STG = BM * STG_LD;
    constexpr int SMEM = SMEM_MAIN > SMEM_STG ? SMEM_MAIN : SMEM_STG;
    constexpr int HALVES = BM / 128, WPH = WAVES_M / HALVES;    // 128-row statistic groups, wave rows per group
    static_assert(WAVES_M % HALVES == 0, "a wave row belongs to one 128-row group");

    // ONE LDS object (hipcc serialises direct-to-LDS copies against ds_reads of any other __shared__ object)
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[SMEM * 2 + WAVES_M * 2 * BN * 4 + BM * 4];
    f16* const smem = reinterpret_cast<f16*>(lds_raw);
    float (*const s_red)[2][BN] = reinterpret_cast<float (*)[2][BN]>(lds_raw + SMEM * 2);
    int* const s_orow = reinterpret_cast<int*>(lds_raw + SMEM * 2 + WAVES_M * 2 * BN * 4);   // output row of a tile row

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tile = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int nt = tile % p.ntiles_n, mt = tile / p.ntiles_n;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- loader state: the rows this lane copies in every stage (instruction j of the wave -> rows (wave + NW*j)*RPI ..)
    const int lrow = lane / KSL;                                    // row inside a copy instruction
    int kslot;                                                      // logical 16-byte K slot this lane fetches
    if constexpr (BK == 64) kslot = (lane & 7) ^ (lrow & 7);
    else kslot = ((lane & 3) - 2 * ((lrow >> 2) & 3)) & 3;          // inverse of lds_tile_off()'s rotation
    int64_t aoff[NA];
    uint32_t amask[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        int m = m0 + (wave + NW * j) * RPI + lrow;
        if (m >= p.M) m = p.M - 1;                                  // clamped rows are computed and never stored
        uint32_t q, a, b, c, n;
        fd_divmod((uint32_t)m, p.fdrW, q, c);
        fd_divmod(q, p.fdrH, q, b);
        fd_divmod(q, p.fdrT, n, a);
        const int bt = (int)a * p.mulT + p.offT, bh = (int)b * p.mulH + p.offH, bw = (int)c * p.mulW + p.offW;
        aoff[j] = ((((int64_t)n * p.sT + bt) * p.sH + bh) * p.sW + bw) * (int64_t)p.ld + kslot * 8;
        uint32_t mk = 0;
        for (int t = 0; t < p.ntaps; ++t) {
            const int st = bt + p.dt[t], sh = bh + p.dh[t], sw = bw + p.dw[t];
            const bool ok = (unsigned)st < (unsigned)p.sT && (unsigned)sh < (unsigned)p.sH && (unsigned)sw < (unsigned)p.sW;
            mk |= (ok ? 1u : 0u) << t;
        }
        amask[j] = mk;
    }
    if (p.omap) {
        for (int r = tid; r < BM; r += NT) {
            int m = m0 + r;
            if (m >= p.M) m = p.M - 1;
            uint32_t q, a, b, c, n;
            fd_divmod((uint32_t)m, p.fdrW, q, c);
            fd_divmod(q, p.fdrH, q, b);
            fd_divmod(q, p.fdrT, n, a);
            s_orow[r] = (((int)n * p.oT + (int)a * p.omT + p.ooT) * p.oH + (int)b * p.omH + p.ooH) * p.oW + (int)c * p.omW + p.ooW;
        }
    }
    const f16* bptr[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        int co = n0 + (wave + NW * j) * RPI + lrow;
        if (co >= p.Nout) co = p.Nout - 1;
        bptr[j] = p.wmat + (int64_t)co * p.ldw + kslot * 8;
    }
    const f16* const zline = reinterpret_cast<const f16*>(sf_zero_line);
    const int csteps = p.C / BK;                                    // K steps per tap
    const int ksteps = p.ntaps * csteps;

    // stage (tap, channel chunk c0) -> LDS buffer `buf`
    auto issue = [&](int tap, int c0, int buf) {
        f16* As = smem + buf * STAGE;
        f16* Bs = As + A_ELEMS;
        const int64_t dsrc = (int64_t)p.taps[tap].dlin * p.ld + c0;
        const int wk = p.taps[tap].wcol + c0;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const f16* g = ((amask[j] >> tap) & 1u) ? p.src + (aoff[j] + dsrc) : zline;
            SF_GLOBAL_LOAD_LDS16(g, As + (wave + NW * j) * 512);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (NBI % NW == 0 || wave + NW * j < NBI) SF_GLOBAL_LOAD_LDS16(bptr[j] + wk, Bs + (wave + NW * j) * 512);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const f16* As = smem + buf * STAGE;
        const f16* Bs = As + A_ELEMS;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            f16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = ld16(As + i2_lds_off<BK>(wm * WM + i * 16 + (lane & 15), kk * 4 + (lane >> 4)));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = ld16(Bs + i2_lds_off<BK>(wn * WN + j * 16 + (lane & 15), kk * 4 + (lane >> 4)));
            if (SF_ABLATE(p) & 4) {
#pragma unroll
                for (int i = 0; i < TM; ++i) SF_KEEP_ALIVE(af[i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) SF_KEEP_ALIVE(bf[j]);
                continue;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = SF_MFMA16(af[i], bf[j], acc[i][j]);
        }
    };

    // ---- main loop: stages ks + 1 (and ks + 2 at NST == 3) are in flight while stage ks is multiplied
    {
        // copies one wave issues per stage: the count s_waitcnt vmcnt leaves outstanding (uniform over the waves whenever
        // NBI % NW == 0; a partial last round only makes some waves wait for one copy more than necessary)
        constexpr int COPIES = NA + (NBI % NW == 0 ? NB : NB - 1);
        // K order: channel chunk OUTER, tap INNER -- the taps of a chunk re-read (shifted) the same 128-byte lines of the
        // rows around the tile, so 8 of 9 gathers of a 3x3 convolution hit the caches instead of walking the whole row
        // set once per tap (reuse distance rows * 128 B instead of rows * C * 2 B)
        int tap_i = 0, c_i = 0;                                     // (tap, chunk) of the next stage to issue
        auto advance = [&]() { if (++tap_i == p.ntaps) { tap_i = 0; c_i += BK; } };
        int issued = 0;
        for (; issued < NST - 1 && issued < ksteps; ++issued) { issue(tap_i, c_i, issued); advance(); }
        int cur = 0, nxt = NST - 1;
        for (int ks = 0; ks < ksteps; ++ks) {
            if constexpr (NST == 3) {
                if (ks + 1 < ksteps) SF_WAIT_VMEM_N(COPIES);        // stage ks landed, stage ks + 1 may still be in flight
                else SF_WAIT_VMEM();
            } else {
                SF_WAIT_VMEM();
            }
            if (!(SF_ABLATE(p) & 16)) SF_BARRIER_KEEP_VMEM();           // ... for every wave; stage ks - 1 is no longer read
            if (issued < ksteps) { if (!(SF_ABLATE(p) & 1)) issue(tap_i, c_i, nxt); advance(); ++issued; }
            if (!(SF_ABLATE(p) & 2)) compute(cur);
            cur = cur == NST - 1 ? 0 : cur + 1;
            nxt = nxt == NST - 1 ? 0 : nxt + 1;
        }
        __syncthreads();                                            // the epilogue staging reuses the operand buffers
        if (SF_ABLATE(p) & 8) return;
    }

    i2_epilogue<BM, BN, WAVES_M, WAVES_N, F32R>(p, acc, smem, s_red, s_orow, mt, nt);
}
