// Weight gradient of the implicit-GEMM convolutions, second generation (gfx950).
//
//   dW[co][tap*C + ci] = sum_m dY[m][co] * X[pos(m) + delta(tap)][ci]          (nn.Conv3d backward w.r.t. weight:
//   BottleneckTransform a/b/c resnet_helper.py:331-369, ResBlock.branch1 :485-493, FuseFastToSlow video_model_builder.py:147-154)
//
// A "TN" GEMM whose reduction axis is the position axis m (12 544 ... 802 816 rows) and whose output is small, so the
// parallelism comes from splitting m; each split leaves an fp32 partial tile and sf_wgrad_reduce_kernel sums them in a
// fixed order.  What changed against sf_wgrad_kernel (sf_igemm.h), whose loaders decoded every row with three magic
// divisions per 32-row step, staged both operands through registers and ended each step in a full barrier:
//   * a ROW TABLE (sf_wgrad2_rowtab_kernel, one 8-byte entry per output position: linear source position of the row's
//     base coordinate + a bit mask "tap t stays inside the source") is built once per call into the caller's workspace;
//     the main loop reads it with SCALAR loads (rows of a copy instruction are wave-uniform), one step ahead;
//   * both operands travel global -> LDS directly (global_load_lds_dwordx4), padding taps and rows beyond the split read a
//     line of zeros; three stages, counted vmcnt, one raw barrier per 32-row step; two workgroups per CU;
//   * both operands are position-major in memory ([m][channel]), the MFMA wants 8 consecutive m per lane: fragments come
//     from ds_read_b64_tr_b16 (hardware transpose read).  The lane-linear LDS image is made conflict-free by an XOR of
//     the 32-byte column pair with (m & 3) | ((m >> 3) & 1) << 2, applied on the SOURCE side;
//   * 128 (co) x 256 (k) tiles: operand bytes per MAC 25 % below the 128 x 128 tile, half as many split partials.
#pragma once
#include "sf_common.h"
#include "sf_igemm2.h"

struct __attribute__((aligned(8))) i32x2 { int32_t x, y; };
typedef int32_t i32x8 __attribute__((ext_vector_type(8)));      // four consecutive row-table entries

struct Wgrad2Params {
    const f16* x; int ldx, C;           // forward input, rows of C channels
    const f16* dy; int ldy, Co;
    int M;                              // output positions
    int Ktot;                           // taps * C
    const i32x2* rowtab;                 // [M] {base position, tap mask}
    int32_t dlin[SF_I2_MAXTAPS];        // linear source-position offset of tap t
    float* ws;                          // split partials [splits][Co_pad][Kpad] fp32
    int Co_pad, Kpad;
    int tiles_k, tiles_c;
    int rows_per_split;                 // multiple of 32
    int stage_stride;                   // thin kernel: 0 = a split is one contiguous run of rows_per_split positions; S > 0 =
                                        // round robin, split z takes the 128-position stages z, z + S, z + 2S, ... (at any
                                        // moment the resident workgroups then stream through ONE contiguous region)
};

struct RowtabParams {
    i32x2* tab;
    int M;
    FastDiv fdW, fdH, fdT;              // output extents
    int sT, sH, sW;                     // source (input) extents
    int strT, strH, strW, padT, padH, padW;
    int ntaps;
    int8_t dt[SF_I2_MAXTAPS], dh[SF_I2_MAXTAPS], dw[SF_I2_MAXTAPS];
};

__global__ __launch_bounds__(SF_THREADS) void sf_wgrad2_rowtab_kernel(RowtabParams p) {
    const int m = blockIdx.x * SF_THREADS + threadIdx.x;
    if (m >= p.M) return;
    uint32_t q, a, b, c, n;
    fd_divmod((uint32_t)m, p.fdW, q, c);
    fd_divmod(q, p.fdH, q, b);
    fd_divmod(q, p.fdT, n, a);
    const int bt = (int)a * p.strT - p.padT, bh = (int)b * p.strH - p.padH, bw = (int)c * p.strW - p.padW;
    uint32_t mk = 0;
    for (int t = 0; t < p.ntaps; ++t) {
        const int st = bt + p.dt[t], sh = bh + p.dh[t], sw = bw + p.dw[t];
        const bool ok = (unsigned)st < (unsigned)p.sT && (unsigned)sh < (unsigned)p.sH && (unsigned)sw < (unsigned)p.sW;
        mk |= (ok ? 1u : 0u) << t;
    }
    i32x2 e;
    e.x = (((int)n * p.sT + bt) * p.sH + bh) * p.sW + bw;
    e.y = (int)mk;
    p.tab[m] = e;
}

// swizzle of the 32-byte column pair of row m inside its 256-byte window (see the header comment)
__device__ __forceinline__ int w2_swz(int m) { return (m & 3) | (((m >> 3) & 1) << 2); }

// BMW = 128: 8 waves as 2 (co) x 4 (k), wave tile 64 x 64.   BMW = 64: 8 waves as 1 x 8, wave tile 64 x 32.
// Three-stage LDS ring (72 KB at BMW = 128): two workgroups per CU.
// DUAL (round 6): ONE 1024-thread workgroup carries two splits of the same tile -- waves 0-7 and 8-15 are two copies of the
// kernel above (own ring, own position range 2*bz and 2*bz + 1, the workgroup barrier shared) and at the end the halves exchange
// half of their accumulators through LDS (the rings are dead by then: 128 KB), add and store ONE partial tile.  Occupancy and
// the loop are what they were (16 waves and 144 KB per CU either way); the fp32 split partials written here and re-read by
// sf_wgrad_reduce_kernel are halved -- they were 1.5x the algorithmic bytes of the layer for three rounds
// (profiles/pmc_traffic_SLOWFAST_8x8_R50.json).  own + other is commutative: both halves of the tile see the same sum order.
template <int BMW, bool DUAL = false>
__global__ __launch_bounds__(DUAL ? 1024 : 512, 4) void sf_wgrad2_kernel(Wgrad2Params p) {
    constexpr int BKW = 256, ROWS = 32, NW = 8, NST = 3;
    constexpr int WAVES_C = BMW / 64, WAVES_K = NW / WAVES_C;
    constexpr int WN = BKW / WAVES_K;                   // 64 or 32 columns of k per wave
    constexpr int TM = 4, TN = WN / 16;
    constexpr int Y_ELEMS = ROWS * BMW, X_ELEMS = ROWS * BKW, STAGE = Y_ELEMS + X_ELEMS;
    constexpr int YI = Y_ELEMS / 512;                   // dY copy instructions per stage (8 or 4), 512 halfs = 1 KB each
    constexpr int YRPI = 512 / BMW;                     // dY rows per instruction (4 or 8)
    constexpr int YCH = BMW / 8;                        // 16-byte chunks per dY row (16 or 8)
    __shared__ __attribute__((aligned(16))) f16 smem_all[(DUAL ? 2 : 1) * NST * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = DUAL ? wave_all >> 3 : 0, wave = wave_all & 7;
    f16* const smem = smem_all + half * (NST * STAGE);
    const int wc = wave / WAVES_K, wk = wave % WAVES_K;
    const uint32_t wg = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = (int)(wg % (uint32_t)p.tiles_k);
    const int by = (int)((wg / (uint32_t)p.tiles_k) % (uint32_t)p.tiles_c);
    const int bz = (int)(wg / ((uint32_t)p.tiles_k * (uint32_t)p.tiles_c));
    const int k0 = bx * BKW, c0 = by * BMW;
    const int r0 = (DUAL ? 2 * bz + half : bz) * p.rows_per_split;
    int r1 = r0 + p.rows_per_split;
    if (r1 > p.M) r1 = p.M;
    const int nsteps = r1 > r0 ? (r1 - r0 + ROWS - 1) / ROWS : 0;
    // DUAL: both halves pass the same barriers -- as many as the first half (never the shorter one) has steps
    int nsteps_wg = nsteps;
    if constexpr (DUAL) {
        const int ra = 2 * bz * p.rows_per_split;
        int rb = ra + p.rows_per_split;
        if (rb > p.M) rb = p.M;
        nsteps_wg = rb > ra ? (rb - ra + ROWS - 1) / ROWS : 0;
    }
    const f16* const zline = reinterpret_cast<const f16*>(sf_zero_line);

    // ---- X loader: instruction j of the wave (2 per stage) copies stage rows 2*(wave + 8j) + {0, 1}; lane -> row bit
    // lane >> 5, physical 16-byte chunk lane & 31 of the 512-byte row; the logical chunk un-does the pair swizzle
    const int xrr = lane >> 5, xpc = lane & 31;
    int64_t xcol[2];        // channel offset + tap displacement (elements) of the logical chunk of instruction j
    uint32_t xbit[2];       // mask bit of the chunk's tap (0: column beyond Ktot, never valid)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int mrow = 2 * (wave + 8 * j) + xrr;                      // stage-local row
        const int lchunk = (((xpc >> 1) ^ w2_swz(mrow)) << 1) | (xpc & 1);
        const int k = k0 + lchunk * 8;
        if (k < p.Ktot) {
            const int tap = k / p.C, ci = k - tap * p.C;
            xbit[j] = 1u << tap;
            xcol[j] = (int64_t)p.dlin[tap] * p.ldx + ci;
        } else {
            xbit[j] = 0u;
            xcol[j] = 0;
        }
    }
    // ---- dY loader: instruction wave (+ 8 only when YI > 8: never) copies stage rows YRPI*wave + lane / YCH
    const int yrr = lane / YCH, ypc = lane % YCH;
    const bool ywave = wave < YI;
    const int ymrow = YRPI * wave + yrr;
    int ylchunk;
    if constexpr (BMW == 128) ylchunk = (((ypc >> 1) ^ w2_swz(ymrow)) << 1) | (ypc & 1);
    else ylchunk = (((ypc >> 1) ^ (((ymrow >> 1) & 1) | (((ymrow >> 3) & 1) << 1))) << 1) | (ypc & 1);
    const int yco = c0 + ylchunk * 8;
    const bool yok = yco < p.Co;

    // row-table entries of the two X rows of each of this wave's copy instructions: wave-uniform -> scalar loads
    auto tab_rows = [&](int step, i32x2 (&e)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int m = r0 + step * ROWS + 2 * (wave + 8 * j) + r;
                i32x2 v;
                v.x = 0; v.y = 0;
                if (m < r1) {
                    v.x = SF_SCALAR_PTR(i32x2, p.rowtab)[m].x;
                    v.y = SF_SCALAR_PTR(i32x2, p.rowtab)[m].y;
                }
                e[j][r] = v;
            }
    };
    auto issue = [&](int step, int buf, const i32x2 (&e)[2][2]) {
        f16* Ys = smem + buf * STAGE;
        f16* Xs = Ys + Y_ELEMS;
        if (ywave) {
            const int m = r0 + step * ROWS + ymrow;
            const f16* g = (m < r1 && yok) ? p.dy + (int64_t)m * p.ldy + yco : zline;
            SF_GLOBAL_LOAD_LDS16_ASM(g, Ys + wave * 512);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pos = xrr ? e[j][1].x : e[j][0].x;
            const uint32_t mk = (uint32_t)(xrr ? e[j][1].y : e[j][0].y);
            const bool ok = (mk & xbit[j]) != 0u;
            const f16* g = ok ? p.x + ((int64_t)pos * p.ldx + xcol[j]) : zline;
            SF_GLOBAL_LOAD_LDS16_ASM(g, Xs + (wave + 8 * j) * 512);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int pl = lane & 15, g4 = lane >> 4;
    auto compute = [&](int buf) {
        const f16* Ys = smem + buf * STAGE;
        const f16* Xs = Ys + Y_ELEMS;
        f16x8 af[TM], bf[TN];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = 8 * g4 + 4 * h + (pl >> 2);                  // stage row this lane addresses
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int col = wc * 64 + i * 16;                       // first channel of the 16-wide fragment
                int off;
                if constexpr (BMW == 128) off = m * BMW + ((((col >> 4) ^ w2_swz(m)) << 4) | (4 * (pl & 3)));
                else off = m * BMW + ((((col >> 4) ^ (((m >> 1) & 1) | (((m >> 3) & 1) << 1))) << 4) | (4 * (pl & 3)));
                const f16x4 t = as_f16x4(SF_LDS_TR16(Ys + off));
                af[i][4 * h + 0] = t[0]; af[i][4 * h + 1] = t[1]; af[i][4 * h + 2] = t[2]; af[i][4 * h + 3] = t[3];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wk * WN + j * 16;
                const int off = m * BKW + ((((col >> 4) ^ w2_swz(m)) << 4) | (4 * (pl & 3)));
                const f16x4 t = as_f16x4(SF_LDS_TR16(Xs + off));
                bf[j][4 * h + 0] = t[0]; bf[j][4 * h + 1] = t[1]; bf[j][4 * h + 2] = t[2]; bf[j][4 * h + 3] = t[3];
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = SF_MFMA16(af[i], bf[j], acc[i][j]);
    };

    {
        constexpr int COPIES = 2;           // X copies per wave and stage; waves 0 .. YI-1 carry one dY copy more and wait for it too
        i32x2 e[2][2];
        int issued = 0;
        for (; issued < NST - 1 && issued < nsteps; ++issued) { tab_rows(issued, e); issue(issued, issued, e); }
        if (issued < nsteps) tab_rows(issued, e);                       // entries of the next step to issue
        int cur = 0, nxt = NST - 1;
        for (int ks = 0; ks < nsteps_wg; ++ks) {
            // stages ks .. ks + NST - 2 are in flight (fewer at the tail): stage ks must have landed
            if (ks + NST - 2 < nsteps) {
                SF_WAIT_VMEM_N(COPIES);                                 // (waves with a dY copy also wait for it: one early)
            } else SF_WAIT_VMEM();
            SF_BARRIER_KEEP_VMEM();
            if (issued < nsteps) {
                issue(issued, nxt, e);
                ++issued;
                if (issued < nsteps) tab_rows(issued, e);               // scalar loads, consumed one step later
            }
            if (!DUAL || ks < nsteps) compute(cur);
            cur = cur == NST - 1 ? 0 : cur + 1;
            nxt = nxt == NST - 1 ? 0 : nxt + 1;
        }
    }

    // DUAL: half h keeps the fragment rows i in [h * TM / 2, (h + 1) * TM / 2) of the tile and hands the others over
    constexpr int I0 = 0, IH = DUAL ? TM / 2 : TM;
    if constexpr (DUAL) {
        static_assert(TM % 2 == 0, "the halves split the fragment rows");
        static_assert(2 * NST * STAGE * 2 >= 2 * IH * TN * 4 * 512 * 4, "the exchange reuses both rings");
        __syncthreads();                                                // every ring is dead
        float* xch = reinterpret_cast<float*>(smem_all);
        const int t512 = tid & 511;
        float* mine = xch + half * (IH * TN * 4 * 512);                 // what this half SENDS
        const float* theirs = xch + (1 - half) * (IH * TN * 4 * 512);
#pragma unroll
        for (int ii = 0; ii < IH; ++ii)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x4& a0 = acc[ii][j];                       // half 1 sends rows [0, IH)
                    const f32x4& a1 = acc[IH + ii][j];                  // half 0 sends rows [IH, TM)
                    mine[((ii * TN + j) * 4 + r) * 512 + t512] = half ? a0[r] : a1[r];
                }
        __syncthreads();
#pragma unroll
        for (int ii = 0; ii < IH; ++ii)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float o = theirs[((ii * TN + j) * 4 + r) * 512 + t512];
                    if (half) acc[IH + ii][j][r] += o; else acc[ii][j][r] += o;
                }
    }

    // every split (pair) owns its slab: plain stores, also when it had no rows to reduce (zeros)
    float* slab = p.ws + (int64_t)bz * p.Co_pad * p.Kpad;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int kcol = k0 + wk * WN + j * 16 + pl;
#pragma unroll
        for (int ii = I0; ii < IH; ++ii)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (DUAL) {
                    const int co0 = c0 + wc * 64 + ii * 16 + 4 * g4 + r;
                    const float v0 = acc[ii][j][r], v1 = acc[IH + ii][j][r];
                    slab[(int64_t)(co0 + (half ? IH * 16 : 0)) * p.Kpad + kcol] = half ? v1 : v0;
                } else {
                    const int co = c0 + wc * 64 + ii * 16 + 4 * g4 + r;
                    slab[(int64_t)co * p.Kpad + kcol] = acc[ii][j][r];
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Thin weight gradient: at most 32 output channels (the Fast pathway: 8 .. 32 wide bottlenecks over 0.8 .. 3.2 M positions).
// dW is tiny (16|32 x K), the layer is a pure stream over dY and X -- what bounds it is the bytes a CU keeps in flight, and
// the 128 x 16 register-staged tiles of sf_wgrad_kernel spend their life in per-workgroup prologues (row decode by division,
// one staged step in flight): 0.8 - 1.9 TB/s measured.  Here:
//   * ONE workgroup tile covers all output channels (BMW = 16 | 32) and 128 (or 32, for K <= 32) columns of K; the grid is
//     k-tiles x position splits;
//   * a stage is 128 positions; the FOUR WAVES SPLIT THE POSITIONS (32 each) and every wave accumulates the whole tile, so
//     each LDS byte is read exactly once; the four partial tiles are summed through LDS at the end;
//   * operands travel global -> LDS directly as in sf_wgrad2_kernel (row table through scalar loads, zero line for padding
//     taps / channels / rows beyond the split), two stages of 37 - 41 KB (two workgroups per CU) or three of 12 - 16 KB;
//   * bank conflicts of the transpose reads are avoided on the source side: 256-byte rows use the pair swizzle of
//     sf_wgrad2_kernel, 64-byte rows flip their 32-byte pair with bit 3 of the row, 32-byte rows (BMW = 16) swap bits 2 and 3
//     of the row position inside a copy block.
__device__ __forceinline__ int w2t_swap23(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }
// element offset of the 16-column fragment `tile` of logical stage row m in a [rows][W] operand image
template <int W>
__device__ __forceinline__ int w2t_frag_off(int m, int tile) {
    if constexpr (W == 128) return m * 128 + ((tile ^ w2_swz(m)) << 4);
    else if constexpr (W == 32) return m * 32 + ((tile ^ ((m >> 3) & 1)) << 4);
    else return w2t_swap23(m) * 16;
}
// loader side: the lane-linear slot (physical row position `prow` inside the stage, physical 16-byte chunk `pc`) holds
// logical row / logical chunk:
template <int W>
__device__ __forceinline__ void w2t_slot(int prow, int pc, int& row, int& chunk) {
    if constexpr (W == 128) { row = prow; chunk = (((pc >> 1) ^ w2_swz(prow)) << 1) | (pc & 1); }
    else if constexpr (W == 32) { row = prow; chunk = (((pc >> 1) ^ ((prow >> 3) & 1)) << 1) | (pc & 1); }
    else { row = w2t_swap23(prow); chunk = pc; }
}

template <int BMW, int BKW, int NST>
__global__ __launch_bounds__(256) void sf_wgrad2t_kernel(Wgrad2Params p) {
    constexpr int ROWS = 128, NW = 4;
    constexpr int TMC = BMW / 16, TNK = BKW / 16;
    constexpr int YCH = BMW / 8, XCH = BKW / 8;         // 16-byte chunks per row
    constexpr int YRPI = 64 / YCH, XRPI = 64 / XCH;     // rows per copy instruction (64 lanes x 16 B)
    constexpr int YPW = ROWS / YRPI / NW, XPW = ROWS / XRPI / NW;   // copy instructions per wave and stage
    constexpr int COPIES = YPW + XPW;
    constexpr int Y_ELEMS = ROWS * BMW, X_ELEMS = ROWS * BKW, STAGE = Y_ELEMS + X_ELEMS;
    static_assert(NST * STAGE * 2 >= NW * TMC * TNK * 256 * 4, "the cross-wave reduction reuses the stage buffers");
    __shared__ __attribute__((aligned(16))) f16 smem[NST * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wg = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = (int)(wg % (uint32_t)p.tiles_k);
    const int bz = (int)(wg / (uint32_t)p.tiles_k);
    const int k0 = bx * BKW;
    // first position of step s of this split: r0 + s * rstep
    const int rr = p.stage_stride > 0;
    const int r0 = rr ? bz * ROWS : bz * p.rows_per_split;
    const int rstep = rr ? p.stage_stride * ROWS : ROWS;
    int r1 = rr ? p.M : r0 + p.rows_per_split;
    if (r1 > p.M) r1 = p.M;
    const int nsteps = r1 > r0 ? (r1 - r0 + rstep - 1) / rstep : 0;
    const f16* const zline = reinterpret_cast<const f16*>(sf_zero_line);

    // ---- X loader: instruction j of the wave is copy block q = wave + NW * j of the stage (rows q * XRPI ...)
    int xrow[XPW];          // logical stage row of this lane's slot
    int xsel[XPW];          // its index inside the block's XRPI row-table entries
    int64_t xcol[XPW];      // channel offset + tap displacement (elements) of this lane's logical chunk
    uint32_t xbit[XPW];     // mask bit of the chunk's tap (0: column beyond Ktot)
#pragma unroll
    for (int j = 0; j < XPW; ++j) {
        const int q = wave + NW * j;
        int row, chunk;
        w2t_slot<BKW>(q * XRPI + lane / XCH, lane % XCH, row, chunk);
        xrow[j] = row;
        xsel[j] = row - q * XRPI;
        const int k = k0 + chunk * 8;
        if (k < p.Ktot) {
            const int tap = k / p.C, ci = k - tap * p.C;
            xbit[j] = 1u << tap;
            xcol[j] = (int64_t)p.dlin[tap] * p.ldx + ci;
        } else {
            xbit[j] = 0u;
            xcol[j] = 0;
        }
    }
    // ---- dY loader
    int yrow[YPW];
    int yco[YPW];
#pragma unroll
    for (int j = 0; j < YPW; ++j) {
        const int q = wave + NW * j;
        int row, chunk;
        w2t_slot<BMW>(q * YRPI + lane / YCH, lane % YCH, row, chunk);
        yrow[j] = row;
        yco[j] = chunk * 8;
    }

    // row-table entries of the XRPI consecutive rows of each of this wave's X copy blocks: wave-uniform -> WIDE scalar loads
    // (32 bytes = 4 entries each, all issued before the first use: one entry per load with its use right behind it made the
    // compiler wait for every load in turn, ~4 us of serial scalar-cache latency per stage).  Entries of positions beyond the
    // table (the plan pads it by one stage) or beyond this split are garbage and masked by the position test.
    auto tab_rows = [&](int step, int (&epos)[XPW], uint32_t (&emk)[XPW]) {
        i32x8 t[XPW][XRPI / 4];
#pragma unroll
        for (int j = 0; j < XPW; ++j) {
            const int m0 = r0 + step * rstep + (wave + NW * j) * XRPI;
#pragma unroll
            for (int r = 0; r < XRPI / 4; ++r) t[j][r] = SF_SCALAR_PTR(i32x8, p.rowtab + m0)[r];
        }
#pragma unroll
        for (int j = 0; j < XPW; ++j) {
            const int m0 = r0 + step * rstep + (wave + NW * j) * XRPI;
            int pos = 0;
            uint32_t mk = 0u;
#pragma unroll
            for (int r = 0; r < XRPI; ++r)
                if (xsel[j] == r) { pos = t[j][r / 4][2 * (r % 4)]; mk = (uint32_t)t[j][r / 4][2 * (r % 4) + 1]; }
            epos[j] = pos;
            emk[j] = (m0 + xsel[j] < r1) ? mk : 0u;
        }
    };
    auto issue = [&](int step, int buf, const int (&epos)[XPW], const uint32_t (&emk)[XPW]) {
        f16* Ys = smem + buf * STAGE;
        f16* Xs = Ys + Y_ELEMS;
#pragma unroll
        for (int j = 0; j < YPW; ++j) {
            const int m = r0 + step * rstep + yrow[j];
            const f16* g = (m < r1 && yco[j] < p.Co) ? p.dy + (int64_t)m * p.ldy + yco[j] : zline;
            SF_GLOBAL_LOAD_LDS16_ASM(g, Ys + (wave + NW * j) * 512);
        }
#pragma unroll
        for (int j = 0; j < XPW; ++j) {
            const bool ok = (emk[j] & xbit[j]) != 0u;
            const f16* g = ok ? p.x + ((int64_t)epos[j] * p.ldx + xcol[j]) : zline;
            SF_GLOBAL_LOAD_LDS16_ASM(g, Xs + (wave + NW * j) * 512);
        }
    };

    f32x4 acc[TMC][TNK];
#pragma unroll
    for (int i = 0; i < TMC; ++i)
#pragma unroll
        for (int j = 0; j < TNK; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int pl = lane & 15, g4 = lane >> 4;
    auto compute = [&](int buf) {
        const f16* Ys = smem + buf * STAGE;
        const f16* Xs = Ys + Y_ELEMS;
        f16x8 af[TMC], bf[TNK];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = wave * 32 + 8 * g4 + 4 * h + (pl >> 2);      // this wave's quarter of the stage
#pragma unroll
            for (int i = 0; i < TMC; ++i) {
                const f16x4 t = as_f16x4(SF_LDS_TR16(Ys + w2t_frag_off<BMW>(m, i) + 4 * (pl & 3)));
                af[i][4 * h + 0] = t[0]; af[i][4 * h + 1] = t[1]; af[i][4 * h + 2] = t[2]; af[i][4 * h + 3] = t[3];
            }
#pragma unroll
            for (int j = 0; j < TNK; ++j) {
                const f16x4 t = as_f16x4(SF_LDS_TR16(Xs + w2t_frag_off<BKW>(m, j) + 4 * (pl & 3)));
                bf[j][4 * h + 0] = t[0]; bf[j][4 * h + 1] = t[1]; bf[j][4 * h + 2] = t[2]; bf[j][4 * h + 3] = t[3];
            }
        }
#pragma unroll
        for (int i = 0; i < TMC; ++i)
#pragma unroll
            for (int j = 0; j < TNK; ++j)
                acc[i][j] = SF_MFMA16(af[i], bf[j], acc[i][j]);
    };

    {
        int epos[XPW];
        uint32_t emk[XPW];
        int issued = 0;
        for (; issued < NST - 1 && issued < nsteps; ++issued) { tab_rows(issued, epos, emk); issue(issued, issued, epos, emk); }
        if (issued < nsteps) tab_rows(issued, epos, emk);
        int cur = 0, nxt = NST - 1;
        for (int ks = 0; ks < nsteps; ++ks) {
            // stages ks .. ks + NST - 2 are in flight (fewer at the tail): stage ks must have landed
            if (NST > 2 && ks + NST - 2 < nsteps) SF_WAIT_VMEM_N((NST - 2) * COPIES);
            else SF_WAIT_VMEM();
            SF_BARRIER_KEEP_VMEM();
            if (issued < nsteps) {
                issue(issued, nxt, epos, emk);
                ++issued;
                if (issued < nsteps) tab_rows(issued, epos, emk);
            }
            compute(cur);
            cur = cur == NST - 1 ? 0 : cur + 1;
            nxt = nxt == NST - 1 ? 0 : nxt + 1;
        }
    }

    // ---- sum the four waves' partial tiles through LDS, store the split's slab (plain stores; zeros when it had no rows)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < TMC; ++i)
#pragma unroll
        for (int j = 0; j < TNK; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((wave * TMC + i) * TNK + j) * 4 + r) * 64 + lane] = acc[i][j][r];
    __syncthreads();
    float* slab = p.ws + (int64_t)bz * p.Co_pad * p.Kpad;
    constexpr int TILE_ELEMS = TMC * TNK * 256;
    for (int e = tid; e < TILE_ELEMS; e += 256) {
        const float v = (red[e] + red[TILE_ELEMS + e]) + (red[2 * TILE_ELEMS + e] + red[3 * TILE_ELEMS + e]);
        const int ln = e & 63, r = (e >> 6) & 3, t = e >> 8;
        const int i = t / TNK, j = t - i * TNK;
        const int co = i * 16 + 4 * (ln >> 4) + r;
        const int kcol = k0 + j * 16 + (ln & 15);
        slab[(int64_t)co * p.Kpad + kcol] = v;
    }
}
