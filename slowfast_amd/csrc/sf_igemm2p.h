// Persistent implicit GEMM with drain waves (gfx950): sf_igemm2's K loop, one workgroup per CU walking a list of tiles, the
// output tile leaving through EIGHT EXTRA WAVES while the eight compute waves are already in the next tile's K loop.
//
// Same contraction and Igemm2Params as sf_igemm2.h (reference call sites: nn.Linear of slowfast/models/attention.py:193-195,
// 354-392 and common.py:7-34 -- qkv / proj / fc1 / fc2 and their data gradients --, the pointwise and gathered nn.Conv3d of
// resnet_helper.py:331-369).  Why (profiles/r5_v23_gemm_k384_ablation.txt, HBM-cold, fc1 of MViTv2-S stage 3, M = 50208, K = 384,
// N = 1536): the launch-per-tile kernel takes 120 us, 75 without its epilogue, 60 with either the copy stream or the MFMA stream
// alone, 26 with neither -- a third of the time is the epilogue (154 MB of stores issued by waves that do nothing else meanwhile),
// a fifth is per-tile skeleton (dispatch, row decode, pipeline fill), and shallow K (12 steps) gives the K loop nothing to hide
// them behind.  Here:
//   * ONE workgroup per CU (grid = min(tiles, CUs)), XCD-contiguous tile lists: no per-tile dispatch, the NEXT tile's first two
//     operand stages are issued before the current tile's accumulators are even converted (pipeline fill behind the epilogue);
//   * the epilogue is split: the compute waves only scale / bias their accumulators, take the BatchNorm statistics and write the
//     fp16 tile into a STAGING image in LDS of its own (69 KB beside the 72 KB of operand stages: 144 KB, one workgroup per CU);
//   * waves 8 .. 15 ("drain waves") turn the staging image into 16-byte row-contiguous global stores -- residual, GELU / GELU'
//     epilogues, bias-gradient column sums included -- slice by slice, one slice per K step of the NEXT tile.  They have their own
//     vmcnt: the stores never enter the counted waits of the copy pipeline (a store issued by a compute wave inside the K loop
//     would: loads and stores share one counter and complete out of order with respect to each other);
//   * every wave executes the same sequence of workgroup barriers: per tile one per K step, B1 (operand stages and the staging
//     image of the previous tile are free), B2 (the staging image of this tile is complete).
#pragma once
#include "sf_igemm2.h"

// what the drain waves keep of a tile between its B2 and the end of its drain
struct I2pTile { int mt, nt; };

template <int BN, bool F32R = false>
__global__ __launch_bounds__(1024, 4) void sf_igemm2p_kernel(Igemm2Params p, int ntiles) {
    constexpr int BM = 256, WAVES_M = 4, WAVES_N = 2, BK = 32, NST = 3;
    constexpr int NW = 8, NT = 512, ND = 512, NDW = ND / 64;       // compute waves / threads, drain threads / waves
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 16, TN = WN / 16;
    constexpr int KSL = BK / 8, RPI = 64 / KSL;
    constexpr int NA = BM / RPI / NW, NBI = BN / RPI, NB = (NBI + NW - 1) / NW;
    static_assert(NBI % NW == 0 || NBI < NW, "B tile: uniform copy count per wave or one partial round");
    constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK, STAGE = A_ELEMS + B_ELEMS;
    constexpr int STG_LD = BN + 8;
    constexpr int HALVES = BM / 128, WPH = WAVES_M / HALVES;
    constexpr int CG = BN / 8;                                      // 16-byte column groups of a tile row
    constexpr int PIECES = BM * CG / ND;                            // 16-byte pieces a drain thread stores per tile
    static_assert(ND % CG == 0 && 64 % CG == 0, "a drain thread keeps one column group");
    constexpr int OFF_STG = NST * STAGE * 2, OFF_RED = OFF_STG + BM * STG_LD * 2, OFF_DRN = OFF_RED + WAVES_M * 2 * BN * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[OFF_DRN + NDW * CG * 16 * 4];
    f16* const smem = reinterpret_cast<f16*>(lds_raw);
    f16* const stg = reinterpret_cast<f16*>(lds_raw + OFF_STG);
    float (*const s_red)[2][BN] = reinterpret_cast<float (*)[2][BN]>(lds_raw + OFF_RED);
    float* const s_drn = reinterpret_cast<float*>(lds_raw + OFF_DRN);       // [drain waves][CG][16]: their column sums

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-contiguous tile lists: workgroup b runs on XCD b % 8 and takes every `slots`-th tile of that XCD's range
    // (fewer than 8 workgroups -- test grids -- : as many ranges as workgroups)
    const int nx = (int)gridDim.x < 8 ? (int)gridDim.x : 8;
    const int xcd = (int)blockIdx.x % nx, slot = (int)blockIdx.x / nx, slots = ((int)gridDim.x + nx - 1 - xcd) / nx;
    const int tiles_per_xcd = (ntiles + nx - 1) / nx;
    const int t_begin = xcd * tiles_per_xcd + slot;
    int t_end = (xcd + 1) * tiles_per_xcd;
    if (t_end > ntiles) t_end = ntiles;
    const int csteps = p.C / BK, ksteps = p.ntaps * csteps;

    if (wave >= NW) {
        // ================================================================ drain waves
        const int td = tid - NT;                                    // 0 .. ND - 1
        const int ecg = td % CG;
        const bool colsum = p.bnb_part != nullptr;                  // plain column sums of the stored tile (bias gradient)
        float bsg[8];
        auto drain = [&](const I2pTile& t, int k0, int k1) {        // pieces k0 .. k1 - 1 of this thread
            const int m0 = t.mt * BM, n0 = t.nt * BN, ecol = n0 + ecg * 8;
            for (int k = k0; k < k1; ++k) {
                const int row = (td + k * ND) / CG, m = m0 + row;
                if (m >= p.M || ecol >= p.Nout) continue;
                f16x8 v = ld16(stg + row * STG_LD + ecg * 8);
                bool rok = p.resid && m >= p.resid_row0;
                if constexpr (F32R) {
                    uint32_t srow;
                    if (f32_row(p.f32, m, srow)) rok = false;       // residual already inside the staged value
                }
                if (rok) {
                    const f16x8 r = ld16(p.resid + (int64_t)m * p.ldr + ecol);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (f16)((float)v[e] + (float)r[e]);
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
                if (colsum) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsg[e] += (float)v[e];
                }
            }
        };
        // column sums of a drained tile: lanes that share a column group fold by shuffles, the two drain waves leave their sums in
        // s_drn; after the next workgroup barrier thread (column group, element) adds the two and writes row mt of the partial
        // table (fixed order: deterministic)
        auto colsum_publish = [&]() {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                for (int mask = CG; mask < 64; mask <<= 1) bsg[e] += __shfl_xor(bsg[e], mask);
            if (lane < CG) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s_drn[((td >> 6) * CG + lane) * 16 + e] = bsg[e];
            }
        };
        auto colsum_store = [&](const I2pTile& t) {
            if (td < CG * 8) {
                const int cgi = td >> 3, e = td & 7, col = t.nt * BN + cgi * 8 + e;
                if (col < p.Nout) {
                    float* prow = p.bnb_part + (int64_t)t.mt * 2 * p.Nout;
                    float sum = 0.f;
#pragma unroll
                    for (int w = 0; w < NDW; ++w) sum += s_drn[(w * CG + cgi) * 16 + e];
                    prow[col] = sum;
                    prow[p.Nout + col] = 0.f;
                }
            }
        };
        const int kdiv = ksteps > 0 ? ksteps : 1;
        const int pps = (PIECES + kdiv - 1) / kdiv;                 // pieces of the previous tile stored per K step
        I2pTile prev = {-1, -1};
        for (int tile = t_begin; tile < t_end; tile += slots) {
            const bool have = prev.mt >= 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) bsg[e] = 0.f;
            for (int ks = 0; ks < ksteps; ++ks) {
                if (have && ks * pps < PIECES) drain(prev, ks * pps, ks * pps + pps < PIECES ? ks * pps + pps : PIECES);
                SF_BARRIER_KEEP_VMEM();
            }
            if (have && ksteps * pps < PIECES) drain(prev, ksteps * pps, PIECES);
            if (have && colsum) colsum_publish();
            SF_BARRIER_KEEP_VMEM();                                 // B1
            if (have && colsum) colsum_store(prev);
            SF_BARRIER_KEEP_VMEM();                                 // B2: the staging image of `tile` is complete
            prev.mt = tile / p.ntiles_n;
            prev.nt = tile % p.ntiles_n;
        }
        if (prev.mt >= 0) {                                         // the last tile of the list
#pragma unroll
            for (int e = 0; e < 8; ++e) bsg[e] = 0.f;
            drain(prev, 0, PIECES);
            if (colsum) colsum_publish();
        }
        SF_BARRIER_KEEP_VMEM();                                     // B3 (the compute waves wait here, then leave)
        if (prev.mt >= 0 && colsum) colsum_store(prev);
        return;
    }

    // ==================================================================== compute waves
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lrow = lane / KSL;
    const int kslot = ((lane & 3) - 2 * ((lrow >> 2) & 3)) & 3;     // inverse of lds_tile_off()'s rotation
    int64_t aoff[NA];
    uint32_t amask[NA];
    const f16* bptr[NB];
    auto decode = [&](int tile) {
        const int nt = tile % p.ntiles_n, mt = tile / p.ntiles_n;
        const int m0 = mt * BM, n0 = nt * BN;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            int m = m0 + (wave + NW * j) * RPI + lrow;
            if (m >= p.M) m = p.M - 1;                              // clamped rows are computed and never stored
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
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int co = n0 + (wave + NW * j) * RPI + lrow;
            if (co >= p.Nout) co = p.Nout - 1;
            bptr[j] = p.wmat + (int64_t)co * p.ldw + kslot * 8;
        }
    };
    const f16* const zline = reinterpret_cast<const f16*>(sf_zero_line);
    auto issue = [&](int tap, int c0, int buf) {
        f16* As = smem + buf * STAGE;
        f16* Bs = As + A_ELEMS;
        const int64_t dsrc = (int64_t)p.taps[tap].dlin * p.ld + c0;
        const int wk = p.taps[tap].wcol + c0;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const f16* g = ((amask[j] >> tap) & 1u) ? p.src + (aoff[j] + dsrc) : zline;
            SF_GLOBAL_LOAD_LDS16_ASM(g, As + (wave + NW * j) * 512);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (NBI % NW == 0 || wave + NW * j < NBI) SF_GLOBAL_LOAD_LDS16_ASM(bptr[j] + wk, Bs + (wave + NW * j) * 512);
    };
    constexpr int COPIES = NA + (NBI % NW == 0 ? NB : NB - 1);      // fewest copies a wave issues per stage
    int tap_i = 0, c_i = 0, issued = 0;
    auto advance = [&]() { if (++tap_i == p.ntaps) { tap_i = 0; c_i += BK; } };
    auto prologue = [&](int tile) {                                 // loader state of `tile`, its first NST - 1 stages
        decode(tile);
        tap_i = 0; c_i = 0;
        for (issued = 0; issued < NST - 1 && issued < ksteps; ++issued) { issue(tap_i, c_i, issued); advance(); }
    };

    f32x4 acc[TM][TN];
    if (t_begin < t_end) prologue(t_begin);
    for (int tile = t_begin; tile < t_end; tile += slots) {
        const int nt = tile % p.ntiles_n, mt = tile / p.ntiles_n;
        const int m0 = mt * BM, n0 = nt * BN;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // ---- K loop (sf_igemm2.h): stages ks + 1, ks + 2 in flight while stage ks is multiplied
        int cur = 0, nxt = NST - 1;
        for (int ks = 0; ks < ksteps; ++ks) {
            if (ks + 1 < ksteps) SF_WAIT_VMEM_N(COPIES);
            else SF_WAIT_VMEM();
            SF_BARRIER_KEEP_VMEM();
            if (issued < ksteps) { issue(tap_i, c_i, nxt); advance(); ++issued; }
            const f16* As = smem + cur * STAGE;
            const f16* Bs = As + A_ELEMS;
            f16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = ld16(As + i2_lds_off<BK>(wm * WM + i * 16 + (lane & 15), lane >> 4));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = ld16(Bs + i2_lds_off<BK>(wn * WN + j * 16 + (lane & 15), lane >> 4));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = SF_MFMA16(af[i], bf[j], acc[i][j]);
            cur = cur == NST - 1 ? 0 : cur + 1;
            nxt = nxt == NST - 1 ? 0 : nxt + 1;
        }
        SF_BARRIER_KEEP_VMEM();                                     // B1: stages free; the drain waves are done with `stg`
        // ---- compute-side epilogue: scale, bias, fp32 side rows (everything that touches global memory FIRST: an ordinary load
        // issued behind the next tile's inline-asm copies would make hipcc drain them at its first use), then the next tile's
        // pipeline fill, then BatchNorm statistics and the fp16 tile into the staging image
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
        if (tile + slots < t_end) prologue(tile + slots);           // the next tile's pipeline fill runs behind the rest
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
        SF_BARRIER_KEEP_VMEM();                                     // B2: the staging image is complete
        if (p.stat_part && tid < HALVES * BN) {                     // (s_red is rewritten after the next tile's B1 at the earliest)
            const int half = tid / BN, c = tid % BN;
            const int col = n0 + c;
            const int prow = mt * HALVES + half;
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
    }
    SF_BARRIER_KEEP_VMEM();                                         // B3: the drain waves have stored the last tile's sums
}
