// Implicit-GEMM convolution, third generation (gfx950): 256 x 256 x 64 tiles on the eight-phase ping-pong schedule.
//
// Same contraction and the same Igemm2Params / epilogue as sf_igemm2.h (reference call sites: slowfast/models/resnet_helper.py:
// 331-369 BottleneckTransform a/b/c, nn.Linear of attention.py:193-195 / common.py:7-34).  What changes is the K loop
// (cdna_hip_programming.md "The 256^2 8-phase template", MI355X_MICROARCH.md "Two waves per SIMD"):
//   * 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 32 accumulator blocks (128 registers): one ds_read_b128 feeds 2.7 MFMAs
//     instead of 2 (64 x 64 wave tiles), one copied operand byte 128 flop instead of 85;
//   * a K tile (64 deep) is worked off in FOUR PHASES, one 64 x 32 quadrant of the wave tile each (16 MFMAs): the phase's
//     fragment reads and ONE half-tile of copies (2 global_load_lds_dwordx4 per wave) first, barrier, the MFMAs, barrier;
//   * the two wave rows run ONE BARRIER APART: while waves 0-3 (one per SIMD) multiply, waves 4-7 (their SIMD partners) read
//     fragments and issue copies, and vice versa -- the matrix pipe of a SIMD always has one wave in its MFMA cluster
//     (s_setprio 1 around it) instead of all eight waves standing in copy issue together (sf_igemm2's measured loss,
//     profiles/r3/r3_v6_igemm2_ablation.md: copy stream and MFMA stream each 75-80 % of the kernel, imperfectly overlapped);
//   * copies run up to six phases ahead of their use: two tile buffers of four half-tiles (A rows 0-127 / 128-255, B rows
//     0-127 / 128-255; 16 KB each, 128 KB in all), a half-tile slot is re-filled in the phase after its last fragment read;
//     ONE counted s_waitcnt vmcnt per K tile, never zero inside the loop.
//
// Schedule of K tile t (buffer b = t & 1; wave (wr, wc) reads A half wr and B half wc >> 1 of the buffer):
//   phase 1  read a0 (rows 0-63 of the half: 8 x b128) + b0 (32 columns: 4 x b128)   copy A0(t+1) -> buffer b^1    MFMA a0 x b0
//   phase 2  read b1 (4 x b128)                                                        copy A1(t+1) -> buffer b^1    MFMA a0 x b1
//   phase 3  read a1 (8 x b128)                                                        copy B0(t+2) -> buffer b      MFMA a1 x b1
//   phase 4  (b0 is still in registers)   copy B1(t+2) -> buffer b, s_waitcnt vmcnt(4): tile t+1 has landed         MFMA a1 x b0
// Hazards (wave rows one barrier apart; interval = the stretch between two barriers):
//   WAR  a slot is copied into in the phase AFTER its last read (B(t): read in phases 1-2, re-filled in phases 3-4; A(t): read
//        in phases 1 and 3, re-filled in phases 1-2 of tile t+1): the reading row finished its reads (lgkmcnt(0)) before the
//        barrier that opens the copying row's interval;
//   RAW  the wait for tile t+1 sits at the END of phase 4's load interval, i.e. BEFORE that interval's barrier in both rows: the
//        late row's barrier is the one the early row passes before its phase-1 reads of tile t+1.
#pragma once
#include "sf_igemm2.h"

template <bool F32R = false>
__global__ __launch_bounds__(512, 2) void sf_igemm3_kernel(Igemm2Params p) {
    constexpr int BM = 256, BN = 256, BK = 64, WAVES_M = 2, WAVES_N = 4;
    constexpr int TM = 8, TN = 4;
    constexpr int HALF = 128 * BK;                  // elements of a half-tile (16 KB)
    constexpr int BUF = 4 * HALF;                   // A0 A1 B0 B1
    constexpr int SMEM_MAIN = 2 * BUF;
    constexpr int STG_LD = BN + 8, SMEM_STG = BM * STG_LD;
    constexpr int SMEM = SMEM_MAIN > SMEM_STG ? SMEM_MAIN : SMEM_STG;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[SMEM * 2 + WAVES_M * 2 * BN * 4 + BM * 4];
    f16* const smem = reinterpret_cast<f16*>(lds_raw);
    float (*const s_red)[2][BN] = reinterpret_cast<float (*)[2][BN]>(lds_raw + SMEM * 2);
    int* const s_orow = reinterpret_cast<int*>(lds_raw + SMEM * 2 + WAVES_M * 2 * BN * 4);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tile = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int nt = tile % p.ntiles_n, mt = tile / p.ntiles_n;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- loader state.  Copy instruction j (0 / 1) of this wave fills rows (wave + 8 j) * 8 .. + 7 of a half-tile; lane -> row
    // lane >> 3, physical 16-byte slot lane & 7 holding the logical K slot (lane & 7) ^ (row & 7) (source-side swizzle of
    // i2_lds_off<64>).  q = half * 2 + j.
    const int lrow = lane >> 3;
    const int kslot = (lane & 7) ^ lrow;
    int64_t aoff[4];
    uint32_t amask[4];
    const f16* bptr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int hrow = (q >> 1) * 128 + (wave + 8 * (q & 1)) * 8 + lrow;
        int m = m0 + hrow;
        if (m >= p.M) m = p.M - 1;                                  // clamped rows are computed and never stored
        uint32_t qq, a, b, c, n;
        fd_divmod((uint32_t)m, p.fdrW, qq, c);
        fd_divmod(qq, p.fdrH, qq, b);
        fd_divmod(qq, p.fdrT, n, a);
        const int bt = (int)a * p.mulT + p.offT, bh = (int)b * p.mulH + p.offH, bw = (int)c * p.mulW + p.offW;
        aoff[q] = ((((int64_t)n * p.sT + bt) * p.sH + bh) * p.sW + bw) * (int64_t)p.ld + kslot * 8;
        uint32_t mk = 0;
        for (int t = 0; t < p.ntaps; ++t) {
            const int st = bt + p.dt[t], sh = bh + p.dh[t], sw = bw + p.dw[t];
            const bool ok = (unsigned)st < (unsigned)p.sT && (unsigned)sh < (unsigned)p.sH && (unsigned)sw < (unsigned)p.sW;
            mk |= (ok ? 1u : 0u) << t;
        }
        amask[q] = mk;
        int co = n0 + hrow;
        if (co >= p.Nout) co = p.Nout - 1;
        bptr[q] = p.wmat + (int64_t)co * p.ldw + kslot * 8;
    }
    if (p.omap) {
        for (int r = tid; r < BM; r += 512) {
            int m = m0 + r;
            if (m >= p.M) m = p.M - 1;
            uint32_t qq, a, b, c, n;
            fd_divmod((uint32_t)m, p.fdrW, qq, c);
            fd_divmod(qq, p.fdrH, qq, b);
            fd_divmod(qq, p.fdrT, n, a);
            s_orow[r] = (((int)n * p.oT + (int)a * p.omT + p.ooT) * p.oH + (int)b * p.omH + p.ooH) * p.oW + (int)c * p.omW + p.ooW;
        }
    }
    const f16* const zline = reinterpret_cast<const f16*>(sf_zero_line);
    const int ktiles = p.ntaps * (p.C / BK);        // K order: channel chunk outer, tap inner (sf_igemm2.h)

    // the two copy streams run at different distances: A one tile ahead, B two tiles ahead
    int a_tap = 0, a_c0 = 0, b_tap = 0, b_c0 = 0;
    auto adv = [&](int& tap, int& c0) { if (++tap == p.ntaps) { tap = 0; c0 += BK; } };
    auto copy_a = [&](int half, int buf) {          // A half `half` of the tile (a_tap, a_c0)
        f16* dst = smem + buf * BUF + half * HALF;
        const int64_t dsrc = (int64_t)p.taps[a_tap].dlin * p.ld + a_c0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = half * 2 + j;
            const f16* g = ((amask[q] >> a_tap) & 1u) ? p.src + (aoff[q] + dsrc) : zline;
            SF_GLOBAL_LOAD_LDS16_ASM(g, dst + (wave + 8 * j) * 512);
        }
    };
    auto copy_b = [&](int half, int buf) {          // B half `half` of the tile (b_tap, b_c0)
        f16* dst = smem + buf * BUF + (2 + half) * HALF;
        const int wk = p.taps[b_tap].wcol + b_c0;
#pragma unroll
        for (int j = 0; j < 2; ++j) SF_GLOBAL_LOAD_LDS16_ASM(bptr[half * 2 + j] + wk, dst + (wave + 8 * j) * 512);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 areg[2][4], b0reg[2][2], b1reg[2][2];     // [K half][16-row block]

    const int fr = lane & 15, fk = lane >> 4;
    auto read_a = [&](const f16* Ah, int sub) {     // rows sub * 64 .. + 63 of the wave's A half
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i) areg[kk][i] = ld16(Ah + i2_lds_off<64>(sub * 64 + i * 16 + fr, kk * 4 + fk));
    };
    auto read_b = [&](const f16* Bh, int sub, f16x8 (&breg)[2][2]) {     // rows (wc & 1) * 64 + sub * 32 .. + 31 of the B half
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) breg[kk][j] = ld16(Bh + i2_lds_off<64>((wc & 1) * 64 + sub * 32 + j * 16 + fr, kk * 4 + fk));
    };
#define SF_I3_MFMA(IB, JB, BREG)                                                                   \
    do {                                                                                           \
        __builtin_amdgcn_s_setprio(1);                                                             \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                           \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                          \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                      \
                    acc[(IB) + i][(JB) + j] = SF_MFMA16(areg[kk][i], BREG[kk][j], acc[(IB) + i][(JB) + j]); \
        __builtin_amdgcn_s_setprio(0);                                                             \
    } while (0)

    // ---- prologue: tile 0 entirely, the B halves of tile 1
    copy_a(0, 0); copy_a(1, 0); adv(a_tap, a_c0);
    copy_b(0, 0); copy_b(1, 0); adv(b_tap, b_c0);
    if (ktiles > 1) {
        copy_b(0, 1); copy_b(1, 1); adv(b_tap, b_c0);
        SF_WAIT_VMEM_N(4);
    } else {
        SF_WAIT_VMEM();
    }
    SF_BARRIER_KEEP_VMEM();
    if (wr == 1) SF_BARRIER_KEEP_VMEM();            // the second wave row runs one barrier behind the first from here on

    for (int t = 0; t < ktiles; ++t) {
        const int b = t & 1;
        const f16* Ah = smem + b * BUF + wr * HALF;
        const f16* Bh = smem + b * BUF + (2 + (wc >> 1)) * HALF;
        const bool more1 = t + 1 < ktiles, more2 = t + 2 < ktiles;
        // ---- phase 1
        read_a(Ah, 0);
        read_b(Bh, 0, b0reg);
        if (more1) copy_a(0, b ^ 1);
        SF_BARRIER_KEEP_VMEM();
        SF_I3_MFMA(0, 0, b0reg);
        SF_BARRIER_KEEP_VMEM();
        // ---- phase 2
        read_b(Bh, 1, b1reg);
        if (more1) { copy_a(1, b ^ 1); adv(a_tap, a_c0); }
        SF_BARRIER_KEEP_VMEM();
        SF_I3_MFMA(0, 2, b1reg);
        SF_BARRIER_KEEP_VMEM();
        // ---- phase 3
        read_a(Ah, 1);
        if (more2) copy_b(0, b);
        SF_BARRIER_KEEP_VMEM();
        SF_I3_MFMA(4, 2, b1reg);
        SF_BARRIER_KEEP_VMEM();
        // ---- phase 4
        if (more2) { copy_b(1, b); adv(b_tap, b_c0); SF_WAIT_VMEM_N(4); }     // tile t + 1 has landed; B(t + 2) may be in flight
        else SF_WAIT_VMEM();
        SF_BARRIER_KEEP_VMEM();
        SF_I3_MFMA(4, 0, b0reg);
        SF_BARRIER_KEEP_VMEM();
    }
#undef SF_I3_MFMA
    if (wr == 0) SF_BARRIER_KEEP_VMEM();            // the first wave row waits for the second one's last interval
    __syncthreads();                                // the epilogue staging reuses the operand buffers

    i2_epilogue<BM, BN, WAVES_M, WAVES_N, F32R>(p, acc, smem, s_red, s_orow, mt, nt);
}
