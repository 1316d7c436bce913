// Thin implicit GEMM (gfx950): convolution forward / data gradient with at most 32 OUTPUT columns and a contraction of at most
// 128 (Fast pathway of SlowFast: 8 - 32 channel bottlenecks over 0.8 - 3.2 M positions; nn.Conv3d at resnet_helper.py:331-369
// and its backward w.r.t. the input).
//
//   y[m][n] = sum_k A(m, k) * W[n][k],   A(m, tap*C + c) = src[pos(m) + delta(tap)][c]  (zero where the tap leaves the source)
//
// These layers are pure streams: the weights are a few KB, every input byte is used once.  The 128 x 16 tiles of sf_igemm_kernel
// spend a workgroup's life in its prologue and epilogue (one tile = three 32-deep steps) and reach 1.7 - 4 TB/s.  Here a WAVE
// streams through many 32-position slices on its own:
//   * the weights live in REGISTERS for the whole kernel (MFMA B fragments, at most 32 VGPRs);
//   * each wave owns a private LDS ring of 32-position x KP slices filled by LDS-DMA (inline-asm form, see sf_common.h); nothing
//     is shared between waves, so the loop has NO barrier -- a wave waits only for its own counted vmcnt, and the waves of a CU
//     drift apart and cover one another's waits;
//   * positions come from the cached row table of the geometry (sf_wgrad2_rowtab_kernel), read with wide scalar loads;
//   * the 32 x BN result slice is staged through the wave's own LDS corner and stored as 16-byte row segments; BatchNorm partial
//     sums are kept in registers across all slices and leave as ONE row per workgroup.
// Slices are dealt round robin (128-position stages over workgroups, four slices of a stage over the four waves).
#pragma once
#include "sf_common.h"
#include "sf_wgrad2.h"

struct Igemm2tParams {
    const f16* src; int ld, C;          // gathered operand: rows of C channels, pitch ld
    int M;                              // output positions
    int Ktot;                           // taps * C
    const i32x2* rowtab;                // [M (+ pad)] {linear source position of the row's base coordinate, tap mask}
    int32_t dlin[SF_I2_MAXTAPS];        // linear source-position offset of tap t
    const f16* wmat; int ldw, Nout;     // [Nout][ldw] fp16, k contiguous
    f16* y; int ldy;
    const float* bias;                  // optional [Nout]
    float* stat_part;                   // optional [gridDim.x][2][Nout]: per-workgroup sum / sum of squares of y (fp32 accumulators)
    int nstages;                        // ceil(M / 128)
};

// 16-byte slot swizzle of a row inside a [rows][KP] fp16 image: the 8 rows one ds_read_b128 phase serves hit 8 distinct slots
template <int KP>
__device__ __forceinline__ int i2t_swz(int row) { return KP == 128 ? (row & 7) : ((row >> 1) & 3); }

template <int BN, int KP, int NST>
__global__ __launch_bounds__(256) void sf_igemm2t_kernel(Igemm2tParams p) {
    constexpr int NW = 4, SLICE = 32;
    constexpr int TN = BN / 16, KS = KP / 32;
    constexpr int SL = KP / 8;                      // 16-byte slots per row
    constexpr int RPI = 64 / SL;                    // rows per copy instruction (4 | 16)
    constexpr int XPW = SLICE / RPI;                // copy instructions per slice (8 | 2)
    constexpr int OST = (SLICE * BN / 8) / 64;      // output store instructions per slice (1 | 2)
    constexpr int BNP = BN + 8;                     // staging pitch (halfs)
    constexpr int A_ELEMS = SLICE * KP;
    constexpr int WAVE_ELEMS = NST * A_ELEMS + SLICE * BNP;
    __shared__ __attribute__((aligned(16))) f16 smem[NW * WAVE_ELEMS];
    __shared__ float s_red[NW][2][BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f16* const ring = smem + wave * WAVE_ELEMS;
    f16* const stg = ring + NST * A_ELEMS;
    const f16* const zline = reinterpret_cast<const f16*>(sf_zero_line);
    const int nwg = gridDim.x;
    const int bz = (int)xcd_remap(blockIdx.x, gridDim.x);

    // ---- weights -> MFMA B fragments (registers): fragment (j, ks) = rows n = j*16 + (lane & 15), k = ks*32 + (lane >> 4)*8 ...
    f16x8 bf[TN][KS];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int n = j * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8;
            f16x8 v = zero8();
            if (n < p.Nout && k < p.Ktot) v = ld16(p.wmat + (int64_t)n * p.ldw + k);      // ldw >= roundup(Ktot, 8): zero padded by prep
            bf[j][ks] = v;
        }
    float bias_v[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = j * 16 + (lane & 15);
        bias_v[j] = (p.bias && n < p.Nout) ? p.bias[n] : 0.f;
    }
    // hipcc must see these loads CONSUMED before the loop: a load still pending at the loop header would make it put its own
    // s_waitcnt vmcnt(0) in front of the first MFMA of every iteration -- draining the asm copies it does not know about
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) SF_CONSUME_V(bf[j][ks]);
        SF_CONSUME_V(bias_v[j]);
    }

    // ---- loader geometry of this lane (fixed for the kernel): copy instruction q covers slice rows q*RPI ... , 64 lanes = RPI rows x SL slots
    int xsel[XPW];
    int64_t xcol[XPW];
    uint32_t xbit[XPW];
#pragma unroll
    for (int q = 0; q < XPW; ++q) {
        const int prow = q * RPI + lane / SL, pslot = lane % SL;
        const int lslot = pslot ^ i2t_swz<KP>(prow);
        xsel[q] = prow - q * RPI;
        const int k = lslot * 8;
        if (k < p.Ktot) {
            const int tap = k / p.C, ci = k - tap * p.C;
            xbit[q] = 1u << tap;
            xcol[q] = (int64_t)p.dlin[tap] * p.ld + ci;
        } else {
            xbit[q] = 0u;
            xcol[q] = 0;
        }
    }

    // slice s of this wave: stage bz + s * nwg, rows (stage * 128 + wave * 32) ...
    int nslices = 0;
    if (bz < p.nstages) nslices = (p.nstages - bz + nwg - 1) / nwg;
    auto slice_row0 = [&](int s) { return (bz + s * nwg) * 128 + wave * SLICE; };

    auto tab_rows = [&](int s, int (&epos)[XPW], uint32_t (&emk)[XPW]) {
        const int m_base = slice_row0(s);
        i32x8 t[XPW][RPI / 4];
#pragma unroll
        for (int q = 0; q < XPW; ++q)
#pragma unroll
            for (int r = 0; r < RPI / 4; ++r) t[q][r] = SF_SCALAR_PTR(i32x8, p.rowtab + (m_base + q * RPI))[r];
#pragma unroll
        for (int q = 0; q < XPW; ++q) {
            int pos = 0;
            uint32_t mk = 0u;
#pragma unroll
            for (int r = 0; r < RPI; ++r)
                if (xsel[q] == r) { pos = t[q][r / 4][2 * (r % 4)]; mk = (uint32_t)t[q][r / 4][2 * (r % 4) + 1]; }
            epos[q] = pos;
            emk[q] = (m_base + q * RPI + xsel[q] < p.M) ? mk : 0u;
        }
    };
    auto issue = [&](int buf, const int (&epos)[XPW], const uint32_t (&emk)[XPW]) {
        f16* As = ring + buf * A_ELEMS;
#pragma unroll
        for (int q = 0; q < XPW; ++q) {
            const bool ok = (emk[q] & xbit[q]) != 0u;
            const f16* g = ok ? p.src + ((int64_t)epos[q] * p.ld + xcol[q]) : zline;
            SF_GLOBAL_LOAD_LDS16_ASM(g, As + q * 512);
        }
    };

    float ssum[TN], ssq[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }

    auto compute_store = [&](int s, int buf) {
        const f16* As = ring + buf * A_ELEMS;
        f32x4 acc[2][TN];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f16x8 af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = i * 16 + (lane & 15);
                af[i] = ld16(As + row * KP + (((ks * 4 + (lane >> 4)) ^ i2t_swz<KP>(row)) << 3));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = SF_MFMA16(af[i], bf[j][ks], acc[i][j]);
        }
        const int m_base = slice_row0(s);
        // accumulator element r of tile (i, j): row i*16 + 4*(lane >> 4) + r, column j*16 + (lane & 15)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = j * 16 + (lane & 15);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i * 16 + 4 * (lane >> 4) + r;
                    const float v = acc[i][j][r] + bias_v[j];
                    if (m_base + row < p.M) { ssum[j] += v; ssq[j] += v * v; }
                    stg[row * BNP + col] = (f16)v;
                }
        }
        SF_WAVE_LDS_SYNC();
        // 16-byte row segments: chunk c of the slice = (row c / (BN/8), channel group c % (BN/8)); rows past M are not stored
#pragma unroll
        for (int o = 0; o < OST; ++o) {
            const int c = lane + 64 * o;
            const int row = c / (BN / 8), cg = c % (BN / 8);
            const f16x8 v = ld16(stg + row * BNP + cg * 8);
            const int m = m_base + row;
            if (m < p.M && cg * 8 < p.Nout) st16(p.y + (int64_t)m * p.ldy + cg * 8, v);
        }
        SF_WAVE_LDS_SYNC();             // the staging corner is rewritten by the next slice
    };

    // ---- the wave's private pipeline: slices s+1 .. s+NST-1 in flight while slice s is multiplied.  vmcnt counts the copies AND
    // the result stores (OST per slice): what was issued after the copies of slice s may stay outstanding when slice s is read.
    // A ragged or empty last slice issues fewer stores than counted, so head and tail wait for everything.
    {
        int epos[XPW];
        uint32_t emk[XPW];
        int issued = 0;
        for (; issued < NST - 1 && issued < nslices; ++issued) { tab_rows(issued, epos, emk); issue(issued, epos, emk); }
        if (issued < nslices) tab_rows(issued, epos, emk);
        int cur = 0, nxt = NST - 1;
        for (int s = 0; s < nslices; ++s) {
            // newer than the copies of slice s: the stores of slice s-NST+1, then (copies + stores) of NST-2 iterations
            const bool steady = s >= NST - 1 && s + NST - 1 < nslices && slice_row0(s + NST - 1) + SLICE <= p.M;
            if (steady) SF_WAIT_VMEM_N((NST - 2) * XPW + (NST - 1) * OST);
            else SF_WAIT_VMEM();
            if (issued < nslices) {
                issue(nxt, epos, emk);
                ++issued;
                if (issued < nslices) tab_rows(issued, epos, emk);
            }
            compute_store(s, cur);
            cur = cur == NST - 1 ? 0 : cur + 1;
            nxt = nxt == NST - 1 ? 0 : nxt + 1;
        }
    }

    if (p.stat_part) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float a = wave_sum_over_row_groups(ssum[j]), b = wave_sum_over_row_groups(ssq[j]);
            if (lane < 16) { s_red[wave][0][j * 16 + lane] = a; s_red[wave][1][j * 16 + lane] = b; }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, col = tid % BN;
            if (col < p.Nout)
                p.stat_part[((int64_t)blockIdx.x * 2 + which) * p.Nout + col] =
                    (s_red[0][which][col] + s_red[1][which][col]) + (s_red[2][which][col] + s_red[3][which][col]);
        }
    }
}
