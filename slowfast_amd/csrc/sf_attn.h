// Fused pooled attention: softmax(scale * Q K^T + rel-pos bias) V and its backward without ever writing the score /
// probability matrices (flash-attention style, online softmax), for the MultiScaleAttention core.
//
// Reference call site: slowfast/models/attention.py:355-385 (attn = (q * scale) @ k^T; cal_rel_pos_spatial /
// cal_rel_pos_temporal add rel_h/rel_w/rel_t to the non-cls block; softmax; x = attn @ v; residual pooling adds q to
// the non-cls rows) and its autograd backward.  The unfused path (sf_bgemm + sf_softmax_* + sf_bgemm_tn) moves the
// [B, heads, Nq, Nk] fp16 tensors S, P, dP, dS through HBM ~10 times per block (up to 1.26 GB each in MViTv2-S).
//
// Everything is computed TRANSPOSED so that all per-query quantities are lane-local in the MFMA accumulator layout
// (lane l: column l & 15, rows 4*(l >> 4) .. +3):
//   forward / dQ kernels: a wave owns 16 queries (= accumulator columns) and walks the keys in chunks of 32:
//       S^T[key][q] = K[key][:] . Q[q][:]              A = K rows (LDS, ds_read_b128), B = Q (registers)
//       O^T[d][q]  += V^T[d][key] P^T[key][q]          A = V^T (LDS, ds_read_b64_tr_b16), B = P^T = the S^T
//                                                      accumulators of two 16-key tiles, exponentiated, as they are:
//       MFMA k-slot (g, j) <-> key 4g + j (j < 4) of the first tile, key 16 + 4g + (j - 4) of the second.
//   dK/dV kernel: the roles swap -- a wave owns 16 keys and walks (a split of) the queries in chunks of 32; the query
//       splits write fp32 partials that sf_attn_reduce_kernel sums (Nk is small: without the split only
//       B*heads*Nk/64 workgroups exist).
// The VALU budget per score is what bounds these kernels (12-20 MFMAs vs 8 scores per lane and chunk), so nothing
// per-score goes through lookups or libm:
//   * rel-pos bias(q, key) = rq[q][kh] + rq[q][KH+kw] + rq[q][KH+KW+kt] is ONE more contraction on the matrix cores,
//     OH[key][j] (a constant 0/1 matrix with three ones per row, built once per shape by the caller) times rq[q][j]
//     split into fp16 hi + lo parts (exact to 2^-22); its gradient drq = OH^T dS likewise;
//   * log2(e) is folded into `scale` and rq, exponentials are raw v_exp_f32;
//   * the running maximum is only advanced (and the accumulators rescaled) when it grows by more than 2^8
//     -- P stays <= 256, far inside fp16 -- which makes the rescale a rare wave-uniform branch.
// fp32 softmax statistics; P and dS enter the MFMAs as fp16, exactly like the unfused path stores them.
#pragma once
#include "sf_common.h"
#include "sf_igemm.h"

#define SF_ATTN_RMAX 48            // max kH + kW + kT of the key grid (MViTv2-S: 7+7+8 .. 14+14+8); OH has 64 columns
// LDS row pitches.  A row of D (or 64) halfs plus 16: the pitch in bytes is 32 mod 64, i.e. an odd multiple of 32 --
//   * ds_read_b128 fragment reads (16 rows x one 16-byte k-slot per lane group {0-3,12-15,20-27} etc., MI355X_MICROARCH.md LDS
//     table): 16-byte slot index = row * pitch/16 + g with pitch/16 = 2 mod 4, so the eight rows of a group's first half land on
//     distinct even slots and the eight rows of its second half (g + 1) on distinct odd ones;
//   * ds_read_b64_tr_b16 reads (8 rows x 32 bytes per 32-lane group): row * pitch/4 = odd multiples of 8 banks, eight distinct
//     8-bank windows.
// The former pitch D + 8 (an odd number of 16-byte slots) put 5 of 16 lanes of every b128 group on a busy slot: 38-43 % of the
// LDS cycles of these kernels were bank-conflict cycles (profiles/r4/r4_v3_pmc_tokens.md).
#define SF_ATTN_OHP 80             // LDS pitch of an OH / rq row
#define SF_LOG2E 1.4426950408889634f
#define SF_LN2 0.6931471805599453f

struct AttnParams {
    const f16* q; const f16* k; const f16* v; int ldq, ldk;   // [B][N][heads*D] rows, head h at column h*D
    const f16* o; const f16* dout; int ldo;                   // backward inputs (o includes the residual)
    f16* out; int ldout;                                      // forward: o; dq kernel: dq
    f16* dk; f16* dv; int lddk;
    const float* rq; float* drq; int R;                       // [(b*Nq + q)*heads + head][R]; R = 0: no bias
    const f16* oh;                                            // [roundup(Nk, 32)][64] 0/1
    float* lse; float* delta;                                 // [(b*heads + head)*Nq + q]; lse in log2 units
    float scale2;                                             // scale * log2(e)
    float scale; int residual;
    int B, heads, Nq, Nk, cls;
    int qtiles, ktiles;
    int qsplits, chunks_per_split;                            // dK/dV kernel: query chunks per split
    float* part;                                              // [qsplits][2][B][Nk][heads*D] fp32 (qsplits > 1)
    f16* rqs;   // [(b*Nq + q)*heads + head][2][64]: rq * log2(e) as fp16 hi / lo parts (zero beyond R and on cls rows), written by
                // the query-side backward kernel, copied straight into LDS by the key-side one
    // DIAGNOSTIC (SF_ATTN_ABLATE, tools/token_bench.py only; results are garbage) -- parts of the key-side backward kernel
    // switched off: bit 0 no S / dP MFMAs, bit 1 no softmax arithmetic (exp, p, dS), bit 2 no dV / dK MFMAs (and their
    // transposed LDS reads), bit 3 no workgroup barriers, bit 4 no Q / dO copies inside the loop, bit 5 no rq side loads / stores;
    // forward kernel: 64 key chunks not refreshed after the first two, 128 staging only (no arithmetic), 256 no output stores, 512 no
    // q / rq loads, 1024 no residual loads, 2048 return at once (profiles/r5_v27_attn_fwd_ablation.txt)
    int ablate;
};

// A operand (16 rows x 32 k) whose k-slots are LDS ROWS: slot (g, j) <-> row 4g + j (j < 4) / 16 + 4g + (j - 4),
// rows of the operand = 16 consecutive columns starting at col0 (transposed read)
__device__ __forceinline__ f16x8 attn_tr_frag(const f16* s, int KP, int col0, int pl, int g) {
    f16x8 a;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f16* ptr = s + (16 * h + 4 * g + (pl >> 2)) * KP + col0 + 4 * (pl & 3);
        f16x4 t = as_f16x4(SF_LDS_TR16(ptr));
        a[4 * h + 0] = t[0]; a[4 * h + 1] = t[1]; a[4 * h + 2] = t[2]; a[4 * h + 3] = t[3];
    }
    return a;
}

// q (or k) times scale * log2(e), rounded to fp16 once (what the reference's `q * self.scale` does under autocast): the
// S^T accumulators then hold the scaled logits and the bias MFMAs chain onto them -- no per-score multiply-add
__device__ __forceinline__ f16x8 attn_scale8(f16x8 v, float s) {
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)((float)v[e] * s);
    return o;
}

// columns j0 .. j0 + 7 of an rq row of R floats (times log2 e) -> fp16 hi / lo parts; columns >= R (and everything when !on) are
// zero.  The eight loads are UNCONDITIONAL (column index clamped into the row) so that they -- and those of the other rows of
// the wave -- are in flight together: the predicated form (`on && e < n ? src[e] : 0`) compiled to one exec-masked block with its
// own `s_waitcnt vmcnt(0)` per element, 32 serial round trips in the prologue of the query-side kernels (round 5).
__device__ __forceinline__ void attn_rq_load8(const float* row, int j0, int R, float* raw) {
#pragma unroll
    for (int e = 0; e < 8; ++e) raw[e] = row[j0 + e < R ? j0 + e : R - 1];
}
__device__ __forceinline__ void attn_rq_split8(const float* raw, int j0, int R, bool on, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = (on && j0 + e < R) ? raw[e] * SF_LOG2E : 0.f;
        const f16 h = (f16)v;
        hi[e] = h;
        lo[e] = (f16)(v - (float)h);
    }
}
// the rq rows of the wave's QT query tiles -> hi / lo parts of columns 0 .. 63: per 32-column half, the loads of every tile, then
// the conversions (one wait per half; the second half only exists for key grids with kH + kW + kT > 32)
template <int QT, int NKS = 2>
__device__ __forceinline__ void attn_rq_rows(const float* const* rqrow, const bool* on, int R, int g, f16x8 (*rqh)[NKS], f16x8 (*rql)[NKS]) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
        for (int u = 0; u < QT; ++u) rqh[u][ks] = rql[u][ks] = zero8();
        if (32 * ks < R) {
            float raw[QT][8];
#pragma unroll
            for (int u = 0; u < QT; ++u) attn_rq_load8(rqrow[u], 32 * ks + 8 * g, R, raw[u]);
#pragma unroll
            for (int u = 0; u < QT; ++u) attn_rq_split8(raw[u], 32 * ks + 8 * g, R, on[u], rqh[u][ks], rql[u][ks]);
        }
    }
}

// lane-local maximum of eight scores in four instructions (v_max3_f32; fmaxf costs a canonicalising v_max x, x per operand)
__device__ __forceinline__ float attn_max8(const float* x) {
    return fmaxf(SF_MAX3(x[0], x[1], x[2]), SF_MAX3(x[3], x[4], SF_MAX3(x[5], x[6], x[7])));
}

// ---------------------------------------------------------------------------------------------
// Key-chunk staging of the query-side kernels (forward, dQ): 32 rows of K, V and of the one-hot bias matrix OH travel global ->
// LDS directly (global_load_lds_dwordx4) into one of two buffers while the previous chunk is multiplied -- no staging
// registers, no ds_write pass, ONE barrier per chunk (round 4; the register-staged version needed two and exposed the load
// latency of cold operands: 164 us in the training step against 108 us cache-warm, profiles/r4/r4_v8_attn_ab.txt).  A padded
// [32][KP] image is SPR 16-byte slots per row (the last is the pad), slot i is written by lane i & 63 of copy instruction
// i >> 6; the OH image [32][SF_ATTN_OHP] likewise with RS slots per row.
template <int D>
struct KeyChunkCopy {
    static constexpr int KP = D + 16, SPR = KP / 8, NS = 32 * SPR, NI = (NS + 63) / 64, MSZ = NI * 512;
    static constexpr int RS = SF_ATTN_OHP / 8, RNS = 32 * RS, RNI = (RNS + 63) / 64, RSZ = RNI * 512;
    static constexpr int NJ = 2 * NI + RNI, NCP = (NJ + 3) / 4;
    static constexpr int BUF = 2 * MSZ + RSZ;       // halfs per buffer: [K | V | OH]
    // Per copy instruction of this wave ONE register: (row inside the chunk) << 27 | byte offset of the lane's 16 bytes from the
    // chunk's first row (K / V: row * ldk * 2 + col * 16, OH: row * 128 + col * 16).  The chunk's first row is a wave-uniform
    // base (scalar arithmetic per chunk), the copy takes base + offset (SF_GLOBAL_LOAD_LDS16_SADDR).  Pad slots -- never read --
    // copy their row's first 16 bytes; rows past Nk (last chunk only) copy row Nk - 1: whatever such a key column holds is
    // multiplied by an exactly-zero probability downstream (the kernels mask keys >= Nk themselves).
    uint32_t desc[NCP];
    __device__ __forceinline__ void init(int wave, int lane, int ldk) {
#pragma unroll
        for (int jj = 0; jj < NCP; ++jj) {
            const int j = wave + 4 * jj;
            if (j < 2 * NI) {
                const int i = j < NI ? j : j - NI;
                const int slot = i * 64 + lane, r = slot < NS ? slot / SPR : 0, col = slot < NS ? slot - r * SPR : 0;
                desc[jj] = ((uint32_t)r << 27) | (uint32_t)(r * ldk * 2 + (col < D / 8 ? col : 0) * 16);
            } else {
                const int slot = (j - 2 * NI) * 64 + lane, r = slot < RNS ? slot / RS : 0, col = slot < RNS ? slot - r * RS : 0;
                desc[jj] = ((uint32_t)r << 27) | (uint32_t)(r * 128 + (col < 8 ? col : 0) * 16);
            }
        }
    }
    // chunk c of (kbase, vbase: rows of pitch ldk, Nk valid rows; oh: [roundup(Nk, 32)][64]) -> buffer at `dst`
    __device__ __forceinline__ void issue(int c, f16* dst, const f16* kbase, const f16* vbase, int ldk, int Nk, const f16* oh,
                                          int wave) const {
        const f16* const kb = kbase + (int64_t)c * 32 * ldk;
        const f16* const vb = vbase + (int64_t)c * 32 * ldk;
        const f16* const ob = oh + (int64_t)c * 32 * 64;
        const int last = Nk - 1 - c * 32;              // last valid row of this chunk (>= 31: every row valid)
#pragma unroll
        for (int jj = 0; jj < NCP; ++jj) {
            const int j = wave + 4 * jj;
            if (j >= NJ) continue;
            uint32_t off = desc[jj] & 0x7ffffffu;
            if (j < 2 * NI) {
                if (last < 31) {                        // wave-uniform: the partial chunk
                    int r = (int)(desc[jj] >> 27);
                    SF_CONSUME_V(r);                    // keeps hipcc from hoisting this arithmetic out of the chunk loop (registers)
                    if (r > last) off -= (uint32_t)((r - last) * ldk * 2);
                }
                SF_GLOBAL_LOAD_LDS16_SADDR(j < NI ? kb : vb, off, dst + (j < NI ? j : MSZ / 512 + (j - NI)) * 512);
            } else {
                if (!oh) continue;
                SF_GLOBAL_LOAD_LDS16_SADDR(ob, off, dst + 2 * MSZ + (j - 2 * NI) * 512);
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// forward: workgroup = 64*QT queries of one (batch, head); wave w owns QT column tiles of 16 queries.  With QT = 2
// every K / V / OH fragment read from LDS feeds two MFMAs (these kernels are LDS-bandwidth bound: 1 KB of operand per
// 16x16x32 MFMA at QT = 1), and the K/V chunks are re-staged half as often.
// B2: the key grid has more than 32 relative-position columns (kH + kW + kT > 32).  Without them (every MViTv2-S stage: 7 + 7 + 8)
// the second hi / lo halves of the rq rows do not exist, and the 96-wide two-tile kernel fits 168 registers: THREE workgroups
// per CU instead of two.  The kernel is a chain of latency phases (rows in, 13 key chunks, rows out) that only overlap ACROSS
// workgroups (profiles/r5_v27_attn_fwd_ablation.txt: the phases add up linearly at two per CU).
template <int KD, int QT, bool B2>
__global__ __launch_bounds__(SF_THREADS, QT == 2 ? ((!B2 && KD <= 3) ? 3 : 2) : 1) void sf_attn_fwd_kernel(AttnParams p) {
    constexpr int D = 32 * KD, KP = D + 16, DT = D / 16, NKS = B2 ? 2 : 1;
    typedef KeyChunkCopy<D> Copy;
    __shared__ __attribute__((aligned(16))) f16 KVO[2 * Copy::BUF];      // two buffers of [K | V | OH]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = (int)(bid / (uint32_t)p.qtiles), qt = (int)(bid % (uint32_t)p.qtiles);
    const int b = bh / p.heads, head = bh % p.heads;
    const bool bias = p.R > 0;
    constexpr bool bias2 = B2;
    if (SF_ABLATE(p) & 2048) return;                // (diagnostic: dispatch cost alone)
    const f16* kbase = p.k + (int64_t)b * p.Nk * p.ldk + head * D;
    const f16* vbase = p.v + (int64_t)b * p.Nk * p.ldk + head * D;
    const int nch = (p.Nk + 31) / 32;
    // the first key chunk is on its way before the wave's own rows are asked for: the two round trips overlap
    Copy cp;
    cp.init(wave, lane, p.ldk);
    cp.issue(0, KVO, kbase, vbase, p.ldk, p.Nk, bias ? p.oh : nullptr, wave);
    int qrow[QT];
    const f16* qptr[QT];
    const float* rqrow[QT];
    bool rq_on[QT];
    f16x8 qf[QT][KD], rqh[QT][NKS], rql[QT][NKS];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        qrow[u] = qt * 64 * QT + (wave * QT + u) * 16 + pl;
        const int qc = qrow[u] < p.Nq ? qrow[u] : p.Nq - 1;
        qptr[u] = p.q + ((int64_t)b * p.Nq + qc) * p.ldq + head * D;
#pragma unroll
        for (int s = 0; s < KD; ++s) qf[u][s] = (SF_ABLATE(p) & 512) ? zero8() : ld16(qptr[u] + 32 * s + 8 * g);
        rq_on[u] = bias && qc >= p.cls;
        rqrow[u] = p.rq + (((int64_t)b * p.Nq + qc) * p.heads + head) * p.R;
    }
    attn_rq_rows<QT, NKS>(rqrow, rq_on, (SF_ABLATE(p) & 512) ? 0 : p.R, g, rqh, rql);
#pragma unroll
    for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int s = 0; s < KD; ++s) qf[u][s] = attn_scale8(qf[u][s], p.scale2);

    float m[QT], l[QT];
    f32x4 oacc[QT][DT];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        m[u] = -INFINITY;
        l[u] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oacc[u][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int c = 0; c < nch; ++c) {
        const f16* const Ks = KVO + (c & 1) * Copy::BUF;
        const f16* const Vs = Ks + Copy::MSZ;
        const f16* const OHs = Ks + 2 * Copy::MSZ;
        SF_WAIT_VMEM();             // this wave's copies of chunk c have landed ...
        __syncthreads();            // ... and everybody else's; nobody reads chunk c - 1 any more
        // (diagnostic bit 64: the chunks are not refreshed after the first two)
        if (c + 1 < nch && !((SF_ABLATE(p) & 64) && c >= 1))
            cp.issue(c + 1, KVO + ((c + 1) & 1) * Copy::BUF, kbase, vbase, p.ldk, p.Nk, bias ? p.oh : nullptr, wave);
        if (SF_ABLATE(p) & 128) continue;           // (diagnostic bit 128: staging only, no arithmetic)
        float x[QT][8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 st[QT];
#pragma unroll
            for (int u = 0; u < QT; ++u) st[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                const f16x8 kf = ld16(Ks + (16 * t + pl) * KP + 32 * s + 8 * g);
#pragma unroll
                for (int u = 0; u < QT; ++u) st[u] = SF_MFMA16(kf, qf[u][s], st[u]);
            }
            if (bias) {
                const f16x8 a0 = ld16(OHs + (16 * t + pl) * SF_ATTN_OHP + 8 * g);
#pragma unroll
                for (int u = 0; u < QT; ++u) {
                    st[u] = SF_MFMA16(a0, rqh[u][0], st[u]);
                    st[u] = SF_MFMA16(a0, rql[u][0], st[u]);
                }
                if constexpr (bias2) {
                    const f16x8 a1 = ld16(OHs + (16 * t + pl) * SF_ATTN_OHP + 32 + 8 * g);
#pragma unroll
                    for (int u = 0; u < QT; ++u) {
                        st[u] = SF_MFMA16(a1, rqh[u][NKS - 1], st[u]);
                        st[u] = SF_MFMA16(a1, rql[u][NKS - 1], st[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < QT; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) x[u][4 * t + r] = st[u][r];
        }
        if (c == nch - 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (c * 32 + 16 * (i >> 2) + 4 * g + (i & 3) >= p.Nk) {
#pragma unroll
                    for (int u = 0; u < QT; ++u) x[u][i] = -INFINITY;
                }
        }
        // The running maximum of a query (shared by the four lanes of its column) moves only when the chunk's maximum exceeds
        // it by 2^8.  Whether ANY column of the wave does is decided from the lane-local maxima alone -- a column's maximum
        // exceeds m + 8 iff one of its lanes' does -- so the two cross-lane exchanges per tile (ds_bpermute round trips in the
        // middle of the S -> P -> PV dependency chain) run only in the rare chunks that rescale (always the first).
        float lm[QT];
        bool any_grow = false;
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            lm[u] = attn_max8(x[u]);
            any_grow = any_grow || lm[u] > m[u] + 8.f;
        }
        if (__any(any_grow)) {
#pragma unroll
            for (int u = 0; u < QT; ++u) {
                float cmax = lm[u];
                cmax = fmaxf(cmax, __shfl_xor(cmax, 16));
                cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
                const bool grow = cmax > m[u] + 8.f;          // also true on the first chunk (m = -inf)
                if (__any(grow)) {
                    const float alpha = grow ? SF_EXP2(m[u] - cmax) : 1.f;
                    if (grow) m[u] = cmax;
                    l[u] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) oacc[u][dt] *= alpha;
                }
            }
        }
        f16x8 pf[QT];
#pragma unroll
        for (int u = 0; u < QT; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float pv = SF_EXP2(x[u][i] - m[u]);
                l[u] += pv;
                pf[u][i] = (f16)pv;
            }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const f16x8 vt = attn_tr_frag(Vs, KP, dt * 16, pl, g);
#pragma unroll
            for (int u = 0; u < QT; ++u) oacc[u][dt] = SF_MFMA16(vt, pf[u], oacc[u][dt]);
        }
    }
    // epilogue: the residual rows (q again) of BOTH tiles are requested before anything is stored -- on gfx950 stores count in
    // vmcnt, so a load issued after a store waits for that store's acknowledgement; the per-tile form (load, wait, store, six
    // times per tile) was a chain of twelve memory round trips per wave
    float inv[QT], lt_[QT];
    f16x4 rv[QT][DT];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        float lt = l[u];
        lt += __shfl_xor(lt, 16);
        lt += __shfl_xor(lt, 32);
        lt_[u] = lt;
        inv[u] = 1.f / lt;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            rv[u][dt] = (f16x4){(f16)0, (f16)0, (f16)0, (f16)0};
            if (p.residual && !(SF_ABLATE(p) & 1024))
                rv[u][dt] = *reinterpret_cast<const f16x4*>(qptr[u] + dt * 16 + 4 * g);   // clamped row: always valid
        }
    }
    // (diagnostic bits 256 / 512 / 1024: no output stores / no q and rq loads / no residual loads)
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        if (qrow[u] < p.Nq && !(SF_ABLATE(p) & 256)) {
            const bool res = p.residual && qrow[u] >= p.cls;
            f16* orow = p.out + ((int64_t)b * p.Nq + qrow[u]) * p.ldout + head * D;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d0 = dt * 16 + 4 * g;
                f16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (f16)(oacc[u][dt][r] * inv[u] + (res ? (float)rv[u][dt][r] : 0.f));
                *reinterpret_cast<f16x4*>(orow + d0) = ov;
            }
            if (g == 0) p.lse[(int64_t)bh * p.Nq + qrow[u]] = m[u] + log2f(lt_[u]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward, query side: dq, drq and delta[q] = sum_d dO (O - residual); same tiling as the forward kernel
template <int KD, int QT>
__global__ __launch_bounds__(SF_THREADS, QT == 2 ? 2 : 3) void sf_attn_bwd_dq_kernel(AttnParams p) {
    constexpr int D = 32 * KD, KP = D + 16, DT = D / 16, JT = SF_ATTN_RMAX / 16;
    typedef KeyChunkCopy<D> Copy;
    __shared__ __attribute__((aligned(16))) f16 KVO[2 * Copy::BUF];      // two buffers of [K | V | OH]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = (int)(bid / (uint32_t)p.qtiles), qt = (int)(bid % (uint32_t)p.qtiles);
    const int b = bh / p.heads, head = bh % p.heads;
    const bool bias = p.R > 0, bias2 = p.R > 32;
    const f16* kbase = p.k + (int64_t)b * p.Nk * p.ldk + head * D;
    const f16* vbase = p.v + (int64_t)b * p.Nk * p.ldk + head * D;
    const int nch = (p.Nk + 31) / 32;
    Copy cp;                                        // first key chunk under way before the wave's own rows (as in the forward kernel)
    cp.init(wave, lane, p.ldk);
    cp.issue(0, KVO, kbase, vbase, p.ldk, p.Nk, bias ? p.oh : nullptr, wave);
    int qrow[QT];
    const f16* doptr[QT];
    bool res[QT];
    f16x8 qf[QT][KD], dof[QT][KD], rqh[QT][2], rql[QT][2];
    float dl[QT], lse[QT];
    // every global load of the prologue first (both tiles: q, dO, O, lse, the rq rows), the stores (delta, the rq split for the
    // key-side kernel) after all of them: a load issued behind a store waits for that store (one vmcnt counter)
    {
        f16x8 of[QT][KD];
        const float* rqrow[QT];
        bool rq_on[QT];
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            qrow[u] = qt * 64 * QT + (wave * QT + u) * 16 + pl;
            const int qc = qrow[u] < p.Nq ? qrow[u] : p.Nq - 1;
            res[u] = p.residual && qc >= p.cls;
            const f16* qptr = p.q + ((int64_t)b * p.Nq + qc) * p.ldq + head * D;
            doptr[u] = p.dout + ((int64_t)b * p.Nq + qc) * p.ldo + head * D;
            const f16* optr = p.o + ((int64_t)b * p.Nq + qc) * p.ldo + head * D;
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                qf[u][s] = ld16(qptr + 32 * s + 8 * g);
                dof[u][s] = ld16(doptr[u] + 32 * s + 8 * g);
                of[u][s] = ld16(optr + 32 * s + 8 * g);
            }
            lse[u] = p.lse[(int64_t)bh * p.Nq + qc];
            rq_on[u] = bias && qc >= p.cls;
            rqrow[u] = p.rq + (((int64_t)b * p.Nq + qc) * p.heads + head) * p.R;
        }
        attn_rq_rows<QT>(rqrow, rq_on, p.R, g, rqh, rql);
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            const bool qok = qrow[u] < p.Nq;
            float d = 0.f;
#pragma unroll
            for (int s = 0; s < KD; ++s) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    d += (float)dof[u][s][e] * ((float)of[u][s][e] - (res[u] ? (float)qf[u][s][e] : 0.f));
            }
            d += __shfl_xor(d, 16);
            d += __shfl_xor(d, 32);
            dl[u] = d;
            SF_CONSUME_V(lse[u]);           // first read inside the key loop otherwise: hipcc's wait for it lands THERE, a vmcnt(0)
                                            // per chunk that also drains the next chunk's copies
#pragma unroll
            for (int s = 0; s < KD; ++s) qf[u][s] = attn_scale8(qf[u][s], p.scale2);      // from here on only S^T uses q
            if (qok && g == 0) p.delta[(int64_t)bh * p.Nq + qrow[u]] = d;
            if (p.rqs && qok) {     // the split the key-side kernel needs for the same rows: made once here, not once per key tile
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    f16* dst = p.rqs + (((int64_t)b * p.Nq + qrow[u]) * p.heads + head) * 128 + 32 * ks + 8 * g;
                    st16(dst, rqh[u][ks]);
                    st16(dst + 64, rql[u][ks]);
                }
            }
        }
    }

    f32x4 dqacc[QT][DT], drqacc[QT][JT];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dqacc[u][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) drqacc[u][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int c = 0; c < nch; ++c) {
        const f16* const Ks = KVO + (c & 1) * Copy::BUF;
        const f16* const Vs = Ks + Copy::MSZ;
        const f16* const OHs = Ks + 2 * Copy::MSZ;
        SF_WAIT_VMEM();             // this wave's copies of chunk c have landed ...
        __syncthreads();            // ... and everybody else's; nobody reads chunk c - 1 any more
        if (c + 1 < nch)
            cp.issue(c + 1, KVO + ((c + 1) & 1) * Copy::BUF, kbase, vbase, p.ldk, p.Nk, bias ? p.oh : nullptr, wave);
        f16x8 dsf[QT];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 st[QT], dp[QT];
#pragma unroll
            for (int u = 0; u < QT; ++u) {
                st[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dp[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                const f16x8 kf = ld16(Ks + (16 * t + pl) * KP + 32 * s + 8 * g);
                const f16x8 vf = ld16(Vs + (16 * t + pl) * KP + 32 * s + 8 * g);
#pragma unroll
                for (int u = 0; u < QT; ++u) {
                    st[u] = SF_MFMA16(kf, qf[u][s], st[u]);
                    dp[u] = SF_MFMA16(vf, dof[u][s], dp[u]);
                }
            }
            if (bias) {
                const f16x8 a0 = ld16(OHs + (16 * t + pl) * SF_ATTN_OHP + 8 * g);
#pragma unroll
                for (int u = 0; u < QT; ++u) {
                    st[u] = SF_MFMA16(a0, rqh[u][0], st[u]);
                    st[u] = SF_MFMA16(a0, rql[u][0], st[u]);
                }
                if (bias2) {
                    const f16x8 a1 = ld16(OHs + (16 * t + pl) * SF_ATTN_OHP + 32 + 8 * g);
#pragma unroll
                    for (int u = 0; u < QT; ++u) {
                        st[u] = SF_MFMA16(a1, rqh[u][1], st[u]);
                        st[u] = SF_MFMA16(a1, rql[u][1], st[u]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // keys beyond Nk: K, V rows are zero -> x = 0, dp = 0; their p must not reach dq / drq
                const bool kin = c * 32 + 16 * t + 4 * g + r < p.Nk;
#pragma unroll
                for (int u = 0; u < QT; ++u) {
                    const float pv = kin ? SF_EXP2(st[u][r] - lse[u]) : 0.f;
                    dsf[u][4 * t + r] = (f16)(pv * (dp[u][r] - dl[u]));
                }
            }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const f16x8 kt_ = attn_tr_frag(Ks, KP, dt * 16, pl, g);
#pragma unroll
            for (int u = 0; u < QT; ++u) dqacc[u][dt] = SF_MFMA16(kt_, dsf[u], dqacc[u][dt]);
        }
        if (bias) {
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
                if (jt * 16 < p.R) {
                    const f16x8 oht = attn_tr_frag(OHs, SF_ATTN_OHP, jt * 16, pl, g);
#pragma unroll
                    for (int u = 0; u < QT; ++u)
                        drqacc[u][jt] = SF_MFMA16(oht, dsf[u], drqacc[u][jt]);
                }
        }
    }
    // the residual rows (dO again) of both tiles are requested before anything is stored (see the forward kernel's epilogue)
    f16x4 rv[QT][DT];
#pragma unroll
    for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            rv[u][dt] = (f16x4){(f16)0, (f16)0, (f16)0, (f16)0};
            if (p.residual) rv[u][dt] = *reinterpret_cast<const f16x4*>(doptr[u] + dt * 16 + 4 * g);   // clamped row: always valid
        }
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        if (qrow[u] >= p.Nq) continue;
        f16* dqrow = p.out + ((int64_t)b * p.Nq + qrow[u]) * p.ldout + head * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = dt * 16 + 4 * g;
            f16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (f16)(dqacc[u][dt][r] * p.scale + (res[u] ? (float)rv[u][dt][r] : 0.f));
            *reinterpret_cast<f16x4*>(dqrow + d0) = ov;
        }
        if (bias) {
            float* drow = p.drq + (((int64_t)b * p.Nq + qrow[u]) * p.heads + head) * p.R;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = jt * 16 + 4 * g + r;
                    if (j < p.R) drow[j] = qrow[u] >= p.cls ? drqacc[u][jt][r] : 0.f;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward, key side: workgroup = 64 * KT keys of one (batch, head) x one split of the queries; wave w owns KT tiles of 16 keys
// (keys (w*KT + u)*16 .. +15) and walks the split's queries in chunks of 32.  KT = 2 (round 4): every Q / dO / rq fragment read
// from LDS feeds TWO MFMAs (one per key tile), as QT = 2 does in the query-side kernels -- at KT = 1 this kernel read ~1 KB of
// LDS per MFMA and took 2.3x the time of the query-side kernel for 8/6 of its flops (profiles/r3/r3_final_mvit_kernel_stats.md).
// B2: more than 32 relative-position columns (see the forward kernel).  Without the second one-hot halves -- and with the copy
// descriptors packed into one register each -- the 96-wide two-tile kernel no longer spills: its 12 spilled registers were
// reloaded inside the query loop, and hipcc puts `s_waitcnt vmcnt(0)` behind every scratch load, which drained the NEXT chunk's
// direct-to-LDS copies six times per chunk (round 5; the loop ran load -> wait -> compute).
template <int KD, int OCC, int KT, bool B2 = true>
__global__ __launch_bounds__(SF_THREADS, OCC) void sf_attn_bwd_dkv_kernel(AttnParams p) {
    constexpr int D = 32 * KD, KP = D + 16, DT = D / 16, NKS = B2 ? 2 : 1;
    // Q and dO chunks travel global -> LDS directly (global_load_lds_dwordx4, two buffers): no staging registers -- the two key
    // tiles of a wave need them for accumulators -- and no ds_write pass.  The padded [32][KP] image is made of SPR 16-byte slots
    // per row (the last one is the pad): slot i of a matrix is written by lane i & 63 of copy instruction i >> 6.
    constexpr int SPR = KP / 8, NS = 32 * SPR, NI = (NS + 63) / 64, MSZ = NI * 512;       // MSZ: halfs per matrix buffer
    // rq rows (hi / lo fp16 parts, 64 columns each) travel the same way from the table the query-side kernel left (p.rqs):
    // RS slots per row (8 data + 2 pad), one buffer [hi | lo] of 2 * RSZ halfs per chunk
    constexpr int RS = SF_ATTN_OHP / 8, RNS = 32 * RS, RNI = (RNS + 63) / 64, RSZ = RNI * 512;
    __shared__ __attribute__((aligned(16))) f16 QO[2 * 2 * MSZ];  // [buffer][Q | dO]
    __shared__ __attribute__((aligned(16))) f16 RHL[2 * 2 * RSZ]; // [buffer][hi | lo]
    __shared__ float s_lse[2][32], s_delta[2][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = (int)(bid % (uint32_t)p.qsplits);
    const int rest = (int)(bid / (uint32_t)p.qsplits);
    const int bh = rest / p.ktiles, kt = rest % p.ktiles;
    const int b = bh / p.heads, head = bh % p.heads;
    const bool bias = p.R > 0;
    constexpr bool bias2 = B2;
    int key[KT];
    f16x8 kf[KT][KD], vf[KT][KD], ohb[KT][NKS];
#pragma unroll
    for (int u = 0; u < KT; ++u) {
        key[u] = kt * 64 * KT + (wave * KT + u) * 16 + pl;
        const int kc_ = key[u] < p.Nk ? key[u] : p.Nk - 1;
        const f16* kptr = p.k + ((int64_t)b * p.Nk + kc_) * p.ldk + head * D;
        const f16* vptr = p.v + ((int64_t)b * p.Nk + kc_) * p.ldk + head * D;
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            kf[u][s] = attn_scale8(ld16(kptr + 32 * s + 8 * g), p.scale2);   // only S = Q K^T uses the wave's own key rows
            vf[u][s] = ld16(vptr + 32 * s + 8 * g);
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) ohb[u][ks] = bias ? ld16(p.oh + (int64_t)kc_ * 64 + 32 * ks + 8 * g) : zero8();
    }
    // the raw operands (v, the one-hot rows) are first READ inside the query loop: without this hipcc's wait for their loads lands
    // there -- `s_waitcnt vmcnt(0)` in every iteration, draining the next chunk's copies (sf_common.h: SF_CONSUME_V)
#pragma unroll
    for (int u = 0; u < KT; ++u) {
#pragma unroll
        for (int s = 0; s < KD; ++s) SF_CONSUME_V(vf[u][s]);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) SF_CONSUME_V(ohb[u][ks]);
    }
    const f16* qbase = p.q + (int64_t)b * p.Nq * p.ldq + head * D;
    const f16* dobase = p.dout + (int64_t)b * p.Nq * p.ldo + head * D;
    const int nch_all = (p.Nq + 31) / 32;
    const int c0 = split * p.chunks_per_split;
    int c1 = c0 + p.chunks_per_split;
    if (c1 > nch_all) c1 = nch_all;

    f32x4 dkacc[KT][DT], dvacc[KT][DT];
#pragma unroll
    for (int u = 0; u < KT; ++u)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            dkacc[u][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dvacc[u][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    // copy instructions of this wave: j = wave, wave + 4, ... over [Q | dO | rq hi | rq lo] instruction slots.  Per instruction
    // ONE register: (row inside the chunk) << 27 | byte offset of the lane's 16 bytes from the chunk's first row (as in
    // KeyChunkCopy: wave-uniform chunk base + 32-bit offset; pad slots copy their row's first bytes, rows past Nq copy row
    // Nq - 1 -- their probabilities are forced to zero below, `qin`)
    constexpr int NJ = 2 * NI + 2 * RNI;
    constexpr int NCP = (NJ + 3) / 4;
    uint32_t cp_desc[NCP];
#pragma unroll
    for (int jj = 0; jj < NCP; ++jj) {
        const int j = wave + 4 * jj;
        if (j < 2 * NI) {
            const int i = j < NI ? j : j - NI;
            const int slot = i * 64 + lane, row = slot < NS ? slot / SPR : 0, col = slot < NS ? slot - row * SPR : 0;
            cp_desc[jj] = ((uint32_t)row << 27) | (uint32_t)(row * (j < NI ? p.ldq : p.ldo) * 2 + (col < D / 8 ? col : 0) * 16);
        } else {
            const int i = j - 2 * NI < RNI ? j - 2 * NI : j - 2 * NI - RNI;
            const int slot = i * 64 + lane, row = slot < RNS ? slot / RS : 0, col = slot < RNS ? slot - row * RS : 0;
            cp_desc[jj] = ((uint32_t)row << 27) |
                          (uint32_t)(row * p.heads * 256 + (col < 8 ? col : 0) * 16 + (j - 2 * NI < RNI ? 0 : 128));
        }
    }
    const f16* const rqs_b = p.rqs ? p.rqs + (((int64_t)b * p.Nq) * p.heads + head) * 128 : nullptr;
    auto issue_chunk = [&](int c, int buf) {
        f16* Qb = QO + buf * 2 * MSZ;
        f16* Rb = RHL + buf * 2 * RSZ;
        const f16* const qb = qbase + (int64_t)c * 32 * p.ldq;
        const f16* const ob = dobase + (int64_t)c * 32 * p.ldo;
        const f16* const rb = rqs_b + (int64_t)c * 32 * p.heads * 128;
        const int last = p.Nq - 1 - c * 32;             // last valid row of this chunk (>= 31: every row valid)
#pragma unroll
        for (int jj = 0; jj < NCP; ++jj) {
            const int j = wave + 4 * jj;
            if (j >= NJ) continue;
            uint32_t off = cp_desc[jj] & 0x7ffffffu;
            int over = 0;                                   // rows past the end (partial chunk only)
            if (last < 31) {
                int r = (int)(cp_desc[jj] >> 27);
                SF_CONSUME_V(r);                            // keeps hipcc from hoisting this arithmetic out of the chunk loop (registers)
                over = r > last ? r - last : 0;
            }
            if (j < 2 * NI) {
                const bool isq = j < NI;
                off -= (uint32_t)(over * (isq ? p.ldq : p.ldo) * 2);
                SF_GLOBAL_LOAD_LDS16_SADDR(isq ? qb : ob, off, Qb + (isq ? 0 : MSZ) + (isq ? j : j - NI) * 512);
            } else {
                if (!bias || (SF_ABLATE(p) & 32)) continue;
                const int jr = j - 2 * NI;
                off -= (uint32_t)(over * p.heads * 256);
                SF_GLOBAL_LOAD_LDS16_SADDR(rb, off, Rb + (jr < RNI ? jr : RSZ / 512 + (jr - RNI)) * 512);
            }
        }
    };
    // log-sum-exp and delta of row tid (tid < 32), prefetched into registers
    float lsev = 0.f, deltav = 0.f;
    // (unconditional, row index clamped: the predicated form `tid < 32 && q2 < Nq ? load : 0` put an exec-masked block with its own
    // `s_waitcnt vmcnt(0)` right behind the chunk copies -- the copies of chunk c + 1 were drained before chunk c was computed.
    // Rows past Nq get row Nq - 1's statistics; their probabilities are forced to zero below, `qin`.)
    auto side_load = [&](int c) {
        const int q2 = c * 32 + (tid & 31);
        const int64_t at = (int64_t)bh * p.Nq + (q2 < p.Nq ? q2 : p.Nq - 1);
        lsev = p.lse[at];
        deltav = p.delta[at];
    };
    if (c0 < c1) {
        issue_chunk(c0, 0);
        side_load(c0);
    }
    // ONE barrier per chunk: every loop input is double-buffered (Q / dO / rq by direct-to-LDS copies, lse / delta through
    // registers).  Chunk c + 1 is copied into the buffers chunk c - 1 was read from; every wave finished chunk c - 1 before it
    // arrived at the barrier of chunk c.
    for (int c = c0; c < c1; ++c) {
        const int buf = (c - c0) & 1;
        const f16* const Qs = QO + buf * 2 * MSZ;
        const f16* const Os = Qs + MSZ;
        const f16* const Rh = RHL + buf * 2 * RSZ;
        const f16* const Rl = Rh + RSZ;
        if (tid < 32) {
            s_lse[buf][tid] = lsev;
            s_delta[buf][tid] = deltav;
        }
        SF_WAIT_VMEM();             // this wave's copies of chunk c have landed ...
        if (!(SF_ABLATE(p) & 8)) __syncthreads();   // ... and everybody else's; chunk c - 1 is no longer read by anyone
        if (c + 1 < c1) {
            if (!(SF_ABLATE(p) & 16)) issue_chunk(c + 1, buf ^ 1);
            side_load(c + 1);
        }
        f16x8 pf[KT], dsf[KT];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 st[KT], dp[KT];
#pragma unroll
            for (int u = 0; u < KT; ++u) {
                st[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dp[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (!(SF_ABLATE(p) & 1)) {
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                const f16x8 qa = ld16(Qs + (16 * t + pl) * KP + 32 * s + 8 * g);
                const f16x8 oa = ld16(Os + (16 * t + pl) * KP + 32 * s + 8 * g);
#pragma unroll
                for (int u = 0; u < KT; ++u) {
                    st[u] = SF_MFMA16(qa, kf[u][s], st[u]);
                    dp[u] = SF_MFMA16(oa, vf[u][s], dp[u]);
                }
            }
            }
            if (bias && !(SF_ABLATE(p) & 1)) {
                const f16x8 rh0 = ld16(Rh + (16 * t + pl) * SF_ATTN_OHP + 8 * g);
                const f16x8 rl0 = ld16(Rl + (16 * t + pl) * SF_ATTN_OHP + 8 * g);
#pragma unroll
                for (int u = 0; u < KT; ++u) {
                    st[u] = SF_MFMA16(rh0, ohb[u][0], st[u]);
                    st[u] = SF_MFMA16(rl0, ohb[u][0], st[u]);
                }
                if constexpr (bias2) {
                    const f16x8 rh1 = ld16(Rh + (16 * t + pl) * SF_ATTN_OHP + 32 + 8 * g);
                    const f16x8 rl1 = ld16(Rl + (16 * t + pl) * SF_ATTN_OHP + 32 + 8 * g);
#pragma unroll
                    for (int u = 0; u < KT; ++u) {
                        st[u] = SF_MFMA16(rh1, ohb[u][NKS - 1], st[u]);
                        st[u] = SF_MFMA16(rl1, ohb[u][NKS - 1], st[u]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = 16 * t + 4 * g + r;
                const bool qin = c * 32 + qi < p.Nq;
                const float ls = s_lse[buf][qi], de = s_delta[buf][qi];
#pragma unroll
                for (int u = 0; u < KT; ++u) {
                    if (SF_ABLATE(p) & 2) {             // diagnostic: keep the dependency, drop the arithmetic
                        pf[u][4 * t + r] = (f16)st[u][r];
                        dsf[u][4 * t + r] = (f16)dp[u][r];
                        continue;
                    }
                    const float pv = qin ? SF_EXP2(st[u][r] - ls) : 0.f;
                    pf[u][4 * t + r] = (f16)pv;
                    dsf[u][4 * t + r] = (f16)(pv * (dp[u][r] - de));
                }
            }
        }
        if (SF_ABLATE(p) & 4) {
#pragma unroll
            for (int u = 0; u < KT; ++u) { SF_KEEP_ALIVE(pf[u]); SF_KEEP_ALIVE(dsf[u]); }
        } else {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const f16x8 ot = attn_tr_frag(Os, KP, dt * 16, pl, g);
            const f16x8 qt_ = attn_tr_frag(Qs, KP, dt * 16, pl, g);
#pragma unroll
            for (int u = 0; u < KT; ++u) {
                dvacc[u][dt] = SF_MFMA16(ot, pf[u], dvacc[u][dt]);
                dkacc[u][dt] = SF_MFMA16(qt_, dsf[u], dkacc[u][dt]);
            }
        }
        }
    }
#pragma unroll
    for (int u = 0; u < KT; ++u) {
        if (key[u] >= p.Nk) continue;
        if (p.qsplits > 1) {
            const int64_t C = (int64_t)p.heads * D;
            const int64_t slab = (int64_t)p.B * p.Nk * C;
            float* pk = p.part + ((int64_t)split * 2) * slab + ((int64_t)b * p.Nk + key[u]) * C + head * D;
            float* pv_ = pk + slab;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d0 = dt * 16 + 4 * g;
                *reinterpret_cast<f32x4*>(pk + d0) = dkacc[u][dt];
                *reinterpret_cast<f32x4*>(pv_ + d0) = dvacc[u][dt];
            }
            continue;
        }
        f16* dkrow = p.dk + ((int64_t)b * p.Nk + key[u]) * p.lddk + head * D;
        f16* dvrow = p.dv + ((int64_t)b * p.Nk + key[u]) * p.lddk + head * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = dt * 16 + 4 * g;
            f16x4 a, c2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a[r] = (f16)(dkacc[u][dt][r] * p.scale);
                c2[r] = (f16)dvacc[u][dt][r];
            }
            *reinterpret_cast<f16x4*>(dkrow + d0) = a;
            *reinterpret_cast<f16x4*>(dvrow + d0) = c2;
        }
    }
}

// dk = scale * sum_splits part[s][0], dv = sum_splits part[s][1]  (fixed order), fp32 -> fp16
struct AttnReduceParams {
    const float* part; int qsplits;
    int64_t slab;                   // B*Nk*C
    int C;
    f16* dk; f16* dv; int lddk;
    float scale;
    FastDiv fdC4;                   // C / 4
};
__global__ __launch_bounds__(SF_THREADS) void sf_attn_reduce_kernel(AttnReduceParams p) {
    const int64_t total = p.slab / 4 * 2;
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * SF_THREADS) {
        const int which = idx >= p.slab / 4;
        const int64_t e4 = idx - (which ? p.slab / 4 : 0);
        const float* src = p.part + (int64_t)which * p.slab + e4 * 4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < p.qsplits; ++k) s += *reinterpret_cast<const f32x4*>(src + (int64_t)k * 2 * p.slab);
        uint32_t row, c4;
        fd_divmod((uint32_t)e4, p.fdC4, row, c4);
        const float sc = which ? 1.f : p.scale;
        f16x4 o = {(f16)(s[0] * sc), (f16)(s[1] * sc), (f16)(s[2] * sc), (f16)(s[3] * sc)};
        *reinterpret_cast<f16x4*>((which ? p.dv : p.dk) + (int64_t)row * p.lddk + c4 * 4) = o;
    }
}
