// Token-space kernels of the MViT path (HBM-bound): LayerNorm, GELU, column sums (bias / affine gradients),
// pooled-attention softmax with the decomposed relative-position bias, K/V transposes.
//
// Reference call sites (slowfast/models): attention.py:428,456 (norm1/norm2), :240-268 (norm_q/k/v over
// head_dim), common.py:7-34 (Mlp: fc1 -> exact-erf GELU -> fc2), attention.py:354-385 (scores, rel-pos bias,
// softmax, attn @ v, residual pooling), :64-147 (cal_rel_pos_spatial / cal_rel_pos_temporal).
// Token tensors are fp16 [rows][C] with a row pitch (elements); statistics and parameter gradients are fp32.
#pragma once
#include "sf_bn.h"
#include "sf_common.h"

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dimension.  L = lanes per row (power of two, 8 channels per lane per slot),
// NS = slots per lane (C <= 8*L*NS).  A wave handles 64/L rows at once.
struct LnParams {
    int M, C;
    const f16* x; int ldx;
    const float* gamma; const float* beta;
    float eps;
    f16* y; int ldy;
    float* mean; float* rstd;       // [M]
    // backward
    const f16* dy; int lddy;
    const f16* resid; int ldr;      // optional: dx += resid (the skip path of a residual stream)
    f16* dx; int lddx;
    float* part;                    // [gridDim.x][part_rows][C]: sum dy*xhat, sum dy [, sum resid, sum dx]
    int part_rows;                  // 2, or 4: also the column sums of the residual operand and of the stored result -- the bias
                                    // gradients of the Linear layers on either side of the LayerNorm (MultiScaleBlock: fc2.bias
                                    // from resid = d(block output), attn.proj.bias from dx), without a pass of their own
    int rows_per_block;
    F32Rows f32;                    // forward: rows with an fp32 side copy (f32.in) are normalised from it instead of from x
};

template <int L>
__device__ __forceinline__ float ln_group_sum(float v) {
#pragma unroll
    for (int m = 1; m < L; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// RU rows per thread and pass are in flight together (round 4: with one row the kernel sat at 2.5 TB/s -- one 16-byte load,
// two shuffle reductions and a store per thread in strict sequence; the loads of the second row now overlap the first row's
// reductions).  RU = 2 for the narrow rows (NS = 1), 1 for C > 512.
template <int L, int NS, int RU>
__global__ __launch_bounds__(SF_THREADS) void sf_layernorm_fwd_kernel(LnParams p) {
    constexpr int RPB = SF_THREADS / L;        // rows per block pass and unroll slot
    const int sub = threadIdx.x % L, rl = threadIdx.x / L;
    float ga[NS][8], be[NS][8];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int c = (sub + s * L) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ga[s][e] = c < p.C ? p.gamma[c + e] : 0.f;
            be[s][e] = c < p.C ? p.beta[c + e] : 0.f;
        }
    }
    const float invC = 1.f / (float)p.C;
    // all lanes of a wave stay in the loop together (shuffles): rows beyond M are computed on zeros, not stored
    for (int base = blockIdx.x * RPB * RU; base < p.M; base += gridDim.x * RPB * RU) {
        float v[RU][NS][8];
        float s1[RU];
        int m[RU];
        bool rowok[RU];
        // every 16-bit row piece of the pass is requested first, UNCONDITIONALLY (offset 0 stands in for rows / columns past the end,
        // the value is replaced by zero afterwards): with the load inside `ok ? ld16(..) : zero8()` / the side-row branch, hipcc
        // waited for row u before it asked for row u + 1 -- the "RU rows in flight" above were one (round 5)
        f16x8 h[RU][NS];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            m[u] = base + u * RPB + rl;
            rowok[u] = m[u] < p.M;
            s1[u] = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int c = (sub + s * L) * 8;
                h[u][s] = ld16(p.x + ((rowok[u] && c < p.C) ? (int64_t)m[u] * p.ldx + c : 0));
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const bool ok = rowok[u] && (sub + s * L) * 8 < p.C;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[u][s][e] = ok ? (float)h[u][s][e] : 0.f;
            }
        if (p.f32.in) {             // rows with an fp32 side copy (class tokens) are normalised from it
#pragma unroll
            for (int u = 0; u < RU; ++u)
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int c = (sub + s * L) * 8;
                    uint32_t srow;
                    if (rowok[u] && c < p.C && f32_row(p.f32, m[u], srow)) load8f(p.f32.in + (int64_t)srow * p.f32.ld + c, v[u][s]);
                }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) s1[u] += v[u][s][e];
            const float mean = ln_group_sum<L>(s1[u]) * invC;
            float s2 = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int c = (sub + s * L) * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = c < p.C ? v[u][s][e] - mean : 0.f;
                    s2 += d * d;
                }
            }
            const float var = ln_group_sum<L>(s2) * invC;
            const float rstd = 1.0f / sqrtf(var + p.eps);
            if (rowok[u]) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int c = (sub + s * L) * 8;
                    if (c < p.C) {
                        f16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (f16)((v[u][s][e] - mean) * rstd * ga[s][e] + be[s][e]);
                        st16(p.y + (int64_t)m[u] * p.ldy + c, o);
                    }
                }
                if (sub == 0) {
                    if (p.mean) p.mean[m[u]] = mean;
                    if (p.rstd) p.rstd[m[u]] = rstd;
                }
            }
        }
    }
}

// dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),  g = dy * gamma;  per-block partial sums of
// dy*xhat (dgamma) and dy (dbeta) for the column reduction.
// RU rows per thread are in flight together and the residual operand is fetched with x and dy at the top of the iteration
// (round 4: with one row per pass and the residual load issued only behind the two shuffle reductions, a thread had two, then
// one, 16-byte loads outstanding -- at the launch's four waves per SIMD that bounded the kernel at ~2.5 TB/s).  RU = 2 takes 154
// VGPRs = three waves per SIMD: the launch is sized to 768 workgroups then, all resident at once (sf_api.hip: ln_bwd_plan).
template <int L, int NS, int RU>
__global__ __launch_bounds__(SF_THREADS) void sf_layernorm_bwd_kernel(LnParams p) {
    constexpr int RPB = SF_THREADS / L;
    __shared__ float s_acc[SF_THREADS][NS * 16 + 1];
    const int sub = threadIdx.x % L, rl = threadIdx.x / L;
    const bool sums = p.part_rows == 4;
    float ga[NS][8], ag[NS][8], ab[NS][8], ar[NS][8], ax[NS][8];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int c = (sub + s * L) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ga[s][e] = c < p.C ? p.gamma[c + e] : 0.f;
            ag[s][e] = 0.f;
            ab[s][e] = 0.f;
            ar[s][e] = 0.f;
            ax[s][e] = 0.f;
        }
    }
    const float invC = 1.f / (float)p.C;
    const int r0 = blockIdx.x * p.rows_per_block;
    int r1 = r0 + p.rows_per_block;
    if (r1 > p.M) r1 = p.M;
    for (int base = r0; base < r1; base += RPB * RU) {
        f16x8 hx[RU][NS], hd[RU][NS], hr[RU][NS];
        float mean[RU], rstd[RU];
        int m[RU];
        bool rowok[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            m[u] = base + u * RPB + rl;
            rowok[u] = m[u] < r1;
            mean[u] = rowok[u] ? p.mean[m[u]] : 0.f;
            rstd[u] = rowok[u] ? p.rstd[m[u]] : 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int c = (sub + s * L) * 8;
                const bool ok = rowok[u] && c < p.C;
                hx[u][s] = ok ? ld16(p.x + (int64_t)m[u] * p.ldx + c) : zero8();
                hd[u][s] = ok ? ld16(p.dy + (int64_t)m[u] * p.lddy + c) : zero8();
                hr[u][s] = (ok && p.resid) ? ld16(p.resid + (int64_t)m[u] * p.ldr + c) : zero8();
            }
        }
        float s1[RU], s2[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            s1[u] = 0.f;
            s2[u] = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const bool ok = rowok[u] && (sub + s * L) * 8 < p.C;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = (float)hd[u][s][e];
                    const float xh = ok ? ((float)hx[u][s][e] - mean[u]) * rstd[u] : 0.f;
                    const float g = d * ga[s][e];
                    s1[u] += g;
                    s2[u] += g * xh;
                    ag[s][e] += d * xh;
                    ab[s][e] += d;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            s1[u] = ln_group_sum<L>(s1[u]) * invC;
            s2[u] = ln_group_sum<L>(s2[u]) * invC;
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            if (!rowok[u]) continue;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int c = (sub + s * L) * 8;
                if (c < p.C) {
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        // xhat and g are recomputed from the 16-bit operands (the same expressions as above): keeping them
                        // for RU rows costs 16 registers per row
                        const float xh = ((float)hx[u][s][e] - mean[u]) * rstd[u];
                        const float g = (float)hd[u][s][e] * ga[s][e];
                        o[e] = (f16)(rstd[u] * (g - s1[u] - xh * s2[u]) + (float)hr[u][s][e]);
                    }
                    st16(p.dx + (int64_t)m[u] * p.lddx + c, o);
                    if (sums) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { ar[s][e] += (float)hr[u][s][e]; ax[s][e] += (float)o[e]; }
                    }
                }
            }
        }
    }
    // fold the RPB row-lanes of the block (fixed order): every thread folds L * NS * 16 / SF_THREADS of the block's
    // L * NS * 16 column sums (a second pass when the extra sums are wanted).  The two passes are spelled out: selecting the
    // accumulator set by a loop variable put all four sets into scratch memory.
    auto fold = [&](const float (&A)[NS][8], const float (&B)[NS][8], int pass) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s_acc[threadIdx.x][s * 16 + e] = A[s][e];
                s_acc[threadIdx.x][s * 16 + 8 + e] = B[s][e];
            }
        __syncthreads();
        float* const o = p.part + ((int64_t)blockIdx.x * p.part_rows + 2 * pass) * p.C;
        for (int idx = threadIdx.x; idx < L * NS * 16; idx += SF_THREADS) {
            const int so = idx / (NS * 16), j = idx - so * (NS * 16), s = j >> 4, jj = j & 15;
            const int c = (so + s * L) * 8 + (jj & 7);
            if (c < p.C) {
                float a = 0.f;
                for (int k = 0; k < RPB; ++k) a += s_acc[so + k * L][j];
                o[(jj >> 3) * p.C + c] = a;
            }
        }
    };
    fold(ag, ab, 0);
    if (sums) {
        __syncthreads();
        fold(ar, ax, 1);
    }
}

// ------------------------------------------------------------------------------------------------
// Column sums of an [M][C] fp16 tensor (bias gradients): part[blk][0][c] = sum_m x[m][c], part[blk][1][c] = 0.
struct ColSumParams {
    RowTile rt;
    const f16* x; int ldx;
    float* part;
};
__global__ __launch_bounds__(SF_THREADS) void sf_colsum_kernel(ColSumParams p) {
    __shared__ float s_red[SF_THREADS][17];
    int gcol, r0, r1, rstep;
    const bool active = p.rt.init(gcol, r0, r1, rstep);
    const int c = gcol * 8;
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = 0.f; b[e] = 0.f; }
    if (active) {
        // four rows in flight per thread (round 4: one load per iteration kept 4 MB in flight chip-wide -- 2.4 TB/s); the
        // additions keep the order of the one-row loop
        int m = r0;
        for (; m + 3 * rstep < r1; m += 4 * rstep) {
            f16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ld16(p.x + (int64_t)(m + u * rstep) * p.ldx + c);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] += (float)v[u][e];
        }
        for (; m < r1; m += rstep) {
            f16x8 v = ld16(p.x + (int64_t)m * p.ldx + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += (float)v[e];
        }
    }
    rowtile_reduce_store(p.rt, active, c, a, b, p.part + (int64_t)blockIdx.x * 2 * p.rt.C, s_red);
}

// out0[c % fold] (+)= scale * sum over rows and over the C/fold channel copies of part[.][0][c]; same for out1.
struct ColFinalizeParams {
    const float* part; int nblk; int row_stride; int C;
    int fold;                // output length (C % fold == 0): channel c contributes to c % fold
    float* out0; float* out1;
    float scale;
    int accumulate;
};
__global__ __launch_bounds__(SF_FIN_THREADS_MAX) void sf_colsum_finalize_kernel(ColFinalizeParams p) {
    __shared__ double s_s[SF_FIN_SEG_MAX][SF_FIN_CH];
    __shared__ double s_q[SF_FIN_SEG_MAX][SF_FIN_CH];
    const int cx = threadIdx.x % SF_FIN_CH, seg = threadIdx.x / SF_FIN_CH;
    const int co = blockIdx.x * SF_FIN_CH + cx;
    double s = 0.0, q = 0.0;
    if (co < p.fold) {
        for (int c = co; c < p.C; c += p.fold) {
            double a, b;
            strided_col_sums(p.part, p.nblk, p.row_stride, p.C, c, seg, a, b);
            s += a;
            q += b;
        }
    }
    fin_fold(s_s, s_q, seg, cx, s, q);
    if (seg == 0 && co < p.fold) {
        const float v0 = (float)(s * p.scale), v1 = (float)(q * p.scale);
        if (p.out0) p.out0[co] = p.accumulate ? p.out0[co] + v0 : v0;
        if (p.out1) p.out1[co] = p.accumulate ? p.out1[co] + v1 : v1;
    }
}

// Several finalizes in ONE launch (round 4: a MultiScaleBlock backward ends in ~9 of them -- two LayerNorms, the three head-dim
// LayerNorms of the pooled q / k / v, four bias gradients -- each a 6-7 us launch for a few KB of work; 149 per MViTv2-S step).
// The descriptors travel BY VALUE in the kernel arguments (no device-side table to keep alive or to upload); block b serves item
// i with first[i] <= b < first[i + 1].
#define SF_COLFIN_BATCH 16
struct ColFinalizeBatch {
    ColFinalizeParams item[SF_COLFIN_BATCH];
    int first[SF_COLFIN_BATCH + 1];
    int n;
};
__global__ __launch_bounds__(SF_FIN_THREADS_MAX) void sf_colsum_finalize_batch_kernel(ColFinalizeBatch b) {
    __shared__ double s_s[SF_FIN_SEG_MAX][SF_FIN_CH];
    __shared__ double s_q[SF_FIN_SEG_MAX][SF_FIN_CH];
    int i = 0;
    while (i + 1 < b.n && (int)blockIdx.x >= b.first[i + 1]) ++i;
    const ColFinalizeParams& p = b.item[i];
    const int cx = threadIdx.x % SF_FIN_CH, seg = threadIdx.x / SF_FIN_CH;
    const int co = ((int)blockIdx.x - b.first[i]) * SF_FIN_CH + cx;
    double s = 0.0, q = 0.0;
    if (co < p.fold) {
        for (int c = co; c < p.C; c += p.fold) {
            double a, bb;
            strided_col_sums(p.part, p.nblk, p.row_stride, p.C, c, seg, a, bb);
            s += a;
            q += bb;
        }
    }
    fin_fold(s_s, s_q, seg, cx, s, q);
    if (seg == 0 && co < p.fold) {
        const float v0 = (float)(s * p.scale), v1 = (float)(q * p.scale);
        if (p.out0) p.out0[co] = p.accumulate ? p.out0[co] + v0 : v0;
        if (p.out1) p.out1[co] = p.accumulate ? p.out1[co] + v1 : v1;
    }
}

// ------------------------------------------------------------------------------------------------
// GELU (exact, erf) on contiguous fp16 arrays; n8 = number of 8-element groups.
__global__ __launch_bounds__(SF_THREADS) void sf_gelu_fwd_kernel(const f16* h, f16* a, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; i < n8; i += (int64_t)gridDim.x * SF_THREADS) {
        f16x8 v = ld16(h + i * 8), o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)gelu_f((float)v[e]);
        st16(a + i * 8, o);
    }
}
__global__ __launch_bounds__(SF_THREADS) void sf_gelu_bwd_kernel(const f16* h, const f16* da, f16* dh, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; i < n8; i += (int64_t)gridDim.x * SF_THREADS) {
        f16x8 v = ld16(h + i * 8), d = ld16(da + i * 8), o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)((float)d[e] * gelu_df((float)v[e]));
        st16(dh + i * 8, o);
    }
}

// ------------------------------------------------------------------------------------------------
// Decomposed relative-position terms of pooled attention.
//   rq[row][j] = sum_c q[row][c] * table(j, q position)[c],   row = (b, token, head), j in [0, KH+KW+KT)
// with table rows gathered through host-built index maps: idx_h[qh][kh], idx_w[qw][kw], idx_t[qt][kt]
// (attention.py:64-147; the UNSCALED q is used).  The cls token (token 0 when cls=1) gets zeros.
struct RelPosParams {
    const f16* q; int ldq;          // [B][Nq][heads*D], pitch of a token row = ldq
    int B, Nq, heads, D;            // D = head dim (96)
    int cls, qT, qH, qW;            // Nq = cls + qT*qH*qW
    int KH, KW, KT;
    const float* rel_h; const float* rel_w; const float* rel_t;   // [rows][D] fp32 parameters
    int rows_h, rows_w, rows_t;
    const int32_t* idx_h; const int32_t* idx_w; const int32_t* idx_t;   // [qH][KH], [qW][KW], [qT][KT]
    float* rq;                      // [B*Nq*heads][R] fp32, R = KH+KW+KT
    // backward
    const float* drq;               // [B*Nq*heads][R]
    f16* dq; int lddq;              // dq += sum_j drq[j] * table_j   (read-modify-write of dq rows)
    float* dtab_part;               // [gridDim.x][(rows_h+rows_w+rows_t)][D] per-block partial tables
    int rows_per_block;
    FastDiv fdHeads, fdNq, fdW, fdH;
};

__device__ __forceinline__ void relpos_row_decode(const RelPosParams& p, uint32_t row, uint32_t& b, uint32_t& tok,
                                                  uint32_t& head, int& qt, int& qh, int& qw, bool& is_cls) {
    uint32_t q;
    fd_divmod(row, p.fdHeads, q, head);
    fd_divmod(q, p.fdNq, b, tok);
    is_cls = p.cls && tok == 0;
    uint32_t pos = is_cls ? 0u : tok - (uint32_t)p.cls, r, w, h, t;
    fd_divmod(pos, p.fdW, r, w);
    fd_divmod(r, p.fdH, t, h);
    qt = (int)t; qh = (int)h; qw = (int)w;
}

// The three contractions with the rel-pos tables run on the MFMA GEMMs over the CONCATENATED table
// Tab = [rel_pos_h; rel_pos_w; rel_pos_t] (TR rows):  G = q Tab^T  (forward),  dq += E Tab,  dTab = E^T q  (backward),
// where E[row][r] scatters drq[row][j] to column r = column of table row j.  These two kernels are the gather
// (G -> rq) and the scatter (drq -> E) between the dense GEMM operands and the per-row (kH+kW+kT) vectors.
// One thread per (row, j) element, consecutive threads on consecutive elements of rq / drq (round 4: the first version gave a
// 64-lane wave to each row of kH + kW + kT = 22 values and moved 44 MB in 54 us); the cls row (token 0 when cls = 1) gets zeros.
__device__ __forceinline__ int relpos_col(const RelPosParams& p, int j, int qt, int qh, int qw) {
    if (j < p.KH) return p.idx_h[qh * p.KH + j];
    if (j < p.KH + p.KW) return p.rows_h + p.idx_w[qw * p.KW + (j - p.KH)];
    return p.rows_h + p.rows_w + p.idx_t[qt * p.KT + (j - p.KH - p.KW)];
}
__global__ __launch_bounds__(SF_THREADS) void sf_relpos_gather_kernel(RelPosParams p, const f16* G, int ldg, FastDiv fdR,
                                                                       int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * SF_THREADS) {
        uint32_t row, j, b, tok, head;
        fd_divmod((uint32_t)i, fdR, row, j);
        int qt, qh, qw;
        bool is_cls;
        relpos_row_decode(p, row, b, tok, head, qt, qh, qw, is_cls);
        p.rq[i] = is_cls ? 0.f : (float)G[(int64_t)row * ldg + relpos_col(p, (int)j, qt, qh, qw)];
    }
}
// The gather through an LDS image of 32 whole rows of G (round 5, as the scatter below): the rows arrive with plain 16-byte loads
// (the element-wise form touched every 128-byte line of G anyway, two bytes at a time behind a dependent index load), the R picks
// per row come out of LDS and rq is written in order.  ldg % 8 == 0, ldg <= SF_RELPOS_SC_LDE,
// G 16-byte aligned; bit-identical.
__global__ __launch_bounds__(SF_THREADS) void sf_relpos_gather_lds_kernel(RelPosParams p, const f16* G, int ldg, FastDiv fdR, int R,
                                                                           int64_t rows) {
    __shared__ __attribute__((aligned(16))) f16 img[32 * 256];
    const int l8 = ldg >> 3;
    for (int64_t r0 = (int64_t)blockIdx.x * 32; r0 < rows; r0 += (int64_t)gridDim.x * 32) {
        const int nr = rows - r0 < 32 ? (int)(rows - r0) : 32;
        const f32x4* const src = reinterpret_cast<const f32x4*>(G + r0 * ldg);
        f32x4* const li = reinterpret_cast<f32x4*>(img);
        for (int i = threadIdx.x; i < nr * l8; i += SF_THREADS) li[i] = src[i];
        __syncthreads();
        for (int i = threadIdx.x; i < nr * R; i += SF_THREADS) {
            uint32_t lr, j, b, tok, head;
            fd_divmod((uint32_t)i, fdR, lr, j);
            const int64_t row = r0 + lr;
            int qt, qh, qw;
            bool is_cls;
            relpos_row_decode(p, (uint32_t)row, b, tok, head, qt, qh, qw, is_cls);
            p.rq[row * R + j] = is_cls ? 0.f : (float)img[(int)lr * ldg + relpos_col(p, (int)j, qt, qh, qw)];
        }
        __syncthreads();            // the image is rewritten by the next chunk
    }
}
// Each workgroup owns chunks of SF_RELPOS_SC_ROWS whole rows of E: it zero-fills them with 16-byte stores and then, behind a
// barrier, drops the rows' R gradient entries into place.  (A hipMemsetAsync in front of a flat scatter did the same in eager
// launches, but as a memset node of a captured graph the fill did not take effect before the readers of E on ROCm 7.2: from
// the second replay on E kept what the block's previous owner had left -- profiles/r4/r4_v13_graph_memset.md.  libsfamd issues
// no memset / memcpy stream operations any more: everything a captured step does is a kernel node.)
#define SF_RELPOS_SC_ROWS 32
__global__ __launch_bounds__(SF_THREADS) void sf_relpos_scatter_kernel(RelPosParams p, f16* E, int lde, FastDiv fdR, int R,
                                                                        int64_t rows) {
    const int l8 = lde >> 3;
    for (int64_t r0 = (int64_t)blockIdx.x * SF_RELPOS_SC_ROWS; r0 < rows; r0 += (int64_t)gridDim.x * SF_RELPOS_SC_ROWS) {
        const int nr = rows - r0 < SF_RELPOS_SC_ROWS ? (int)(rows - r0) : SF_RELPOS_SC_ROWS;
        f32x4* const dst = reinterpret_cast<f32x4*>(E + r0 * lde);
        for (int i = threadIdx.x; i < nr * l8; i += SF_THREADS) dst[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        for (int i = threadIdx.x; i < nr * R; i += SF_THREADS) {
            uint32_t lr, j, b, tok, head;
            fd_divmod((uint32_t)i, fdR, lr, j);
            const int64_t row = r0 + lr;
            int qt, qh, qw;
            bool is_cls;
            relpos_row_decode(p, (uint32_t)row, b, tok, head, qt, qh, qw, is_cls);
            if (!is_cls) E[row * lde + relpos_col(p, (int)j, qt, qh, qw)] = (f16)p.drq[row * R + j];
        }
    }
}

// The same through an LDS image of the chunk (round 5): the chunk's 32 rows are contiguous in E, so the image -- zero-filled, the
// rows' R entries dropped in with 2-byte LDS writes -- leaves as plain 16-byte stores.  The global form above writes every row
// twice, the second time as isolated 2-byte stores (read-modify-write of 32-byte sectors at the L2): 56 us for 47 MB at the
// MViTv2-S stage-3 shape.  Needs lde <= SF_RELPOS_SC_LDE halfs (16 KB image); bit-identical.
#define SF_RELPOS_SC_LDE 256
__global__ __launch_bounds__(SF_THREADS) void sf_relpos_scatter_lds_kernel(RelPosParams p, f16* E, int lde, FastDiv fdR, int R,
                                                                            int64_t rows) {
    __shared__ __attribute__((aligned(16))) f16 img[SF_RELPOS_SC_ROWS * SF_RELPOS_SC_LDE];
    const int l8 = lde >> 3;
    for (int64_t r0 = (int64_t)blockIdx.x * SF_RELPOS_SC_ROWS; r0 < rows; r0 += (int64_t)gridDim.x * SF_RELPOS_SC_ROWS) {
        const int nr = rows - r0 < SF_RELPOS_SC_ROWS ? (int)(rows - r0) : SF_RELPOS_SC_ROWS;
        f32x4* const li = reinterpret_cast<f32x4*>(img);
        for (int i = threadIdx.x; i < nr * l8; i += SF_THREADS) li[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        for (int i = threadIdx.x; i < nr * R; i += SF_THREADS) {
            uint32_t lr, j, b, tok, head;
            fd_divmod((uint32_t)i, fdR, lr, j);
            const int64_t row = r0 + lr;
            int qt, qh, qw;
            bool is_cls;
            relpos_row_decode(p, (uint32_t)row, b, tok, head, qt, qh, qw, is_cls);
            if (!is_cls) img[(int)lr * lde + relpos_col(p, (int)j, qt, qh, qw)] = (f16)p.drq[row * R + j];
        }
        __syncthreads();
        f32x4* const dst = reinterpret_cast<f32x4*>(E + r0 * lde);
        for (int i = threadIdx.x; i < nr * l8; i += SF_THREADS) dst[i] = li[i];
        __syncthreads();            // the image is rewritten by the next chunk
    }
}

// The concatenated table Tab = [rel_pos_h; rel_pos_w; rel_pos_t] as 16-bit GEMM operands in ONE launch: t16 [TRp][D] (rows
// beyond the tables are zero) and its transpose t16t [D][TRp] (round 4: torch.cat + zeros + copy + transpose-copy per block and
// step before); and the way back, the rows of dTab [TRp][D] fp32 into the three parameter gradients (copy or accumulate).
struct RelPosTabParams {
    const float* tab[3]; float* grad[3];
    int rows[3]; int acc[3];
    int D, TRp;
    f16* t16; f16* t16t;            // pack
    const float* dtab;              // unpack
};
__global__ __launch_bounds__(SF_THREADS) void sf_relpos_pack_kernel(RelPosTabParams p) {
    const int total = p.TRp * p.D;
    for (int i = blockIdx.x * SF_THREADS + threadIdx.x; i < total; i += gridDim.x * SF_THREADS) {
        const int r = i / p.D, c = i - r * p.D;
        float v = 0.f;
        if (r < p.rows[0]) v = p.tab[0][r * p.D + c];
        else if (r < p.rows[0] + p.rows[1]) v = p.tab[1][(r - p.rows[0]) * p.D + c];
        else if (r < p.rows[0] + p.rows[1] + p.rows[2]) v = p.tab[2][(r - p.rows[0] - p.rows[1]) * p.D + c];
        p.t16[i] = (f16)v;
        p.t16t[(int64_t)c * p.TRp + r] = (f16)v;
    }
}
__global__ __launch_bounds__(SF_THREADS) void sf_relpos_unpack_kernel(RelPosTabParams p) {
    const int total = (p.rows[0] + p.rows[1] + p.rows[2]) * p.D;
    for (int i = blockIdx.x * SF_THREADS + threadIdx.x; i < total; i += gridDim.x * SF_THREADS) {
        const int r = i / p.D, c = i - r * p.D;
        const int k = r < p.rows[0] ? 0 : (r < p.rows[0] + p.rows[1] ? 1 : 2);
        const int rr = r - (k > 0 ? p.rows[0] : 0) - (k > 1 ? p.rows[1] : 0);
        float* dst = p.grad[k] + rr * p.D + c;
        const float v = p.dtab[i];
        *dst = p.acc[k] ? *dst + v : v;
    }
}

// ------------------------------------------------------------------------------------------------
// Row softmax of the pooled-attention scores with the rel-pos bias, in place:
//   P[row][k] = softmax_k( scale * S[row][k] + bias(row, k) ),   bias = rq[row][kh] + rq[row][KH+kw] + rq[row][KH+KW+kt]
// for non-cls query rows and non-cls keys (attention.py:101-106, 141-145), 0 otherwise.  One wave per row, a row
// is up to 64*8*NSM keys.  Scores are [B][heads][Nq][lds] fp16 (lds = Nk rounded up to 8; pad columns get 0).
struct SoftmaxParams {
    f16* s; int lds;
    int rows;                       // B*heads*Nq
    int Nq, Nk, heads;
    int cls, kT, kH, kW;            // Nk = cls + kT*kH*kW
    float scale;
    const float* rq;                // [B*Nq*heads][R] or null (no rel-pos)
    int R, KH, KW;
    // backward: s holds dP on entry and scale*dS on exit; p = saved probabilities
    const f16* prob;
    float* drq;                     // [B*Nq*heads][R]
    FastDiv fdNq, fdHeads, fdkW, fdkH;
};

__device__ __forceinline__ int64_t softmax_rq_row(const SoftmaxParams& p, uint32_t row, bool& q_is_cls) {
    // score rows are ordered (b, head, q); rq rows are ordered (b, q, head)
    uint32_t bh, q, b, head;
    fd_divmod(row, p.fdNq, bh, q);
    fd_divmod(bh, p.fdHeads, b, head);
    q_is_cls = p.cls && q == 0;
    return ((int64_t)b * p.Nq + q) * p.heads + head;
}

template <int NSM>
__global__ __launch_bounds__(SF_THREADS) void sf_softmax_fwd_kernel(SoftmaxParams p) {
    __shared__ float s_rq[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = blockIdx.x * 4; base < p.rows; base += gridDim.x * 4) {
        const int row = base + wave;
        const bool rowok = row < p.rows;       // surplus waves of the last pass idle through the barriers
        bool qcls = true;
        const bool has_bias = p.rq != nullptr && rowok;
        if (has_bias) {
            const int64_t rr = softmax_rq_row(p, (uint32_t)row, qcls);
            if (lane < p.R) s_rq[wave][lane] = p.rq[rr * p.R + lane];
        }
        __syncthreads();
        f16* srow = p.s + (int64_t)(rowok ? row : 0) * p.lds;
        float v[NSM][8];
        float mx = -INFINITY;
#pragma unroll
        for (int s = 0; s < NSM; ++s) {
            const int k0 = (lane + s * 64) * 8;
            f16x8 h = (rowok && k0 < p.lds) ? ld16(srow + k0) : zero8();
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e;
                float x = -INFINITY;
                if (k < p.Nk) {
                    x = (float)h[e] * p.scale;
                    if (has_bias && !qcls && k >= p.cls) {
                        uint32_t pos = (uint32_t)(k - p.cls), r, kw, kh, kt;
                        fd_divmod(pos, p.fdkW, r, kw);
                        fd_divmod(r, p.fdkH, kt, kh);
                        x += s_rq[wave][kh] + s_rq[wave][p.KH + kw] + s_rq[wave][p.KH + p.KW + kt];
                    }
                }
                v[s][e] = x;
                mx = x > mx ? x : mx;
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 4));
        mx = fmaxf(mx, __shfl_xor(mx, 8)); mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < NSM; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float ex = v[s][e] == -INFINITY ? 0.f : expf(v[s][e] - mx);
                v[s][e] = ex;
                sum += ex;
            }
        sum = ln_group_sum<64>(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int s = 0; s < NSM; ++s) {
            const int k0 = (lane + s * 64) * 8;
            if (rowok && k0 < p.lds) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (f16)(v[s][e] * inv);
                st16(srow + k0, o);
            }
        }
        __syncthreads();
    }
}

// dS = P * (dP - sum_k P*dP); writes scale*dS in place over dP and the bias gradients drq (sums of the UNSCALED dS
// over the keys that share kh / kw / kt).
template <int NSM>
__global__ __launch_bounds__(SF_THREADS) void sf_softmax_bwd_kernel(SoftmaxParams p) {
    __shared__ float s_ds[4][NSM * 512 + 8];     // the wave's unscaled dS row, for the per-(kh | kw | kt) sums
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = blockIdx.x * 4; base < p.rows; base += gridDim.x * 4) {
        const int row = base + wave;
        const bool rowok = row < p.rows;
        bool qcls = true;
        const bool has_bias = p.drq != nullptr && rowok;
        int64_t rr = 0;
        if (has_bias) rr = softmax_rq_row(p, (uint32_t)row, qcls);
        f16* drow = p.s + (int64_t)(rowok ? row : 0) * p.lds;
        const f16* prow = p.prob + (int64_t)(rowok ? row : 0) * p.lds;
        float pv[NSM][8], dv[NSM][8];
        float dot = 0.f;
#pragma unroll
        for (int s = 0; s < NSM; ++s) {
            const int k0 = (lane + s * 64) * 8;
            f16x8 hp = (rowok && k0 < p.lds) ? ld16(prow + k0) : zero8();
            f16x8 hd = (rowok && k0 < p.lds) ? ld16(drow + k0) : zero8();
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = k0 + e < p.Nk;
                pv[s][e] = ok ? (float)hp[e] : 0.f;
                dv[s][e] = ok ? (float)hd[e] : 0.f;
                dot += pv[s][e] * dv[s][e];
            }
        }
        dot = ln_group_sum<64>(dot);
#pragma unroll
        for (int s = 0; s < NSM; ++s) {
            const int k0 = (lane + s * 64) * 8;
            if (rowok && k0 < p.lds) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float ds = pv[s][e] * (dv[s][e] - dot);
                    o[e] = (f16)(ds * p.scale);
                    s_ds[wave][k0 + e] = ds;
                }
                st16(drow + k0, o);
            }
        }
        __syncthreads();
        if (has_bias && lane < p.R) {
            // lane j sums the keys (kt, kh, kw) that share its coordinate, in a fixed order
            float acc = 0.f;
            if (!qcls) {
                const float* d = s_ds[wave] + p.cls;
                if (lane < p.KH) {
                    for (int kt = 0; kt < p.kT; ++kt)
                        for (int kw = 0; kw < p.kW; ++kw) acc += d[(kt * p.kH + lane) * p.kW + kw];
                } else if (lane < p.KH + p.KW) {
                    const int kw = lane - p.KH;
                    for (int kt = 0; kt < p.kT; ++kt)
                        for (int kh = 0; kh < p.kH; ++kh) acc += d[(kt * p.kH + kh) * p.kW + kw];
                } else {
                    const int kt = lane - p.KH - p.KW;
                    for (int i = 0; i < p.kH * p.kW; ++i) acc += d[kt * p.kH * p.kW + i];
                }
            }
            p.drq[rr * p.R + lane] = acc;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// xt[b][head][c][k] = x[b][k][head*D + c] for k < Nk, 0 for Nk <= k < ldk (K-contiguous operand of a GEMM whose
// reduction runs over the keys: P.V and dS.K).
struct TransposeParams {
    const f16* x; int ldx;
    f16* xt; int ldk;
    int B, Nk, heads, D;
    int64_t total;                  // B*heads*D*(ldk/8)
    FastDiv fdK8, fdD, fdHeads;
};
__global__ __launch_bounds__(SF_THREADS) void sf_transpose_heads_kernel(TransposeParams p) {
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < p.total;
         idx += (int64_t)gridDim.x * SF_THREADS) {
        uint32_t q, k8, c, head, b;
        fd_divmod((uint32_t)idx, p.fdK8, q, k8);
        fd_divmod(q, p.fdD, q, c);
        fd_divmod(q, p.fdHeads, b, head);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = (int)k8 * 8 + e;
            o[e] = k < p.Nk ? p.x[((int64_t)b * p.Nk + k) * p.ldx + head * p.D + c] : (f16)0;
        }
        st16(p.xt + (((int64_t)b * p.heads + head) * p.D + c) * p.ldk + k8 * 8, o);
    }
}

// ------------------------------------------------------------------------------------------------
// Stochastic depth (slowfast/models/common.py:46-59 drop_path, used at attention.py:500-510):
//   y[m][c] = (resid ? resid[m][c] : 0) + scale[m / rows_per_sample] * x[m][c],  scale[b] = mask_b / keep_prob.
struct RowScaleParams {
    const f16* x; int ldx;
    const float* scale;
    const f16* resid; int ldr;
    f16* y; int ldy;
    int64_t total;                  // M * (C/8)
    FastDiv fdG, fdRows;            // C/8, rows per sample
    F32Rows f32;                    // fp32 side rows: residual read from f32.in (when given), sum also written to f32.out
};
__global__ __launch_bounds__(SF_THREADS) void sf_row_scale_add_kernel(RowScaleParams p) {
    for (int64_t idx = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; idx < p.total;
         idx += (int64_t)gridDim.x * SF_THREADS) {
        uint32_t m, g8;
        fd_divmod((uint32_t)idx, p.fdG, m, g8);
        const float sc = p.scale[fd_div(m, p.fdRows)];
        const f16x8 v = ld16(p.x + (int64_t)m * p.ldx + g8 * 8);
        f16x8 r = zero8(), o;
        if (p.resid) r = ld16(p.resid + (int64_t)m * p.ldr + g8 * 8);
        uint32_t srow;
        if (p.f32.out && f32_row(p.f32, (int)m, srow)) {
            float rf[8], of[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) rf[e] = (float)r[e];
            if (p.f32.in) load8f(p.f32.in + (int64_t)srow * p.f32.ld + g8 * 8, rf);
#pragma unroll
            for (int e = 0; e < 8; ++e) { of[e] = rf[e] + sc * (float)v[e]; o[e] = (f16)of[e]; }
            float* dst = p.f32.out + (int64_t)srow * p.f32.ld + g8 * 8;
            *reinterpret_cast<f32x4*>(dst) = (f32x4){of[0], of[1], of[2], of[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){of[4], of[5], of[6], of[7]};
            st16(p.y + (int64_t)m * p.ldy + g8 * 8, o);
            continue;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)((float)r[e] + sc * (float)v[e]);
        st16(p.y + (int64_t)m * p.ldy + g8 * 8, o);
    }
}
