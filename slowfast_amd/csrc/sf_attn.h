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
//   dK/dV kernel: the roles swap -- a wave owns 16 keys and walks the queries in chunks of 32.
// The rel-pos bias is bias(q, key) = rq[q][kh] + rq[q][KH + kw] + rq[q][KH + KW + kt] (rq from sf_relpos_gather);
// its gradient drq[q][j] = sum over the keys of bucket j of dS is one more MFMA against a 0/1 operand built in
// registers.  fp32 softmax statistics; P and dS enter the MFMAs as fp16, exactly like the unfused path stores them.
#pragma once
#include "sf_common.h"
#include "sf_igemm.h"

#define SF_ATTN_RMAX 48            // kH + kW + kT of the key grid (MViTv2-S: 7+7+8 .. 14+14+8)

struct AttnParams {
    const f16* q; const f16* k; const f16* v; int ldq, ldk;   // [B][N][heads*D] rows, head h at column h*D
    const f16* o; const f16* dout; int ldo;                   // backward inputs (o includes the residual)
    f16* out; int ldout;                                      // forward: o; dq kernel: dq
    f16* dk; f16* dv; int lddk;
    const float* rq; float* drq; int R;                       // [(b*Nq + q)*heads + head][R]
    float* lse; float* delta;                                 // [(b*heads + head)*Nq + q]
    float scale; int residual;
    int B, heads, Nq, Nk, cls, KH, KW;
    FastDiv fdKW, fdKH;
    int qtiles, ktiles;
};

// packed (kh | kw << 8 | kt << 16) of a key, -1 for the cls key and for keys beyond Nk
__device__ __forceinline__ int attn_key_code(const AttnParams& p, int key) {
    if (key < p.cls || key >= p.Nk) return -1;
    uint32_t r, kw, kt, kh;
    fd_divmod((uint32_t)(key - p.cls), p.fdKW, r, kw);
    fd_divmod(r, p.fdKH, kt, kh);
    return (int)(kh | (kw << 8) | (kt << 16));
}
__device__ __forceinline__ float attn_bias(const float* rqrow, int code, int KH, int KW) {
    return rqrow[code & 255] + rqrow[KH + ((code >> 8) & 255)] + rqrow[KH + KW + (code >> 16)];
}

// stage rows [r0, r0 + 32) of a [N][ld] matrix (columns [0, D)) into LDS rows of pitch KP; rows >= N are zero
template <int D, int KP>
struct RowChunk {
    f16x8 v[2];
    __device__ __forceinline__ void load(const f16* base, int ld, int r0, int N, int tid) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int slot = tid + 256 * u;
            const int row = slot / (D / 8), col = slot % (D / 8);
            const bool ok = slot < 32 * (D / 8) && r0 + row < N;
            v[u] = ok ? ld16(base + (int64_t)(r0 + row) * ld + col * 8) : zero8();
        }
    }
    __device__ __forceinline__ void store(f16* s, int tid) const {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int slot = tid + 256 * u;
            if (slot < 32 * (D / 8)) st16(s + (slot / (D / 8)) * KP + (slot % (D / 8)) * 8, v[u]);
        }
    }
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

// ---------------------------------------------------------------------------------------------
// forward: workgroup = 64 queries of one (batch, head); wave w owns queries 16w .. 16w+15
template <int KD>
__global__ __launch_bounds__(SF_THREADS) void sf_attn_fwd_kernel(AttnParams p) {
    constexpr int D = 32 * KD, KP = D + 8, DT = D / 16;
    __shared__ __attribute__((aligned(16))) f16 Ks[32 * KP];
    __shared__ __attribute__((aligned(16))) f16 Vs[32 * KP];
    __shared__ float s_rq[4][16][SF_ATTN_RMAX];
    __shared__ int s_code[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = (int)(bid / (uint32_t)p.qtiles), qt = (int)(bid % (uint32_t)p.qtiles);
    const int b = bh / p.heads, head = bh % p.heads;
    const int qrow = qt * 64 + wave * 16 + pl;
    const bool qok = qrow < p.Nq;
    const int qc = qok ? qrow : p.Nq - 1;
    const f16* qptr = p.q + ((int64_t)b * p.Nq + qc) * p.ldq + head * D;
    f16x8 qf[KD];
#pragma unroll
    for (int s = 0; s < KD; ++s) qf[s] = ld16(qptr + 32 * s + 8 * g);
    for (int i = lane; i < 16 * p.R; i += 64) {
        const int rr = i / p.R, j = i - rr * p.R;
        const int qr = qt * 64 + wave * 16 + rr;
        float v = 0.f;
        if (p.rq && qr < p.Nq && qr >= p.cls) v = p.rq[(((int64_t)b * p.Nq + qr) * p.heads + head) * p.R + j];
        s_rq[wave][rr][j] = v;
    }
    const bool qbias = p.rq != nullptr && qc >= p.cls;
    const float* rqrow = s_rq[wave][pl];
    const f16* kbase = p.k + (int64_t)b * p.Nk * p.ldk + head * D;
    const f16* vbase = p.v + (int64_t)b * p.Nk * p.ldk + head * D;
    const int nch = (p.Nk + 31) / 32;

    float m = -INFINITY, l = 0.f;
    f32x4 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    RowChunk<D, KP> kc, vc;
    kc.load(kbase, p.ldk, 0, p.Nk, tid);
    vc.load(vbase, p.ldk, 0, p.Nk, tid);
    for (int c = 0; c < nch; ++c) {
        __syncthreads();
        kc.store(Ks, tid);
        vc.store(Vs, tid);
        if (tid < 32) s_code[tid] = attn_key_code(p, c * 32 + tid);
        __syncthreads();
        if (c + 1 < nch) {
            kc.load(kbase, p.ldk, (c + 1) * 32, p.Nk, tid);
            vc.load(vbase, p.ldk, (c + 1) * 32, p.Nk, tid);
        }
        float x[8];
        float cmax = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KD; ++s)
                st = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld16(Ks + (16 * t + pl) * KP + 32 * s + 8 * g), qf[s], st, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = 16 * t + 4 * g + r;
                float v = st[r] * p.scale;
                const int code = s_code[kk];
                if (qbias && code >= 0) v += attn_bias(rqrow, code, p.KH, p.KW);
                if (c * 32 + kk >= p.Nk) v = -INFINITY;
                x[4 * t + r] = v;
                cmax = fmaxf(cmax, v);
            }
        }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 16));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
        const float m_new = fmaxf(m, cmax);
        const float alpha = expf(m - m_new);
        l *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oacc[dt] *= alpha;
        f16x8 pf;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float pv = expf(x[i] - m_new);
            l += pv;
            pf[i] = (f16)pv;
        }
        m = m_new;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(attn_tr_frag(Vs, KP, dt * 16, pl, g), pf, oacc[dt], 0, 0, 0);
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
    if (qok) {
        const bool res = p.residual && qrow >= p.cls;
        f16* orow = p.out + ((int64_t)b * p.Nq + qrow) * p.ldout + head * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = dt * 16 + 4 * g;
            f16x4 rv = {(f16)0, (f16)0, (f16)0, (f16)0};
            if (res) rv = *reinterpret_cast<const f16x4*>(qptr + d0);
            f16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (f16)(oacc[dt][r] * inv + (float)rv[r]);
            *reinterpret_cast<f16x4*>(orow + d0) = ov;
        }
        if (g == 0) p.lse[(int64_t)bh * p.Nq + qrow] = m + logf(l);
    }
}

// ---------------------------------------------------------------------------------------------
// backward, query side: dq, drq and delta[q] = sum_d dO (O - residual); same tiling as the forward kernel
template <int KD>
__global__ __launch_bounds__(SF_THREADS) void sf_attn_bwd_dq_kernel(AttnParams p) {
    constexpr int D = 32 * KD, KP = D + 8, DT = D / 16, JT = SF_ATTN_RMAX / 16;
    __shared__ __attribute__((aligned(16))) f16 Ks[32 * KP];
    __shared__ __attribute__((aligned(16))) f16 Vs[32 * KP];
    __shared__ float s_rq[4][16][SF_ATTN_RMAX];
    __shared__ int s_code[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = (int)(bid / (uint32_t)p.qtiles), qt = (int)(bid % (uint32_t)p.qtiles);
    const int b = bh / p.heads, head = bh % p.heads;
    const int qrow = qt * 64 + wave * 16 + pl;
    const bool qok = qrow < p.Nq;
    const int qc = qok ? qrow : p.Nq - 1;
    const bool res = p.residual && qc >= p.cls;
    const f16* qptr = p.q + ((int64_t)b * p.Nq + qc) * p.ldq + head * D;
    const f16* doptr = p.dout + ((int64_t)b * p.Nq + qc) * p.ldo + head * D;
    const f16* optr = p.o + ((int64_t)b * p.Nq + qc) * p.ldo + head * D;
    f16x8 qf[KD], dof[KD];
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < KD; ++s) {
        qf[s] = ld16(qptr + 32 * s + 8 * g);
        dof[s] = ld16(doptr + 32 * s + 8 * g);
        const f16x8 ov = ld16(optr + 32 * s + 8 * g);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += (float)dof[s][e] * ((float)ov[e] - (res ? (float)qf[s][e] : 0.f));
    }
    dl += __shfl_xor(dl, 16);
    dl += __shfl_xor(dl, 32);
    const float lse = p.lse[(int64_t)bh * p.Nq + qc];
    if (qok && g == 0) p.delta[(int64_t)bh * p.Nq + qrow] = dl;
    for (int i = lane; i < 16 * p.R; i += 64) {
        const int rr = i / p.R, j = i - rr * p.R;
        const int qr = qt * 64 + wave * 16 + rr;
        float v = 0.f;
        if (p.rq && qr < p.Nq && qr >= p.cls) v = p.rq[(((int64_t)b * p.Nq + qr) * p.heads + head) * p.R + j];
        s_rq[wave][rr][j] = v;
    }
    const bool qbias = p.rq != nullptr && qc >= p.cls;
    const float* rqrow = s_rq[wave][pl];
    const f16* kbase = p.k + (int64_t)b * p.Nk * p.ldk + head * D;
    const f16* vbase = p.v + (int64_t)b * p.Nk * p.ldk + head * D;
    const int nch = (p.Nk + 31) / 32;

    f32x4 dqacc[DT], drqacc[JT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dqacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) drqacc[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    RowChunk<D, KP> kc, vc;
    kc.load(kbase, p.ldk, 0, p.Nk, tid);
    vc.load(vbase, p.ldk, 0, p.Nk, tid);
    for (int c = 0; c < nch; ++c) {
        __syncthreads();
        kc.store(Ks, tid);
        vc.store(Vs, tid);
        if (tid < 32) s_code[tid] = attn_key_code(p, c * 32 + tid);
        __syncthreads();
        if (c + 1 < nch) {
            kc.load(kbase, p.ldk, (c + 1) * 32, p.Nk, tid);
            vc.load(vbase, p.ldk, (c + 1) * 32, p.Nk, tid);
        }
        f16x8 dsf;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                st = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld16(Ks + (16 * t + pl) * KP + 32 * s + 8 * g), qf[s], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld16(Vs + (16 * t + pl) * KP + 32 * s + 8 * g), dof[s], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = 16 * t + 4 * g + r;
                float v = st[r] * p.scale;
                const int code = s_code[kk];
                if (qbias && code >= 0) v += attn_bias(rqrow, code, p.KH, p.KW);
                const float pv = c * 32 + kk < p.Nk ? expf(v - lse) : 0.f;
                dsf[4 * t + r] = (f16)(pv * (dp[r] - dl));
            }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            dqacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(attn_tr_frag(Ks, KP, dt * 16, pl, g), dsf, dqacc[dt], 0, 0, 0);
        if (p.drq) {
            int codes[8];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) codes[4 * h + e] = s_code[16 * h + 4 * g + e];
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                if (jt * 16 < p.R) {
                    const int j = jt * 16 + pl;
                    f16x8 a;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int code = codes[e];
                        const bool hit = code >= 0 && (j == (code & 255) || j == p.KH + ((code >> 8) & 255) ||
                                                       j == p.KH + p.KW + (code >> 16));
                        a[e] = hit ? (f16)1 : (f16)0;
                    }
                    drqacc[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, dsf, drqacc[jt], 0, 0, 0);
                }
            }
        }
    }
    if (qok) {
        f16* dqrow = p.out + ((int64_t)b * p.Nq + qrow) * p.ldout + head * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = dt * 16 + 4 * g;
            f16x4 rv = {(f16)0, (f16)0, (f16)0, (f16)0};
            if (res) rv = *reinterpret_cast<const f16x4*>(doptr + d0);
            f16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (f16)(dqacc[dt][r] * p.scale + (float)rv[r]);
            *reinterpret_cast<f16x4*>(dqrow + d0) = ov;
        }
        if (p.drq) {
            float* drow = p.drq + (((int64_t)b * p.Nq + qrow) * p.heads + head) * p.R;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = jt * 16 + 4 * g + r;
                    if (j < p.R) drow[j] = qrow >= p.cls ? drqacc[jt][r] : 0.f;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward, key side: workgroup = 64 keys of one (batch, head); wave w owns keys 16w .. 16w+15 and walks the queries
template <int KD>
__global__ __launch_bounds__(SF_THREADS) void sf_attn_bwd_dkv_kernel(AttnParams p) {
    constexpr int D = 32 * KD, KP = D + 8, DT = D / 16;
    __shared__ __attribute__((aligned(16))) f16 Qs[32 * KP];
    __shared__ __attribute__((aligned(16))) f16 Os[32 * KP];      // dO rows
    __shared__ float s_rq[32][SF_ATTN_RMAX];
    __shared__ float s_lse[32], s_delta[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = (int)(bid / (uint32_t)p.ktiles), kt = (int)(bid % (uint32_t)p.ktiles);
    const int b = bh / p.heads, head = bh % p.heads;
    const int key = kt * 64 + wave * 16 + pl;
    const bool kok = key < p.Nk;
    const int kc_ = kok ? key : p.Nk - 1;
    const f16* kptr = p.k + ((int64_t)b * p.Nk + kc_) * p.ldk + head * D;
    const f16* vptr = p.v + ((int64_t)b * p.Nk + kc_) * p.ldk + head * D;
    f16x8 kf[KD], vf[KD];
#pragma unroll
    for (int s = 0; s < KD; ++s) {
        kf[s] = ld16(kptr + 32 * s + 8 * g);
        vf[s] = ld16(vptr + 32 * s + 8 * g);
    }
    const int code = attn_key_code(p, kc_);
    const bool kbias = p.rq != nullptr && code >= 0;
    const f16* qbase = p.q + (int64_t)b * p.Nq * p.ldq + head * D;
    const f16* dobase = p.dout + (int64_t)b * p.Nq * p.ldo + head * D;
    const int nch = (p.Nq + 31) / 32;

    f32x4 dkacc[DT], dvacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        dkacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dvacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    RowChunk<D, KP> qc, oc;
    qc.load(qbase, p.ldq, 0, p.Nq, tid);
    oc.load(dobase, p.ldo, 0, p.Nq, tid);
    for (int c = 0; c < nch; ++c) {
        __syncthreads();
        qc.store(Qs, tid);
        oc.store(Os, tid);
        if (tid < 32) {
            const int qr = c * 32 + tid;
            s_lse[tid] = qr < p.Nq ? p.lse[(int64_t)bh * p.Nq + qr] : 0.f;
            s_delta[tid] = qr < p.Nq ? p.delta[(int64_t)bh * p.Nq + qr] : 0.f;
        }
        if (p.rq) {
            for (int i = tid; i < 32 * p.R; i += SF_THREADS) {
                const int rr = i / p.R, j = i - rr * p.R;
                const int qr = c * 32 + rr;
                s_rq[rr][j] = (qr < p.Nq && qr >= p.cls) ? p.rq[(((int64_t)b * p.Nq + qr) * p.heads + head) * p.R + j] : 0.f;
            }
        }
        __syncthreads();
        if (c + 1 < nch) {
            qc.load(qbase, p.ldq, (c + 1) * 32, p.Nq, tid);
            oc.load(dobase, p.ldo, (c + 1) * 32, p.Nq, tid);
        }
        f16x8 pf, dsf;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                st = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld16(Qs + (16 * t + pl) * KP + 32 * s + 8 * g), kf[s], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld16(Os + (16 * t + pl) * KP + 32 * s + 8 * g), vf[s], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = 16 * t + 4 * g + r;
                const int qr = c * 32 + qi;
                float v = st[r] * p.scale;
                if (kbias && qr >= p.cls) v += attn_bias(s_rq[qi], code, p.KH, p.KW);
                const float pv = qr < p.Nq ? expf(v - s_lse[qi]) : 0.f;
                pf[4 * t + r] = (f16)pv;
                dsf[4 * t + r] = (f16)(pv * (dp[r] - s_delta[qi]));
            }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            dvacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(attn_tr_frag(Os, KP, dt * 16, pl, g), pf, dvacc[dt], 0, 0, 0);
            dkacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(attn_tr_frag(Qs, KP, dt * 16, pl, g), dsf, dkacc[dt], 0, 0, 0);
        }
    }
    if (kok) {
        f16* dkrow = p.dk + ((int64_t)b * p.Nk + key) * p.lddk + head * D;
        f16* dvrow = p.dv + ((int64_t)b * p.Nk + key) * p.lddk + head * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = dt * 16 + 4 * g;
            f16x4 a, c2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a[r] = (f16)(dkacc[dt][r] * p.scale);
                c2[r] = (f16)dvacc[dt][r];
            }
            *reinterpret_cast<f16x4*>(dkrow + d0) = a;
            *reinterpret_cast<f16x4*>(dvrow + d0) = c2;
        }
    }
}
