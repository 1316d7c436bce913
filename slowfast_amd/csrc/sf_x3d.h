// X3D-specific HBM-bound kernels: per-sample channel means (SE squeeze, head average pool), the SE gate (two tiny
// fully connected layers per sample), and the fused gate * BatchNorm -> Swish/ReLU elementwise pass with its backward.
//
// Reference call sites: slowfast/models/operators.py:15-59 (SE: AdaptiveAvgPool3d -> fc1 -> ReLU -> fc2 -> Sigmoid ->
// x * gate), pytorchvideo Swish x * sigmoid(x) (operators.py:11), resnet_helper.py:229-237 (SE between b_bn and the
// Swish on every other block), head_helper.py:413-438 (X3DHead conv_5 -> BN -> ReLU -> AvgPool3d).
// Tensors are channels-last fp16 rows (n, pos) with S positions per sample; channel counts padded to 8.
#pragma once
#include "sf_bn.h"
#include "sf_common.h"

// ------------------------------------------------------------------------------------------------
// part[(n*chunks + chunk)][0][c] = sum over the chunk's positions of f(y[n,pos,c]),
//   mode 0: f = relu?(y*scale + shift)                       (squeeze / average pool; scale may be null)
//   mode 1: f = dz * act'(g*u) * u,  u = y*scale + shift      (gradient of the SE gate)
struct SampleSumParams {
    RowTile rt;                     // rows = positions of ONE sample (S), blockIdx.z = sample
    int64_t S;
    const f16* y; int ldy;
    const float* scale; const float* shift; int relu;
    int mode;
    const f16* dz; int lddz;
    const float* gate;              // [N][C] or null (gate = 1)
    int swish;                      // mode 1: activation is swish (1) or relu (0)
    float* part;                    // [N*gridDim.x][2][C] (mode 2: [N*gridDim.x][4][C])
    f16* z; int ldz;                // mode 2: du0 = dz * act'(gate * u) * gate, the gate / Swish backward WITHOUT the squeeze term
};

__device__ __forceinline__ float act_grad(float s, int swish) {
    if (swish) {
        const float sg = 1.f / (1.f + expf(-s));
        return sg * (1.f + s * (1.f - sg));
    }
    return s > 0.f ? 1.f : 0.f;
}
__device__ __forceinline__ float act_val(float s, int swish) {
    if (swish) return s / (1.f + expf(-s));
    return s > 0.f ? s : 0.f;
}

__global__ __launch_bounds__(SF_THREADS) void sf_sample_sum_kernel(SampleSumParams p) {
    __shared__ float s_red[SF_THREADS][17];
    int gcol, r0, r1, rstep;
    const bool active = p.rt.init(gcol, r0, r1, rstep);
    const int c = gcol * 8;
    const int n = blockIdx.z;
    float a[8], b[8], a2[8], b2[8], sc[8], sh[8], gt[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = 0.f; b[e] = 0.f; a2[e] = 0.f; b2[e] = 0.f; sc[e] = 1.f; sh[e] = 0.f; gt[e] = 1.f; }
    if (active) {
        if (p.scale) { load8f(p.scale + c, sc); load8f(p.shift + c, sh); }
        if (p.gate) load8f(p.gate + (int64_t)n * p.rt.C + c, gt);
        for (int m = r0; m < r1; m += rstep) {
            const int64_t row = (int64_t)n * p.S + m;
            f16x8 v = ld16(p.y + row * p.ldy + c);
            if (p.mode == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float u = (float)v[e] * sc[e] + sh[e];
                    if (p.relu) u = u > 0.f ? u : 0.f;
                    a[e] += u;
                }
            } else if (p.mode == 1) {
                f16x8 d = ld16(p.dz + row * p.lddz + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float u = (float)v[e] * sc[e] + sh[e];
                    a[e] += (float)d[e] * act_grad(gt[e] * u, p.swish) * u;
                }
            } else {
                // mode 2 (round 6): ONE pass for what sf_gate_grad + sf_gate_act_bwd + sf_bn_bwd_reduce read y and dz three times
                // for -- the gate's gradient sum (a), du0 stored, and the per-sample sums of the stored du0 (b) and du0 * y (a2)
                f16x8 d = ld16(p.dz + row * p.lddz + c), o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float u = (float)v[e] * sc[e] + sh[e];
                    const float t = (float)d[e] * act_grad(gt[e] * u, p.swish);
                    a[e] += t * u;
                    o[e] = (f16)(t * gt[e]);
                    b[e] += (float)o[e];
                    a2[e] += (float)o[e] * (float)v[e];
                }
                st16(p.z + row * p.ldz + c, o);
            }
        }
    }
    if (p.mode == 2) {
        float* o = p.part + ((int64_t)n * gridDim.x + blockIdx.x) * 4 * p.rt.C;
        rowtile_reduce_store(p.rt, active, c, a, b, o, s_red);
        __syncthreads();
        rowtile_reduce_store(p.rt, active, c, a2, b2, o + 2 * p.rt.C, s_red);
        return;
    }
    rowtile_reduce_store(p.rt, active, c, a, b, p.part + ((int64_t)n * gridDim.x + blockIdx.x) * 2 * p.rt.C, s_red);
}

// sums[n][j][c] = sum_chunks part[(n*chunks + k)][j][c], j < 3 of the 4 slots of mode 2
__global__ __launch_bounds__(SF_THREADS) void sf_sample_fold3_kernel(const float* part, int chunks, int C, float* out) {
    const int n = blockIdx.y, j = blockIdx.z;
    const int c = blockIdx.x * SF_THREADS + threadIdx.x;
    if (c >= C) return;
    float s0 = 0.f, s1 = 0.f;
    const float* src = part + ((int64_t)n * chunks * 4 + j) * C + c;
    int k = 0;
    for (; k + 2 <= chunks; k += 2) { s0 += src[(int64_t)k * 4 * C]; s1 += src[(int64_t)(k + 1) * 4 * C]; }
    if (k < chunks) s0 += src[(int64_t)k * 4 * C];
    out[((int64_t)n * 3 + j) * C + c] = s0 + s1;
}

// mean[n][c] = inv_count * sum_chunks part[(n*chunks + k)][0][c]
__global__ __launch_bounds__(SF_THREADS) void sf_sample_fold_kernel(const float* part, int chunks, int C, float inv_count,
                                                                     float* out) {
    const int n = blockIdx.y;
    const int c = blockIdx.x * SF_THREADS + threadIdx.x;
    if (c >= C) return;
    float s0 = 0.f, s1 = 0.f;
    const float* src = part + (int64_t)n * chunks * 2 * C + c;
    int k = 0;
    for (; k + 2 <= chunks; k += 2) { s0 += src[(int64_t)k * 2 * C]; s1 += src[(int64_t)(k + 1) * 2 * C]; }
    if (k < chunks) s0 += src[(int64_t)k * 2 * C];
    out[(int64_t)n * C + c] = (s0 + s1) * inv_count;
}

// ------------------------------------------------------------------------------------------------
// SE gate, one workgroup per sample: h = relu(W1 m + b1), g = sigmoid(W2 h + b2).
// W1 [F][C], W2 [C][F] are the nn.Conv3d(.,.,1) weights of SE.fc1 / SE.fc2 (operators.py:49-51); C, F <= 1024.
struct SeGateParams {
    int C, Cp, F;                   // real channels, padded pitch of m/gate rows, squeeze width
    const float* m;                 // [N][Cp] squeezed means
    const float* w1; const float* b1; const float* w2; const float* b2;
    float* h;                       // [N][F] post-ReLU hidden
    float* gate;                    // [N][Cp] (pad channels <- 0)
    // backward
    const float* dgate;             // [N][Cp] d(loss)/d(gate)
    float* dpre2;                   // [N][Cp] gradient w.r.t. fc2 output (pre-sigmoid)
    float* dpre1;                   // [N][F]  gradient w.r.t. fc1 output (pre-ReLU)
    float* dm;                      // [N][Cp] gradient w.r.t. the squeezed means
};
__global__ __launch_bounds__(SF_THREADS) void sf_se_gate_fwd_kernel(SeGateParams p) {
    __shared__ float s_m[1024];
    __shared__ float s_h[1024];
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < p.C; c += SF_THREADS) s_m[c] = p.m[(int64_t)n * p.Cp + c];
    __syncthreads();
    for (int j = threadIdx.x; j < p.F; j += SF_THREADS) {
        float acc = p.b1[j];
        for (int c = 0; c < p.C; ++c) acc += p.w1[(int64_t)j * p.C + c] * s_m[c];
        acc = acc > 0.f ? acc : 0.f;
        s_h[j] = acc;
        p.h[(int64_t)n * p.F + j] = acc;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < p.Cp; c += SF_THREADS) {
        float g = 0.f;
        if (c < p.C) {
            float acc = p.b2[c];
            for (int j = 0; j < p.F; ++j) acc += p.w2[(int64_t)c * p.F + j] * s_h[j];
            g = 1.f / (1.f + expf(-acc));
        }
        p.gate[(int64_t)n * p.Cp + c] = g;
    }
}
__global__ __launch_bounds__(SF_THREADS) void sf_se_gate_bwd_kernel(SeGateParams p) {
    __shared__ float s_d2[1024];
    __shared__ float s_d1[1024];
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < p.Cp; c += SF_THREADS) {
        float d = 0.f;
        if (c < p.C) {
            const float g = p.gate[(int64_t)n * p.Cp + c];
            d = p.dgate[(int64_t)n * p.Cp + c] * g * (1.f - g);
        }
        if (c < 1024) s_d2[c] = d;
        p.dpre2[(int64_t)n * p.Cp + c] = d;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < p.F; j += SF_THREADS) {
        float acc = 0.f;
        for (int c = 0; c < p.C; ++c) acc += p.w2[(int64_t)c * p.F + j] * s_d2[c];
        acc = p.h[(int64_t)n * p.F + j] > 0.f ? acc : 0.f;
        s_d1[j] = acc;
        p.dpre1[(int64_t)n * p.F + j] = acc;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < p.Cp; c += SF_THREADS) {
        float acc = 0.f;
        if (c < p.C)
            for (int j = 0; j < p.F; ++j) acc += p.w1[(int64_t)j * p.C + c] * s_d1[j];
        p.dm[(int64_t)n * p.Cp + c] = acc;
    }
}
// out[i][j] (+)= scale * sum_n a[n*lda + i] * b[n*ldb + j]  (weight gradients of the SE layers); b == null: b = 1, J = 1
__global__ __launch_bounds__(SF_THREADS) void sf_outer_sum_kernel(const float* a, int lda, const float* b, int ldb, int N,
                                                                   int I, int J, float* out, float scale, int accumulate) {
    const int idx = blockIdx.x * SF_THREADS + threadIdx.x;
    if (idx >= I * J) return;
    const int i = idx / J, j = idx % J;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += a[(int64_t)n * lda + i] * (b ? b[(int64_t)n * ldb + j] : 1.f);
    s *= scale;
    out[idx] = accumulate ? out[idx] + s : s;
}

// ------------------------------------------------------------------------------------------------
// z = act(gate[n][c] * (y*scale + shift)),  act = swish | relu   (gate null = 1)
// backward: du = dz * act'(g*u) * g + dmean[n][c] * inv_S    (gradient w.r.t. u = BatchNorm output)
struct GateActParams {
    RowTile rt;                     // rows = N*S
    int64_t S;
    const f16* y; int ldy;
    const float* scale; const float* shift;
    const float* gate;              // [N][C] or null
    int swish;
    f16* z; int ldz;                // fwd out / bwd out (du)
    const f16* dz; int lddz;        // bwd in
    const float* dmean; float inv_S;
    FastDiv fdS;
    float* bn_part;                 // bwd, optional: [gridDim.x][2][C] column sums of du and du * y over the workgroup's rows --
                                    // what sf_bn_bwd_reduce would take in a pass of its own over du and y (round 6)
};
__global__ __launch_bounds__(SF_THREADS) void sf_gate_act_fwd_kernel(GateActParams p) {
    int gcol, r0, r1, rstep;
    if (!p.rt.init(gcol, r0, r1, rstep)) return;
    const int c = gcol * 8;
    float sc[8], sh[8];
    load8f(p.scale + c, sc);
    load8f(p.shift + c, sh);
    for (int m = r0; m < r1; m += rstep) {
        float gt[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gt[e] = 1.f;
        if (p.gate) load8f(p.gate + (int64_t)fd_div((uint32_t)m, p.fdS) * p.rt.C + c, gt);
        f16x8 v = ld16(p.y + (int64_t)m * p.ldy + c), o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)act_val(gt[e] * ((float)v[e] * sc[e] + sh[e]), p.swish);
        st16(p.z + (int64_t)m * p.ldz + c, o);
    }
}
// BNP: the BatchNorm-backward reduction of the BatchNorm in front of the gate rides on this pass (sums of the STORED, 16-bit du
// and of du * y, one partial row per workgroup, fixed order -- the quantities and the rounding of sf_bn_bwd_reduce_kernel)
template <bool BNP>
__global__ __launch_bounds__(SF_THREADS) void sf_gate_act_bwd_kernel(GateActParams p) {
    __shared__ float s_red[BNP ? SF_THREADS : 1][17];
    int gcol, r0, r1, rstep;
    const bool active = p.rt.init(gcol, r0, r1, rstep);
    if (!BNP && !active) return;
    const int c = gcol * 8;
    float sc[8], sh[8], sg[8], sgy[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 0.f; sh[e] = 0.f; sg[e] = 0.f; sgy[e] = 0.f; }
    if (active) {
        load8f(p.scale + c, sc);
        load8f(p.shift + c, sh);
        for (int m = r0; m < r1; m += rstep) {
            const int64_t n = fd_div((uint32_t)m, p.fdS);
            float gt[8], dm[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { gt[e] = 1.f; dm[e] = 0.f; }
            if (p.gate) load8f(p.gate + n * p.rt.C + c, gt);
            if (p.dmean) load8f(p.dmean + n * p.rt.C + c, dm);
            f16x8 v = ld16(p.y + (int64_t)m * p.ldy + c), d = ld16(p.dz + (int64_t)m * p.lddz + c), o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float u = (float)v[e] * sc[e] + sh[e];
                o[e] = (f16)((float)d[e] * act_grad(gt[e] * u, p.swish) * gt[e] + dm[e] * p.inv_S);
                if constexpr (BNP) { sg[e] += (float)o[e]; sgy[e] += (float)o[e] * (float)v[e]; }
            }
            st16(p.z + (int64_t)m * p.ldz + c, o);
        }
    }
    if constexpr (BNP) rowtile_reduce_store(p.rt, active, c, sg, sgy, p.bn_part + (int64_t)blockIdx.x * 2 * p.rt.C, s_red);
}
