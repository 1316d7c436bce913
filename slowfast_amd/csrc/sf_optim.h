// Training-step glue on the flat gradient memory: what tools/train_net.py:150-172 does between
// loss.backward() and the next iteration --
//     scaler.unscale_(optimizer); [clip_grad_norm_ | clip_grad_value_]; grad_norm = get_grad_norm_(params);
//     scaler.step(optimizer) (skipped when a gradient is inf / NaN); scaler.update()
// -- as THREE launches over one flat fp32 buffer, with no host synchronisation:
//   1. sf_flat_sumsq_kernel     per-block sum of squares + non-finite count of the (still loss-scaled, summed over ranks) gradients
//   2. sf_step_control_kernel   one workgroup: global norm, found_inf, clip coefficient, GradScaler state update
//                               (scale *= backoff on overflow, *= growth after `growth_interval` clean steps), step counter
//   3. sf_flat_sgd_kernel / sf_flat_adamw_kernel   unscale + clip + weight decay + momentum / Adam moments + parameter
//                               update in ONE pass (torch.optim.SGD / AdamW arithmetic, per-parameter lr / weight decay from a
//                               segment table); every thread skips when found_inf is set (GradScaler.step semantics)
// Parameters, gradients and optimizer state are views of flat buffers with identical layout (slowfast_amd.optim.FlatOptimizer).
#pragma once
#include "sf_common.h"

// control block (device memory, fp32 words) shared by the three kernels and readable by the host AFTER the fact
//  [0] loss scale S          [1] growth tracker (clean steps since the last change)   [2] found_inf of this step (0 / 1)
//  [3] global gradient norm of this step (unscaled, mean over ranks; inf when found_inf)   [4] multiplier applied to the raw
//  gradients in the update = clip_coef / (world * S)    [5] optimizer step count (only clean steps count)
//  [6] skipped steps so far   [7] reserved
#define SF_CTL_WORDS 8

struct FlatSumsqParams {
    const float* g;
    int64_t n;
    float* part;        // [gridDim.x][2]: sum of squares, non-finite count
};

__global__ __launch_bounds__(SF_THREADS) void sf_flat_sumsq_kernel(FlatSumsqParams p) {
    __shared__ double s_s[SF_THREADS];
    __shared__ float s_b[SF_THREADS];
    // a bucket of the gradient buffer may start at any element (slowfast_amd.optim: per-bucket partial sums while later buckets
    // are still being exchanged): up to three leading elements are peeled so that the body reads aligned 16-byte vectors
    int64_t head = (4 - (int64_t)((reinterpret_cast<uintptr_t>(p.g) >> 2) & 3)) & 3;
    if (head > p.n) head = p.n;
    const float* const g = p.g + head;
    const int64_t nb = p.n - head;
    const int64_t n4 = nb >> 2;
    double acc = 0.0;
    float bad = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * SF_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * SF_THREADS) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = v[e];
            if (!(fabsf(x) <= 3.0e38f)) bad += 1.f;     // inf or NaN
            else acc += (double)x * (double)x;
        }
    }
    if (blockIdx.x == 0) {
        for (int64_t i = 4 * n4 + threadIdx.x; i < nb; i += SF_THREADS) {
            const float x = g[i];
            if (!(fabsf(x) <= 3.0e38f)) bad += 1.f;
            else acc += (double)x * (double)x;
        }
        if ((int64_t)threadIdx.x < head) {
            const float x = p.g[threadIdx.x];
            if (!(fabsf(x) <= 3.0e38f)) bad += 1.f;
            else acc += (double)x * (double)x;
        }
    }
    s_s[threadIdx.x] = acc;
    s_b[threadIdx.x] = bad;
    __syncthreads();
    for (int h = SF_THREADS / 2; h >= 1; h >>= 1) {
        if ((int)threadIdx.x < h) { s_s[threadIdx.x] += s_s[threadIdx.x + h]; s_b[threadIdx.x] += s_b[threadIdx.x + h]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        p.part[2 * blockIdx.x] = (float)s_s[0];
        p.part[2 * blockIdx.x + 1] = s_b[0];
    }
}

struct StepControlParams {
    const float* part;
    int nblk;
    float* ctl;             // SF_CTL_WORDS
    float world;            // ranks the gradients were SUMMED over
    float clip_norm;        // <= 0: no norm clipping
    int dynamic;            // GradScaler on: update the scale
    float growth, backoff;
    int growth_interval;
};

__global__ __launch_bounds__(SF_THREADS) void sf_step_control_kernel(StepControlParams p) {
    __shared__ double s_s[SF_THREADS];
    __shared__ float s_b[SF_THREADS];
    double acc = 0.0;
    float bad = 0.f;
    for (int i = threadIdx.x; i < p.nblk; i += SF_THREADS) { acc += (double)p.part[2 * i]; bad += p.part[2 * i + 1]; }
    s_s[threadIdx.x] = acc;
    s_b[threadIdx.x] = bad;
    __syncthreads();
    for (int h = SF_THREADS / 2; h >= 1; h >>= 1) {
        if ((int)threadIdx.x < h) { s_s[threadIdx.x] += s_s[threadIdx.x + h]; s_b[threadIdx.x] += s_b[threadIdx.x + h]; }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const float S = p.ctl[0];
    const bool inf = s_b[0] > 0.f;
    const float unscale = 1.f / (p.world * S);
    const float norm = inf ? INFINITY : (float)sqrt(s_s[0]) * unscale;
    float coef = 1.f;
    if (p.clip_norm > 0.f && !inf) {            // torch.nn.utils.clip_grad_norm_: min(1, max_norm / (norm + 1e-6))
        coef = p.clip_norm / (norm + 1e-6f);
        if (coef > 1.f) coef = 1.f;
    }
    p.ctl[2] = inf ? 1.f : 0.f;
    p.ctl[3] = norm;
    p.ctl[4] = unscale * coef;
    if (inf) p.ctl[6] += 1.f; else p.ctl[5] += 1.f;
    if (p.dynamic) {                            // torch.cuda.amp.GradScaler.update()
        float tracker = p.ctl[1];
        float scale = S;
        if (inf) { scale *= p.backoff; tracker = 0.f; }
        else {
            tracker += 1.f;
            if ((int)tracker >= p.growth_interval) { scale *= p.growth; tracker = 0.f; }
        }
        p.ctl[0] = scale;
        p.ctl[1] = tracker;
    }
}

// one entry per parameter: its range in the flat buffers and its parameter group (lr / weight decay travel as kernel
// arguments per group, so a learning-rate schedule needs no device-side table update)
struct FlatSeg {
    int64_t start, end;
    int32_t group, pad;
};
#define SF_OPT_MAX_GROUPS 8

struct FlatUpdateParams {
    float* param;
    const float* grad;
    float* m1;              // momentum buffer / Adam exp_avg
    float* m2;              // Adam exp_avg_sq (unused by SGD)
    const FlatSeg* segs;
    const int32_t* blk_seg; // block -> segment
    const int32_t* blk_off; // block -> first element offset inside the segment (multiples of SF_THREADS * 4)
    const float* ctl;
    float lr[SF_OPT_MAX_GROUPS], wd[SF_OPT_MAX_GROUPS];
    float clip_val;         // > 0: clip_grad_value_ on the unscaled gradient
    // SGD
    float momentum, dampening;
    int nesterov;
    // AdamW
    float beta1, beta2, eps;
};

#define SF_OPT_BLOCK_ELEMS (SF_THREADS * 4)

__device__ __forceinline__ float flat_grad(const FlatUpdateParams& p, int64_t i, float mult) {
    float g = p.grad[i] * mult;
    if (p.clip_val > 0.f) g = fminf(fmaxf(g, -p.clip_val), p.clip_val);
    return g;
}

__global__ __launch_bounds__(SF_THREADS) void sf_flat_sgd_kernel(FlatUpdateParams p) {
    if (p.ctl[2] != 0.f) return;                // overflow: GradScaler.step() skips optimizer.step()
    const FlatSeg sg = p.segs[p.blk_seg[blockIdx.x]];
    const int64_t base = sg.start + p.blk_off[blockIdx.x];
    const float mult = p.ctl[4];
    const float lr = p.lr[sg.group], wd = p.wd[sg.group];
    const bool first = p.ctl[5] == 1.f;         // first clean step: momentum buffer = gradient (torch.optim.SGD)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t i = base + threadIdx.x + (int64_t)e * SF_THREADS;
        if (i >= sg.end) break;
        float w = p.param[i];
        float g = flat_grad(p, i, mult);
        if (wd != 0.f) g += wd * w;
        if (p.momentum != 0.f) {
            float b = first ? g : p.momentum * p.m1[i] + (1.f - p.dampening) * g;
            p.m1[i] = b;
            g = p.nesterov ? g + p.momentum * b : b;
        }
        p.param[i] = w - lr * g;
    }
}

__global__ __launch_bounds__(SF_THREADS) void sf_flat_adamw_kernel(FlatUpdateParams p) {
    if (p.ctl[2] != 0.f) return;
    const FlatSeg sg = p.segs[p.blk_seg[blockIdx.x]];
    const int64_t base = sg.start + p.blk_off[blockIdx.x];
    const float mult = p.ctl[4];
    const float t = p.ctl[5];
    const float lr = p.lr[sg.group], wd = p.wd[sg.group];
    const float bc1 = 1.f - powf(p.beta1, t), bc2 = 1.f - powf(p.beta2, t);
    const float step_size = lr / bc1, bc2s = sqrtf(bc2);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t i = base + threadIdx.x + (int64_t)e * SF_THREADS;
        if (i >= sg.end) break;
        float w = p.param[i];
        const float g = flat_grad(p, i, mult);
        w *= 1.f - lr * wd;                     // decoupled weight decay (torch.optim.AdamW)
        const float m = p.beta1 * p.m1[i] + (1.f - p.beta1) * g;
        const float v = p.beta2 * p.m2[i] + (1.f - p.beta2) * g * g;
        p.m1[i] = m;
        p.m2[i] = v;
        const float denom = sqrtf(v) / bc2s + p.eps;
        p.param[i] = w - step_size * (m / denom);
    }
}
