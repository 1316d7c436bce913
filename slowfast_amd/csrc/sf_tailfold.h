// In-launch fold of a per-tile partial-sum table: the BatchNorm finalize without its own launch (gfx950).
//
// Reference op: the batch statistics of F.batch_norm(training=True) behind every nn.Conv3d of the ResNet / SlowFast path
// (slowfast/models/resnet_helper.py:340-372, batchnorm_helper.py:16-37) and the per-channel sums of its backward.
//
// Until round 4 a convolution left one row of 2*C partial sums per M tile and a separate launch (sf_bn_finalize /
// sf_bn_bwd_finalize, plus sf_part_fold above 2048 rows) turned the table into per-channel constants: 220 launches of ~7 us per
// SlowFast step, 2.07 ms of a 41.1 ms step (profiles/r5_v1_finalize_ablation.txt) -- latency chains, not work.  Here the
// producing kernel folds its own table, in TWO levels so that no workgroup reads more than ~sqrt(rows) rows:
//   level 0  the M tiles of a column tile are cut into groups of `group` consecutive tiles; the workgroup that draws the LAST
//            ticket of its group sums the group's rows (fixed row order -> the result does not depend on who arrives last) into
//            one fp64 row of `lvl1`;
//   level 1  that workgroup then draws a ticket of the column tile; the last one sums the group rows (fixed order) and runs the
//            finalize arithmetic for its columns.
// Nobody waits for anybody (tickets only): no co-residency assumption, nothing to deadlock.  Visibility across CUs / XCDs
// follows MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility": partial rows and level-1 rows are
// stored write-through (agent-scope relaxed atomic stores = `global_store ... sc1`), every wave drains vmcnt(0), the workgroup
// barrier orders them before lane 0's agent-scope ticket, and the folding workgroup reads them with agent-scope (sc1, L1-
// bypassing) loads.  Tickets are reset by their last drawer, so a table of zeroed counters stays zeroed from launch to launch
// (graph replay included).
#pragma once
#include "sf_common.h"

// agent-scope accessors (the host functional simulator pre-defines them on std atomics)
#ifndef SF_AGENT_LOAD
#define SF_AGENT_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SF_AGENT_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SF_AGENT_FETCH_ADD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SF_DRAIN_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

// Ticket counters sit SF_FOLD_CNT_STRIDE ints (256 bytes) apart: device-scope atomics on ONE line retire at ~12 ns each
// (MI355X_MICROARCH.md "fanin" / "dequeue" rows) -- the 3136 .. 6272 tickets of a res2 layer packed into two lines cost
// 40 - 75 us and made the first version of this fold SLOWER than the launches it removed (profiles/r5_v2_fold_ab.txt).
#define SF_FOLD_CNT_STRIDE 64
struct TailFold {
    int32_t* cnt;           // nullptr: off.  [ngroups * ntiles_n] group tickets, then [ntiles_n] column-tile tickets (each
                            // SF_FOLD_CNT_STRIDE ints apart); all zero
    double* lvl1;           // [ngroups][2][C]
    int group, ngroups;     // M tiles per group, number of groups
    int rows_per_tile;      // partial-table rows one M tile writes (sf_igemm: 1, sf_igemm2 forward statistics: 2)
    int nrows;              // rows of the table that exist (ceil(M / 128) for the statistics, M tiles for the backward sums)
    int mode;               // 1: forward statistics -> scale / shift / mean / rstd (+ running statistics)
                            // 2: backward sums -> coef[3][C], dgamma, dbeta
    int Creal;              // parameter length (channels [Creal, C) are zero padding)
    float count;            // positions the statistics run over
    // mode 1 (BnFinalizeParams of sf_bn.h)
    const float* gamma; const float* beta;
    float* running_mean; float* running_var;
    float momentum, eps;
    float* scale; float* shift; float* save_mean; float* save_rstd;
    // mode 2 (BnBwdFinalizeParams of sf_bn.h): gamma as above
    const float* mean; const float* rstd;
    float inv_loss_scale;
    float* dgamma; float* dbeta; int accumulate;
    float* coef;
};

// the arithmetic of sf_bn_finalize_kernel / sf_bn_bwd_finalize_kernel for ONE channel
__device__ __forceinline__ void tail_finalize_channel(const TailFold& tf, int C, int c, double s, double q) {
    if (tf.mode == 1) {
        if (c >= tf.Creal) {
            tf.scale[c] = 0.f; tf.shift[c] = 0.f;
            if (tf.save_mean) tf.save_mean[c] = 0.f;
            if (tf.save_rstd) tf.save_rstd[c] = 0.f;
            return;
        }
        const double m = s / (double)tf.count;
        double v = q / (double)tf.count - m * m;
        if (v < 0.0) v = 0.0;
        const float mean = (float)m, var = (float)v;
        if (tf.running_mean) {
            const double unb = tf.count > 1.f ? v * (double)tf.count / ((double)tf.count - 1.0) : v;
            tf.running_mean[c] = (1.f - tf.momentum) * tf.running_mean[c] + tf.momentum * mean;
            tf.running_var[c] = (1.f - tf.momentum) * tf.running_var[c] + tf.momentum * (float)unb;
        }
        const float rstd = 1.0f / sqrtf(var + tf.eps);
        const float sc = tf.gamma[c] * rstd;
        tf.scale[c] = sc;
        tf.shift[c] = tf.beta[c] - mean * sc;
        if (tf.save_mean) tf.save_mean[c] = mean;
        if (tf.save_rstd) tf.save_rstd[c] = rstd;
    } else {
        if (c >= tf.Creal) {
            tf.coef[c] = 0.f; tf.coef[C + c] = 0.f; tf.coef[2 * C + c] = 0.f;
            return;
        }
        const double mean = tf.mean[c], rstd = tf.rstd[c], gam = tf.gamma[c];
        const double dbeta = s;                       // sum g
        const double dgamma = rstd * (q - mean * s);  // sum g * xhat
        const double n = tf.count;
        const double k1 = gam * rstd;
        const double k3 = -gam * rstd * rstd * dgamma / n;
        const double k2 = -gam * rstd * dbeta / n - k3 * mean;
        tf.coef[c] = (float)k1;
        tf.coef[C + c] = (float)k2;
        tf.coef[2 * C + c] = (float)k3;
        const float dg = (float)(dgamma * tf.inv_loss_scale), db = (float)(dbeta * tf.inv_loss_scale);
        if (tf.accumulate) { tf.dgamma[c] += dg; tf.dbeta[c] += db; }
        else { tf.dgamma[c] = dg; tf.dbeta[c] = db; }
    }
}

// One ticket of a counter shared by `expected` workgroups: true on the workgroup that draws the last one (which also re-arms the
// counter).  Every thread of the workgroup calls it; `flag` is one int of the kernel's single LDS object.
__device__ __forceinline__ bool tail_last_ticket(int32_t* counter, int expected, volatile int* flag) {
    SF_DRAIN_VMEM();                    // this wave's write-through stores have left
    __syncthreads();                    // ... and every other wave's
    if (threadIdx.x == 0) {
        const int t = SF_AGENT_FETCH_ADD(counter, 1);
        const int last = t == expected - 1;
        if (last) SF_AGENT_STORE(counter, 0);
        *flag = last;
    }
    __syncthreads();
    const bool last = *flag != 0;
    if (!last) return false;            // (uniform over the workgroup) nothing of this workgroup touches LDS again
    __syncthreads();                    // the flag word and `red` are reused right away
    return true;
}

// Sum rows [r0, r1) of a table with 2 * C values per row for the column tile [n0, n0 + ncols): thread t owns (which, column) =
// t % (2 * ncols) and the rows r0 + t / (2 * ncols), + L, + 2L, ...; the L row lanes are combined through `red` in lane order.
// Returns the total on the threads of row lane 0 (0.0 elsewhere / on columns beyond C).
template <int NT, typename T>
__device__ __forceinline__ double tail_col_sums(const T* table, int C, int r0, int r1, int n0, int ncols, double* red) {
    const int cols2 = 2 * ncols;
    const int L = NT / cols2;                       // row lanes (>= 1: the callers' tiles are at most NT / 2 columns wide)
    const int cidx = (int)threadIdx.x % cols2, lane = (int)threadIdx.x / cols2;
    const int which = cidx / ncols, col = n0 + cidx % ncols;
    double a0 = 0.0, a1 = 0.0;
    if (lane < L && col < C) {
        const T* base = table + (int64_t)which * C + col;
        const int64_t rs = (int64_t)2 * C;
        int r = r0 + lane;
#pragma unroll 1
        for (; r + L < r1; r += 2 * L) {            // two loads in flight
            const T u0 = SF_AGENT_LOAD(base + r * rs);
            const T u1 = SF_AGENT_LOAD(base + (r + L) * rs);
            a0 += (double)u0; a1 += (double)u1;
        }
        if (r < r1) a0 += (double)SF_AGENT_LOAD(base + r * rs);
    }
    red[threadIdx.x] = a0 + a1;
    __syncthreads();
    double tot = 0.0;
    if (lane == 0 && col < C)
        for (int l = 0; l < L; ++l) tot += red[l * cols2 + cidx];
    __syncthreads();
    return tot;
}

// EARLY group ticket, drawn per WAVE right after that wave stored its share of the workgroup's partial rows (the `nwaves` waves
// of a workgroup that store rows each call this once, every lane of them).  The wave waits for ITS OWN few write-through stores
// only and goes on to the output-tile stores; nothing else of the workgroup waits.  (The first version drew ONE ticket per
// workgroup at the very end, behind `s_waitcnt vmcnt(0)` on the whole output tile: every workgroup then held its CU slot until
// its 32 - 64 KB of stores were acknowledged instead of retiring behind them -- +0.6 ms per SlowFast step where -1.3 ms was
// expected, profiles/r5_v2_fold_ab.txt.)  The wave that draws the last ticket of the group raises `flag` (an LDS int the
// kernel cleared at its start and reads again in tail_fold()).
__device__ __forceinline__ void tail_group_ticket_wave(const TailFold& tf, int mt, int mtiles, int nt, int ntiles_n, int nwaves,
                                                       volatile int* flag) {
    SF_DRAIN_VMEM();
    if ((threadIdx.x & 63) == 0) {
        const int g = mt / tf.group;
        const int t0 = g * tf.group;
        const int t1 = t0 + tf.group < mtiles ? t0 + tf.group : mtiles;
        int32_t* counter = tf.cnt + (int64_t)(g * ntiles_n + nt) * SF_FOLD_CNT_STRIDE;
        const int t = SF_AGENT_FETCH_ADD(counter, 1);
        if (t == (t1 - t0) * nwaves - 1) {
            SF_AGENT_STORE(counter, 0);
            *flag = 1;
        }
    }
}

// Call at the very end of the producing kernel, by every thread.  `early`: the group ticket was drawn by
// tail_group_ticket_wave() (forward statistics: the rows exist before the output tile is stored); otherwise the workgroup's rows
// of `part` were just stored with SF_AGENT_STORE and the ticket is drawn here behind a full drain (backward sums: the rows are
// the LAST thing the workgroup produces).  `mt` / `nt`: the workgroup's M / column tile; `mtiles` / `ntiles_n`: tile counts;
// `ncols`: columns per tile (2 * ncols <= NT); `red`: NT doubles of LDS nobody else uses any more; `flag`: the LDS int.
template <int NT>
__device__ __forceinline__ void tail_fold(const TailFold& tf, const float* part, int C, int mt, int mtiles, int nt, int ntiles_n,
                                          int n0, int ncols, double* red, volatile int* flag, bool early) {
    const int g = mt / tf.group;
    const int t0 = g * tf.group;
    const int t1 = t0 + tf.group < mtiles ? t0 + tf.group : mtiles;
    if (early) {
        __syncthreads();                    // the ticket-drawing waves are past their atomics; everybody is done with `red`
        if (*flag == 0) return;
        __syncthreads();
    } else if (!tail_last_ticket(tf.cnt + (int64_t)(g * ntiles_n + nt) * SF_FOLD_CNT_STRIDE, t1 - t0, flag)) return;
    // ---- level 0: this group's rows -> one fp64 row
    int r0 = t0 * tf.rows_per_tile, r1 = t1 * tf.rows_per_tile;
    if (r1 > tf.nrows) r1 = tf.nrows;
    const int cols2 = 2 * ncols;
    const int cidx = (int)threadIdx.x % cols2, lane = (int)threadIdx.x / cols2;
    const int which = cidx / ncols, col = n0 + cidx % ncols;
    double tot = tail_col_sums<NT, float>(part, C, r0, r1, n0, ncols, red);
    if (tf.ngroups > 1) {
        if (lane == 0 && col < C) SF_AGENT_STORE(tf.lvl1 + ((int64_t)g * 2 + which) * C + col, tot);
        if (!tail_last_ticket(tf.cnt + (int64_t)(tf.ngroups * ntiles_n + nt) * SF_FOLD_CNT_STRIDE, tf.ngroups, flag)) return;
        // ---- level 1: the group rows of this column tile
        tot = tail_col_sums<NT, double>(tf.lvl1, C, 0, tf.ngroups, n0, ncols, red);
    }
    // ---- finalize: sum (which 0) and second sum (which 1) of a column meet through LDS
    if (lane == 0) red[cidx] = tot;
    __syncthreads();
    if ((int)threadIdx.x < ncols && n0 + (int)threadIdx.x < C)
        tail_finalize_channel(tf, C, n0 + (int)threadIdx.x, red[threadIdx.x], red[ncols + threadIdx.x]);
}
