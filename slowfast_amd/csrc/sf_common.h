// Shared device/host helpers for the gfx950 (CDNA4, wave64) kernels of slowfast_amd.
//
// Data model: activations live in HBM as fp16, channels-last -- logical (N,C,T,H,W)
// tensors whose memory order is N,T,H,W,C ("position rows" of C channels, row pitch `ld` elements so
// channel-slice views work).  All contractions run on v_mfma_f32_16x16x32_{f16,bf16} with fp32 accumulation;
// BatchNorm statistics, affine parameters and weight gradients are fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// `f16` is THE 16-bit storage type of activations, packed weights and MFMA operands throughout the library: IEEE half in the
// default build (libsfamd.so), bfloat16 when compiled with -DSF_ACT_BF16 (libsfamd_bf16.so, SF_ACT_DTYPE=bf16 on the Python
// side; sf_act_dtype() tells which).  Everything else -- accumulators, statistics, parameter gradients -- is fp32 in both.
#ifdef SF_ACT_BF16
typedef __bf16 f16;
#define SF_ACT_DTYPE_ID 1
#ifndef SF_MFMA16
#define SF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif
#else
typedef _Float16 f16;
#define SF_ACT_DTYPE_ID 0
#ifndef SF_MFMA16
#define SF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#endif
#endif
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(8))) u32x2 { uint32_t x, y; };
typedef __fp16 fp16x4_raw __attribute__((__vector_size__(4 * sizeof(__fp16))));

// LDS transpose read (ds_read_b64_tr_b16).  The host functional simulator (tests/hostsim) pre-defines
// this hook; on gfx950 it is the builtin.
#ifndef SF_LDS_TR16
#define SF_LDS_TR16(p) \
    __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_raw*)(p))
#endif

// Direct global -> LDS copy (global_load_lds_dwordx4): each lane supplies its own 16-byte global source, the wave's
// 64 x 16 B land contiguously at the wave-uniform LDS base (+ 16 * lane).  SF_WAIT_VMEM() retires them.
#ifndef SF_GLOBAL_LOAD_LDS16
#define SF_GLOBAL_LOAD_LDS16(g, l)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g),                \
                                     (__attribute__((address_space(3))) void*)(l), 16, 0, 0)
#define SF_WAIT_VMEM()                                                          \
    do {                                                                        \
        __builtin_amdgcn_sched_barrier(0); /* keep the MFMAs of the step above the wait */ \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        \
    } while (0)
#endif
// The same copy as an INLINE-ASM statement.  hipcc keeps its own books on the builtin above: every LDS read it cannot prove
// disjoint from the copy's destination -- in particular the ds_read_b64_tr_b16 intrinsic, which carries no address
// information -- gets an `s_waitcnt vmcnt(0)` in front of it, which drains the copies of the NEXT stage before the current one
// is computed (the three-stage ring of the weight-gradient kernels then runs as load -> wait -> compute, serially).  An asm
// copy is absent from that bookkeeping; its completion is counted by the kernel's own SF_WAIT_VMEM_N + barrier.  M0 (the
// LDS-DMA destination base) is compiler-reserved: saved and restored inside the statement; the LDS byte address must be
// wave-uniform.
#ifndef SF_GLOBAL_LOAD_LDS16_ASM
#ifdef SF_GLDS_KEEP_M0
#define SF_GLOBAL_LOAD_LDS16_ASM(g, l)                                                                               \
    do {                                                                                                                \
        unsigned sf_keep_m0_;                                                                                           \
        const unsigned sf_lds_addr_ = (unsigned)__builtin_amdgcn_readfirstlane(                                         \
            (int)(uintptr_t)(__attribute__((address_space(3))) void*)(l));                                              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(sf_keep_m0_)                                                                               \
                     : "v"((const void*)(g)), "s"(sf_lds_addr_)                                                         \
                     : "memory");                                                                                       \
    } while (0)
#else
// default: M0 is written and left (nothing else in these kernels reads it: no builtin LDS-DMA, no s_movrel / s_sendmsg between
// the copies; hipcc itself never assumes a value in M0 across an asm statement) -- two scalar instructions fewer per copy
#define SF_GLOBAL_LOAD_LDS16_ASM(g, l)                                                                               \
    do {                                                                                                                \
        const unsigned sf_lds_addr_ = (unsigned)__builtin_amdgcn_readfirstlane(                                         \
            (int)(uintptr_t)(__attribute__((address_space(3))) void*)(l));                                              \
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"                                 \
                     :                                                                                                  \
                     : "v"((const void*)(g)), "s"(sf_lds_addr_)                                                         \
                     : "memory");                                                                                       \
    } while (0)
#endif
#endif
// The same copy with the source given as a WAVE-UNIFORM base (scalar register pair) plus a 32-bit per-lane byte offset
// (`global_load_lds_dwordx4 voff, s[base]`): one VGPR per copy slot instead of a 64-bit per-lane pointer, and the per-stage
// address update is scalar arithmetic on the base.  (Round 5: the loop-invariant 64-bit lane pointers hipcc hoisted out of the
// attention kernels' chunk loops were what it spilled; their reloads drained the copy pipeline, see sf_attn.h.)
#ifndef SF_GLOBAL_LOAD_LDS16_SADDR
#define SF_GLOBAL_LOAD_LDS16_SADDR(base, voff, l)                                                                    \
    do {                                                                                                                \
        const unsigned sf_lds_addr_ = (unsigned)__builtin_amdgcn_readfirstlane(                                         \
            (int)(uintptr_t)(__attribute__((address_space(3))) void*)(l));                                              \
        const uint64_t sf_b_ = (uint64_t)(base);                                                                        \
        const uint64_t sf_bu_ = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(sf_b_ >> 32)) << 32) |       \
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sf_b_);                         \
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                                  \
                     :                                                                                                  \
                     : "v"((uint32_t)(voff)), "s"(sf_bu_), "s"(sf_lds_addr_)                                            \
                     : "memory");                                                                                       \
    } while (0)
#endif
// ... and with lanes masked off: their 16 bytes of LDS keep what they held (a pre-zeroed halo); `on` may diverge inside the wave
#ifndef SF_GLOBAL_LOAD_LDS16_SADDR_IF
#define SF_GLOBAL_LOAD_LDS16_SADDR_IF(on, base, voff, l)                \
    do {                                                                \
        if (on) SF_GLOBAL_LOAD_LDS16_SADDR(base, voff, l);              \
    } while (0)
#endif
// LDS hand-over between the lanes of ONE wave (ds_write by some lanes, ds_read of the same bytes by others): the hardware runs
// a wave's LDS instructions in order, so only the compiler must be kept from reordering them; the host simulator, whose lanes
// are fibers, needs a real rendezvous here
#ifndef SF_WAVE_LDS_SYNC
#define SF_WAVE_LDS_SYNC() __builtin_amdgcn_wave_barrier()
#endif
// make hipcc treat a (vector-register) value as read and rewritten here: loads that produced it are waited for at this point
#ifndef SF_CONSUME_V
#define SF_CONSUME_V(x) asm volatile("" : "+v"(x))
#endif
// keeps a fragment register live without using it (diagnostic ablations only)
#ifndef SF_KEEP_ALIVE
#define SF_KEEP_ALIVE(x) asm volatile("" ::"v"(x))
#endif
// wait until at most N (compile-time) of this wave's vector-memory operations are outstanding: retires everything but
// the newest N, i.e. a whole copy stage while the next one stays in flight
#ifndef SF_WAIT_VMEM_N
#define SF_WAIT_VMEM_N(N)                                                       \
    do {                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                      \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");                \
    } while (0)
#endif

// raw workgroup barrier that does NOT drain the direct-to-LDS copies in flight (__syncthreads() would: its fence waits
// vmcnt(0)).  LDS reads of this wave are retired first (WAR: another wave may overwrite the stage after the barrier).
#ifndef SF_BARRIER_KEEP_VMEM
#define SF_BARRIER_KEEP_VMEM()                                          \
    do {                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              \
        __builtin_amdgcn_s_barrier();                                   \
        asm volatile("" ::: "memory");                                  \
    } while (0)
#endif

// Diagnostic ablation switches of the hot kernels (Igemm2Params::ablate, AttnParams::ablate: parts of a kernel switched off,
// results are garbage) exist only in -DSF_DIAG builds (`python -m slowfast_amd.build_ext --diag` -> libsfamd_diag.so, used by
// the tools/ sweeps through SFAMD_LIBRARY).  In the product build the tests fold to constants: no branch, no register, and a stray
// environment variable cannot corrupt a training run.
#ifdef SF_DIAG
#define SF_ABLATE(p) ((p).ablate)
#else
#define SF_ABLATE(p) 0
#endif

// Read-only table read through the SCALAR cache: a wave-uniform index into constant-address-space memory becomes s_load
// (lgkmcnt), which keeps it out of the vector-memory queue whose counted waits pace the direct-to-LDS copies (an ordinary
// global_load beside them makes hipcc wait vmcnt(0) at its first use and drains the pipeline).  The table must have been
// written by an EARLIER kernel.
#ifndef SF_SCALAR_PTR
#define SF_SCALAR_PTR(T, p) ((const __attribute__((address_space(4))) T*)(p))
#endif

// 2^x on the transcendental unit (v_exp_f32: -inf -> 0, no range fix-ups)
// max of three floats as ONE v_max3_f32: fmaxf(fmaxf(a, b), c) compiles (IEEE mode) to a canonicalising v_max_f32 x, x per operand
// in front of the two maxima.  NaN operands are ignored exactly like fmaxf does.  (Host simulator: plain fmaxf.)
#ifndef SF_MAX3
__device__ __forceinline__ float sf_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#define SF_MAX3(a, b, c) sf_max3(a, b, c)
#endif
#ifndef SF_EXP2
#define SF_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif

// GELU (exact, erf) and its derivative (nn.GELU, slowfast/models/common.py:17)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_df(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

#define SF_WAVE 64
#define SF_THREADS 256

// ---------------------------------------------------------------------------------------------
// Division by a runtime-constant divisor (host computes the magic): q = (umulhi(x, mul) + x) >> shr,
// exact for 0 <= x < 2^31 and 1 <= d < 2^31.
struct FastDiv {
    uint32_t d, mul, shr;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d ? d : 1;
    uint32_t s = 0;
    while ((1ull << s) < f.d) ++s;
    uint64_t m = ((1ull << (32 + s)) + f.d - 1) / f.d - (1ull << 32);
    f.mul = (uint32_t)m;
    f.shr = s;
    return f;
}
__device__ __forceinline__ uint32_t fd_div(uint32_t x, const FastDiv& f) {
    uint32_t hi = (uint32_t)(((uint64_t)x * f.mul) >> 32);
    return (hi + x) >> f.shr;
}
__device__ __forceinline__ void fd_divmod(uint32_t x, const FastDiv& f, uint32_t& q, uint32_t& r) {
    q = fd_div(x, f);
    r = x - q * f.d;
}

// 64 bytes of zeros: the source of every direct-to-LDS copy that stands for padding (the module's own constant: the callee
// allocates nothing)
__device__ __attribute__((aligned(64))) const uint32_t sf_zero_line[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// ---------------------------------------------------------------------------------------------
// fp32 side rows of a token residual stream (MultiScaleBlock's residual sums, attention.py:500-510):
// rows m with m % period == 0 (period 1: every row) carry an fp32 copy at side row m / period, pitch ld floats.  The
// class-token row of MViT is the only row the classifier reads, and 32 fp16 roundings of it (two residual sums per block) were
// the largest single term of the logits' deviation from the fp32 reference (profiles/r3/r3_mvit_logits_bisect.md).
struct F32Rows {
    const float* in;    // optional residual operand rows (nullptr: the kernel's 16-bit residual operand is used)
    float* out;         // result rows; nullptr = feature off
    int ld;
    FastDiv fd;         // period
};
// (is row m a side row, its side-row index)
__device__ __forceinline__ bool f32_row(const F32Rows& f, int m, uint32_t& s) {
    uint32_t rem;
    fd_divmod((uint32_t)m, f.fd, s, rem);
    return rem == 0u;
}

// Side rows of a GEMM tile, taken from the fp32 accumulators BEFORE the 16-bit staging of the epilogue: out = acc (+ bias,
// already applied) + residual (fp32 side rows when given, else the 16-bit residual operand); the accumulator is replaced by the
// sum, so the 16-bit row the store loop writes is round(out) and the loop must not add the residual again on these rows.
template <int TM, int TN>
__device__ __forceinline__ void f32_rows_epilogue(f32x4 (&acc)[TM][TN], const F32Rows& f, int row0, int col0, int M, int Nout,
                                                  const f16* resid, int ldr, int resid_row0) {
    const int lane = threadIdx.x & 63;
    {   // Wave-uniform early-out: no side row among the 16 * TM rows of this wave's tile (with one class token per ~1600 rows, true
        // for 5 tiles in 6).  Without it every wave walked TM * 4 * TN exec-masked blocks -- a compare, a saveexec and a branch
        // each, ~300 instructions per tile -- to find all of them empty.
        uint32_t q, rem;
        fd_divmod((uint32_t)__builtin_amdgcn_readfirstlane(row0), f.fd, q, rem);
        const uint32_t first = (uint32_t)row0 + (rem ? f.fd.d - rem : 0u);        // first side row >= row0
        if (first >= (uint32_t)(row0 + 16 * TM) || first >= (uint32_t)M) return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = row0 + i * 16 + 4 * (lane >> 4) + r;
            uint32_t s;
            if (m >= M || !f32_row(f, m, s)) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = col0 + j * 16 + (lane & 15);
                if (col >= Nout) continue;
                float rr = 0.f;
                if (f.in) rr = f.in[(int64_t)s * f.ld + col];
                else if (resid && m >= resid_row0) rr = (float)resid[(int64_t)m * ldr + col];
                const float o = acc[i][j][r] + rr;
                f.out[(int64_t)s * f.ld + col] = o;
                acc[i][j][r] = o;
            }
        }
}

// ---------------------------------------------------------------------------------------------
// XCD-aware workgroup order.  Workgroup b runs on XCD b % 8 (observed dispatch order, MI355X_MICROARCH.md); each XCD
// has its own L2.  Remapping the linear id so that every XCD walks a CONTIGUOUS range of tiles keeps the tiles that
// share an operand panel (same rows, neighbouring columns / same split) behind one L2 instead of eight.  Bijective for
// any grid size; affects speed only.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t total) {
    const uint32_t q = total >> 3, r = total & 7u;
    const uint32_t xcd = bid & 7u, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// ---------------------------------------------------------------------------------------------
// 16-byte global/LDS moves.
__device__ __forceinline__ f16x8 ld16(const f16* p) { return *reinterpret_cast<const f16x8*>(p); }
__device__ __forceinline__ void st16(f16* p, f16x8 v) { *reinterpret_cast<f16x8*>(p) = v; }
__device__ __forceinline__ f16x8 zero8() {
    f16x8 z = {(f16)0, (f16)0, (f16)0, (f16)0, (f16)0, (f16)0, (f16)0, (f16)0};
    return z;
}

// ---------------------------------------------------------------------------------------------
// Gathered ("im2col on the fly") operand of the implicit GEMMs.  Rows are positions of the ROW
// space (fwd/wgrad: conv output positions; dgrad: conv input positions), columns are the flattened
// (tap, channel) index k = tap*C + c with tap = (kt*kH + kh)*kW + kw.  The element at (row, k) is
// src[position(row, tap)][c] or 0 when the tap falls outside the SOURCE tensor (zero padding) --
// optionally passed through the producer's BatchNorm(+ReLU) on the fly (per-channel scale/shift).
struct GatherSide {
    const f16* src;
    int ld;                 // row pitch of src in elements
    int C;                  // channels per tap on the K axis (multiple of 8)
    int sT, sH, sW;         // SOURCE spatial dims
    int kT, kH, kW;
    int strT, strH, strW;
    int padT, padH, padW;
    int dilT, dilH, dilW;
    int mode;               // 0: rows are conv outputs (src pos = out*str - pad + tap*dil)
                            // 1: rows are conv inputs  (src pos = (in + pad - tap*dil)/str, exact)
    int Ktot;               // taps*C
    FastDiv fdC, fdkW, fdkH;      // k -> tap, tap -> (kt, kh, kw)
    FastDiv fdrW, fdrH, fdrT;     // row -> (n, a, b, c) in the ROW space
    FastDiv fdsT, fdsH, fdsW;     // dgrad stride divisibility
    FastDiv fdHW;                 // pointwise fast path: row -> (n*T + t, h*W + w)
    int rowT;                     // T extent of the ROW space
    const float* scale;     // optional on-the-fly BN of the source (nullptr = none), length C
    const float* shift;
    int relu;
};

struct RowPos {
    int n;
    int bt, bh, bw;   // mode 0: out*str - pad ; mode 1: in + pad
    bool valid;
};

__device__ __forceinline__ RowPos decode_row(const GatherSide& g, uint32_t row, bool valid) {
    RowPos r;
    uint32_t q, w, h, t, n;
    fd_divmod(row, g.fdrW, q, w);
    fd_divmod(q, g.fdrH, q, h);
    fd_divmod(q, g.fdrT, n, t);
    r.n = (int)n;
    if (g.mode == 0) {
        r.bt = (int)t * g.strT - g.padT;
        r.bh = (int)h * g.strH - g.padH;
        r.bw = (int)w * g.strW - g.padW;
    } else {
        r.bt = (int)t + g.padT;
        r.bh = (int)h + g.padH;
        r.bw = (int)w + g.padW;
    }
    r.valid = valid;
    return r;
}

// Source offset (in elements) of (row, k-group starting at column k0); returns false when the group
// is padding (outside the source, beyond Ktot, or an inactive row).
__device__ __forceinline__ bool gather_offset(const GatherSide& g, const RowPos& r, uint32_t k0, int64_t& off,
                                              uint32_t& c0) {
    if (!r.valid || k0 >= (uint32_t)g.Ktot) return false;
    uint32_t tap, kt, kh, kw, q;
    fd_divmod(k0, g.fdC, tap, c0);
    fd_divmod(tap, g.fdkW, q, kw);
    fd_divmod(q, g.fdkH, kt, kh);
    int t, h, w;
    if (g.mode == 0) {
        t = r.bt + (int)kt * g.dilT;
        h = r.bh + (int)kh * g.dilH;
        w = r.bw + (int)kw * g.dilW;
        if ((unsigned)t >= (unsigned)g.sT || (unsigned)h >= (unsigned)g.sH || (unsigned)w >= (unsigned)g.sW) return false;
    } else {
        int ut = r.bt - (int)kt * g.dilT, uh = r.bh - (int)kh * g.dilH, uw = r.bw - (int)kw * g.dilW;
        if (ut < 0 || uh < 0 || uw < 0) return false;
        uint32_t qt, rt, qh, rh, qw, rw;
        fd_divmod((uint32_t)ut, g.fdsT, qt, rt);
        fd_divmod((uint32_t)uh, g.fdsH, qh, rh);
        fd_divmod((uint32_t)uw, g.fdsW, qw, rw);
        if (rt | rh | rw) return false;
        if (qt >= (uint32_t)g.sT || qh >= (uint32_t)g.sH || qw >= (uint32_t)g.sW) return false;
        t = (int)qt; h = (int)qh; w = (int)qw;
    }
    off = ((((int64_t)r.n * g.sT + t) * g.sH + h) * g.sW + w) * (int64_t)g.ld + c0;
    return true;
}

// Pointwise fast path (kH = kW = 1, all strides 1, no H/W padding -- the 1x1x1 and (kT,1,1) convolutions, 60 % of
// the SlowFast MACs): a row and its source positions share (n, h, w), only t moves with the tap, so the gather is
// one division by C per K-step instead of the full tap decomposition.  RowPos reuse: bt = t -/+ padT, bh = h*W+w.
__device__ __forceinline__ RowPos decode_row_pw(const GatherSide& g, uint32_t row, bool valid) {
    RowPos r;
    uint32_t q, hw, n, t;
    fd_divmod(row, g.fdHW, q, hw);
    fd_divmod(q, g.fdrT, n, t);
    r.n = (int)n;
    r.bt = g.mode == 0 ? (int)t - g.padT : (int)t + g.padT;
    r.bh = (int)hw;
    r.bw = 0;
    r.valid = valid;
    return r;
}
__device__ __forceinline__ bool gather_offset_pw(const GatherSide& g, const RowPos& r, uint32_t k0, int64_t& off,
                                                 uint32_t& c0) {
    if (!r.valid || k0 >= (uint32_t)g.Ktot) return false;
    uint32_t kt;
    fd_divmod(k0, g.fdC, kt, c0);
    const int t = g.mode == 0 ? r.bt + (int)kt * g.dilT : r.bt - (int)kt * g.dilT;
    if ((unsigned)t >= (unsigned)g.sT) return false;
    off = (((int64_t)r.n * g.sT + t) * (int64_t)g.fdHW.d + r.bh) * (int64_t)g.ld + c0;
    return true;
}

// y = max(lo, x*scale + shift) on 8 consecutive channels (lo = 0: ReLU, -inf: none), fp32 math, tables in LDS read as
// four 16-byte vectors
__device__ __forceinline__ f16x8 bn_act8(f16x8 v, const float* sc, const float* sh, float lo) {
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
    const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh), h1 = *reinterpret_cast<const f32x4*>(sh + 4);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o[e] = (f16)((float)v[e] * s0[e] + h0[e]);
        o[e + 4] = (f16)((float)v[e + 4] * s1[e] + h1[e]);
    }
    // ReLU after the fp16 rounding (rounding is monotonic and keeps the sign, so max(round(x), 0) == round(max(x, 0))):
    // four packed fp16 max instead of eight fp32 ones
    if (lo == 0.f) o = __builtin_elementwise_max(o, zero8());
    return o;
}

// y = max(0, x*scale + shift) on 8 consecutive channels, fp32 math, tables in LDS.
__device__ __forceinline__ f16x8 bn_relu8(f16x8 v, const float* sc, const float* sh, int relu) {
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float x = (float)v[e] * sc[e] + sh[e];
        if (relu) x = x > 0.f ? x : 0.f;
        o[e] = (f16)x;
    }
    return o;
}

// ---------------------------------------------------------------------------------------------
// LDS operand tile [rows][32] fp16 (64-byte rows = four 16-byte slots).  The slot is rotated by
// 2*((row>>2)&3) so that each ds_read_b128 lane group of the MFMA fragment read (16 rows x one slot)
// lands on 16 distinct 16-byte bank slots (MI355X_MICROARCH.md, LDS table).
__device__ __forceinline__ int lds_tile_off(int row, int slot) {
    return row * 32 + (((slot + 2 * ((row >> 2) & 3)) & 3) << 3);
}

__device__ __forceinline__ float wave_sum_over_row_groups(float v) {
    // sums the 4 lanes {l, l^16, l^32, l^48} (the 4 row groups of a 16x16 MFMA accumulator column)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
