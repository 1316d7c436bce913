// Host functional simulator for the HIP kernels in slowfast_amd/csrc  (TEST INFRASTRUCTURE ONLY).
//
// This header shadows <hip/hip_runtime.h> when the kernel sources are compiled as plain host C++
// (tests/hostsim/build_sim.py).  It executes a HIP launch on the CPU: every thread of a block is a
// ucontext fiber, __syncthreads()/wave-level operations are cooperative barriers, and the gfx950
// builtins the kernels use (MFMA 16x16x32 f16, ds_read_b64_tr_b16, shuffles) are emulated from their
// documented lane -> element maps.  It lets `pytest -m "not gpu"` exercise the REAL kernel index
// math, tile/launch selection and C-ABI on a machine without a GPU.  It is never loaded by the
// product package: slowfast_amd/lib.py only ever opens the hipcc-built library and fails loudly
// when it is missing.
#pragma once
#include <ucontext.h>
#include <sys/mman.h>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define SF_HOSTSIM 1

// ---------------------------------------------------------------- HIP surface (subset)
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "hostsim"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)

namespace hipsim {

constexpr int kWave = 64;
constexpr size_t kStack = 256 * 1024;

struct Barrier { int arrived = 0; int gen = 0; };

struct Fiber {
    ucontext_t ctx;
    void* stack = nullptr;
    bool done = false;
    dim3 tid;
};

struct WaveScratch {
    alignas(16) unsigned char a[kWave][16];
    alignas(16) unsigned char b[kWave][16];
    const void* ptr[kWave];
    unsigned long long u[kWave];
    Barrier bar;
    int nlanes = kWave;
};

// Fiber stacks are recycled through a process-wide pool: launch() starts fresh worker threads for every kernel launch,
// and a worker's BlockCtx (thread_local) dies with its thread -- without the pool every launch leaked its mmap'ed stacks
// (tens of GB of resident memory over a test session).
struct StackPool {
    std::mutex mu;
    std::vector<void*> free_list;
    void* get() {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!free_list.empty()) { void* s = free_list.back(); free_list.pop_back(); return s; }
        }
        void* s = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (s == MAP_FAILED) { perror("mmap"); abort(); }
        return s;
    }
    void put(void* s) { std::lock_guard<std::mutex> lk(mu); free_list.push_back(s); }
};
inline StackPool& stack_pool() { static StackPool* p = new StackPool(); return *p; }   // never destroyed: threads may outlive main

struct BlockCtx {
    std::vector<Fiber> fibers;
    std::vector<WaveScratch> waves;
    Barrier block_bar;
    ucontext_t main_ctx;
    int nthreads = 0;
    int cur = -1;
    const std::function<void()>* body = nullptr;
    ~BlockCtx() {
        for (Fiber& f : fibers)
            if (f.stack) stack_pool().put(f.stack);
    }
};

inline thread_local BlockCtx* g_ctx = nullptr;
inline thread_local dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

inline void yield_fiber() {
    BlockCtx* c = g_ctx;
    swapcontext(&c->fibers[c->cur].ctx, &c->main_ctx);
}

inline void barrier_wait(Barrier& b, int n) {
    int g = b.gen;
    if (++b.arrived == n) { b.arrived = 0; b.gen++; }
    else { while (b.gen == g) yield_fiber(); }
}

inline int lane_id() { return g_ctx->cur % kWave; }
inline WaveScratch& wave() { return g_ctx->waves[g_ctx->cur / kWave]; }
inline void wave_sync() { WaveScratch& w = wave(); barrier_wait(w.bar, w.nlanes); }

inline void fiber_entry() {
    BlockCtx* c = g_ctx;
    (*c->body)();
    c->fibers[c->cur].done = true;
    swapcontext(&c->fibers[c->cur].ctx, &c->main_ctx);
}

inline void run_block(BlockCtx& c, dim3 block, dim3 bidx, dim3 grid, const std::function<void()>& body) {
    int n = block.x * block.y * block.z;
    if ((int)c.fibers.size() < n) {
        size_t old = c.fibers.size();
        c.fibers.resize(n);
        for (size_t i = old; i < (size_t)n; ++i) c.fibers[i].stack = stack_pool().get();
    }
    c.nthreads = n;
    c.body = &body;
    c.block_bar = Barrier();
    int nw = (n + kWave - 1) / kWave;
    c.waves.assign(nw, WaveScratch());
    for (int w = 0; w < nw; ++w) c.waves[w].nlanes = std::min(kWave, n - w * kWave);
    g_ctx = &c;
    g_blockIdx = bidx; g_blockDim = block; g_gridDim = grid;
    for (int i = 0; i < n; ++i) {
        Fiber& f = c.fibers[i];
        f.done = false;
        f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = &c.main_ctx;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    }
    int alive = n;
    while (alive > 0) {
        for (int i = 0; i < n; ++i) {
            Fiber& f = c.fibers[i];
            if (f.done) continue;
            c.cur = i;
            g_threadIdx = f.tid;
            swapcontext(&c.main_ctx, &f.ctx);
            if (f.done) --alive;
        }
    }
    g_ctx = nullptr;
}

inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    unsigned hw = std::thread::hardware_concurrency();
    const char* env = getenv("SF_SIM_THREADS");
    if (env) hw = (unsigned)atoi(env);
    if (hw < 1) hw = 1;
    size_t nthr = std::min<size_t>(hw, nblocks);
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        static thread_local BlockCtx ctx;
        for (;;) {
            size_t b = next.fetch_add(1);
            if (b >= nblocks) break;
            dim3 bidx(b % grid.x, (b / grid.x) % grid.y, b / ((size_t)grid.x * grid.y));
            run_block(ctx, block, bidx, grid, body);
        }
    };
    if (nthr <= 1) { worker(); return; }
    std::vector<std::thread> th;
    for (size_t i = 0; i < nthr; ++i) th.emplace_back(worker);
    for (auto& t : th) t.join();
}

}  // namespace hipsim

#define threadIdx (hipsim::g_threadIdx)
#define blockIdx (hipsim::g_blockIdx)
#define blockDim (hipsim::g_blockDim)
#define gridDim (hipsim::g_gridDim)

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    hipsim::launch((grid), (block), [=]() { kern(__VA_ARGS__); })

inline void __syncthreads() { hipsim::barrier_wait(hipsim::g_ctx->block_bar, hipsim::g_ctx->nthreads); }

// ---------------------------------------------------------------- atomics
inline float atomicAdd(float* p, float v) {
    unsigned* pu = reinterpret_cast<unsigned*>(p);
    unsigned old = __atomic_load_n(pu, __ATOMIC_RELAXED);
    for (;;) {
        float f; memcpy(&f, &old, 4); f += v;
        unsigned nu; memcpy(&nu, &f, 4);
        if (__atomic_compare_exchange_n(pu, &old, nu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
            float r; memcpy(&r, &old, 4); return r;
        }
    }
}
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

// ---------------------------------------------------------------- wave-level emulation
template <typename T>
inline T __shfl_xor(T v, int mask) {
    static_assert(sizeof(T) <= 8, "shfl payload");
    hipsim::WaveScratch& w = hipsim::wave();
    int l = hipsim::lane_id();
    unsigned long long u = 0; memcpy(&u, &v, sizeof(T));
    w.u[l] = u;
    hipsim::wave_sync();
    int src = l ^ mask;
    if (src >= w.nlanes) src = l;
    unsigned long long r = w.u[src];
    hipsim::wave_sync();
    T out; memcpy(&out, &r, sizeof(T));
    return out;
}
template <typename T>
inline T __shfl_down(T v, int delta) {
    hipsim::WaveScratch& w = hipsim::wave();
    int l = hipsim::lane_id();
    unsigned long long u = 0; memcpy(&u, &v, sizeof(T));
    w.u[l] = u;
    hipsim::wave_sync();
    int src = l + delta;
    if (src >= w.nlanes) src = l;
    unsigned long long r = w.u[src];
    hipsim::wave_sync();
    T out; memcpy(&out, &r, sizeof(T));
    return out;
}

// wave vote: non-zero when the predicate holds on any lane
inline int __any(int pred) {
    hipsim::WaveScratch& w = hipsim::wave();
    int l = hipsim::lane_id();
    w.u[l] = pred ? 1ull : 0ull;
    hipsim::wave_sync();
    int r = 0;
    for (int i = 0; i < w.nlanes; ++i) r |= (int)w.u[i];
    hipsim::wave_sync();
    return r;
}
#define SF_EXP2(x) exp2f(x)
#define SF_MAX3(a, b, c) fmaxf(fmaxf((a), (b)), (c))

#ifdef SF_ACT_BF16
typedef __bf16 hipsim_f16;      // the library's 16-bit storage type (see csrc/sf_common.h)
#else
typedef _Float16 hipsim_f16;
#endif
typedef hipsim_f16 hipsim_f16x8 __attribute__((ext_vector_type(8)));
typedef hipsim_f16 hipsim_f16x4 __attribute__((ext_vector_type(4)));
typedef float hipsim_f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x32_f16: D(16x16) = A(16x32) * B(32x16) + C.
// lane l holds A[i = l&15][k = 8*(l>>4) + e], B[k = 8*(l>>4) + e][j = l&15], e = 0..7;
// C/D: lane l, reg r -> row 4*(l>>4) + r, col l&15.
inline hipsim_f32x4 hipsim_mfma_16x16x32_f16(hipsim_f16x8 a, hipsim_f16x8 b, hipsim_f32x4 c) {
    hipsim::WaveScratch& w = hipsim::wave();
    int l = hipsim::lane_id();
    memcpy(w.a[l], &a, 16);
    memcpy(w.b[l], &b, 16);
    hipsim::wave_sync();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r;
        float acc = 0.f;
        for (int k = 0; k < 32; ++k) {
            hipsim_f16 av, bv;
            memcpy(&av, &w.a[row + 16 * (k >> 3)][2 * (k & 7)], 2);
            memcpy(&bv, &w.b[col + 16 * (k >> 3)][2 * (k & 7)], 2);
            acc += (float)av * (float)bv;
        }
        c[r] += acc;
    }
    hipsim::wave_sync();
    return c;
}
#define SF_MFMA16(a, b, c) hipsim_mfma_16x16x32_f16((a), (b), (c))

// ds_read_b64_tr_b16: per 16-lane group, lane p supplies the address of 4 contiguous b16
// (row p/4, columns 4*(p%4)..+3 of a 4x16 block); lane q receives column q: element j comes from
// the lane (4*j + q/4) of its group, that lane's element (q%4).
inline hipsim_f16x4 hipsim_ds_read_tr16(const void* p) {
    hipsim::WaveScratch& w = hipsim::wave();
    int l = hipsim::lane_id();
    w.ptr[l] = p;
    hipsim::wave_sync();
    int g = l & ~15, q = l & 15;
    hipsim_f16x4 out;
    for (int j = 0; j < 4; ++j) {
        const hipsim_f16* src = (const hipsim_f16*)w.ptr[g + 4 * j + (q >> 2)];
        out[j] = src[q & 3];
    }
    hipsim::wave_sync();
    return out;
}
#define SF_LDS_TR16(p) hipsim_ds_read_tr16((const void*)(p))

// global_load_lds_dwordx4: every lane fetches 16 bytes from its own global address; the wave's 1 KiB lands in LDS
// at (wave-uniform base = lane 0's destination operand) + 16 * lane.
inline void hipsim_global_load_lds16(const void* gptr, void* lds_base) {
    hipsim::WaveScratch& w = hipsim::wave();
    int l = hipsim::lane_id();
    w.ptr[l] = lds_base;
    hipsim::wave_sync();
    unsigned char* base = (unsigned char*)w.ptr[0];
    hipsim::wave_sync();
    memcpy(base + 16 * l, gptr, 16);
}
// the same with some lanes masked off (EXEC): they take part in the rendezvous, their 16 bytes of LDS stay as they are
inline void hipsim_global_load_lds16_if(bool on, const void* gptr, void* lds_base) {
    hipsim::WaveScratch& w = hipsim::wave();
    int l = hipsim::lane_id();
    w.ptr[l] = lds_base;
    hipsim::wave_sync();
    unsigned char* base = (unsigned char*)w.ptr[0];
    hipsim::wave_sync();
    if (on) memcpy(base + 16 * l, gptr, 16);
}
#define SF_GLOBAL_LOAD_LDS16_SADDR_IF(on, base, voff, l) \
    hipsim_global_load_lds16_if((on), (const void*)((const char*)(base) + ((on) ? (uint32_t)(voff) : 0u)), (void*)(l))
#define SF_GLOBAL_LOAD_LDS16(g, l) hipsim_global_load_lds16((const void*)(g), (void*)(l))
#define SF_GLOBAL_LOAD_LDS16_ASM(g, l) hipsim_global_load_lds16((const void*)(g), (void*)(l))
#define SF_GLOBAL_LOAD_LDS16_SADDR(base, voff, l) hipsim_global_load_lds16((const void*)((const char*)(base) + (uint32_t)(voff)), (void*)(l))
// a wave's own vmcnt wait makes its LDS-DMA data visible to all of ITS lanes (no workgroup barrier needed for a wave that reads
// only what it copied itself): the emulated copies above finish with a per-lane memcpy, so the wait is a wave rendezvous here
#define SF_WAIT_VMEM() hipsim::wave_sync()
#define SF_CONSUME_V(x) ((void)0)
#define SF_WAVE_LDS_SYNC() hipsim::wave_sync()
#define SF_WAIT_VMEM_N(N) hipsim::wave_sync()
#define SF_BARRIER_KEEP_VMEM() __syncthreads()
#define SF_KEEP_ALIVE(x) ((void)(x))
#define SF_SCALAR_PTR(T, p) ((const T*)(p))

inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
