// VALU issue-rate probe for gfx950: cycles per wave-instruction of the candidate inner-loop instructions of the depthwise kernels
// (v_fma_f32, v_fma_mix_f32 with fp16 operands, v_pk_fma_f32, v_cvt_f32_f16, v_dot2_f32_f16), 1 or 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 64
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    uint32_t x = 0x3c003c00u + threadIdx.x;
    float w = 1.0001f;
    float a2[2] = {1.f, 2.f};
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pa[8], pb[8], pw = {1.0001f, 0.9999f};
    for (int i = 0; i < 8; ++i) pb[i] = (f2){1.f + threadIdx.x * 1e-6f * i, 1.f - i * 1e-6f};
    for (int i = 0; i < 8; ++i) pa[i] = (f2){threadIdx.x * 0.001f + i, 1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w), "v"(w));
                if (KIND == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(x), "v"(w));
                if (KIND == 2) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "+v"(a[i]) : "v"(x), "v"(x));
                if (KIND == 3) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(x));
                if (KIND == 4) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(x));
                if (KIND == 5) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i]) : "v"(x));
                if (KIND == 6) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(w));
                if (KIND == 7) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(x));
                if (KIND == 8) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(x) : "v"(x), "v"(x));
                if (KIND == 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[i]) : "v"(pw), "v"(pw));
                if (KIND == 11) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[i]) : "v"(pw), "v"(pb[i]));
                if (KIND == 12) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[i]) : "v"(pb[(i + 3) & 7]), "v"(pb[i]));
                if (KIND == 10) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(pa[i]) : "v"(pw), "v"(pw));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 8; ++i) s += pa[i].x + pa[i].y + pb[i].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + a2[0] + x;
}
template <int KIND>
void run(const char* name, float* d, int blocks) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // one wave per SIMD when blocks == 256 (4 waves per CU), two when 512
    const double inst_per_simd = (double)iters * REP * (blocks / 256.0);
    printf("%-44s blocks %4d: %8.1f us, %5.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, blocks, ms * 1e3,
           ms * 1e-3 * 2.4e9 / inst_per_simd);
}
int main() {
    float* d;
    hipMalloc(&d, 4 << 20);
    for (int b : {256, 512, 1024}) {
        run<0>("v_fma_f32", d, b);
        run<6>("v_fmac_f32", d, b);
        run<1>("v_fma_mix_f32 (f16 lo x f32)", d, b);
        run<2>("v_fma_mix_f32 (f16 hi x f16 lo)", d, b);
        run<3>("v_cvt_f32_f16", d, b);
        run<5>("v_cvt_f32_f16_sdwa WORD_1", d, b);
        run<4>("v_dot2_f32_f16", d, b);
        run<7>("v_dot2c_f32_f16", d, b);
        run<8>("v_pk_fma_f16", d, b);
        run<9>("v_pk_fma_f32", d, b);
        run<10>("v_pk_fma_f32 (src0 lo broadcast)", d, b);
        run<11>("v_pk_fma_f32 (3 distinct pairs, 1 shared)", d, b);
        run<12>("v_pk_fma_f32 (3 distinct pairs)", d, b);
    }
    return 0;
}
