// pkfma_probe.hip -- issue cost of v_pk_fma_f32 (with the op_sel / neg forms of the complex multiply-add) against
// plain v_fma_f32 on gfx950, at 1, 2 and 4 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 pkfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    f2 acc[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) acc[i] = f2{seed + i, seed - i};
    f2 t = {seed * 0.5f, seed * 0.25f}, v = {seed * 0.125f, 1.f - seed};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            if (MODE == 0) {   // the two packed instructions of one complex multiply-add (as in fft_mixed.hip)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "+v"(acc[i]) : "v"(t), "v"(v));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(t), "v"(v));
            } else if (MODE == 1) {   // the same arithmetic as four plain FMAs
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(t.x), "v"(v.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(t.x), "v"(v.y));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(t.y), "v"(v.y));
                asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(acc[i].y) : "v"(t.y), "v"(v.x));
            } else {   // packed, no operand selection
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(t), "v"(v));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(v), "v"(t));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 24; ++i) s += acc[i].x + acc[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int waves_per_simd) {
    float *out;
    hipMalloc(&out, 4096);
    const int iters = 20000, blocks = 256 * waves_per_simd;   // 256-thread blocks = 4 waves = one per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(out, 100, 0.3f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(out, iters, 0.3f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double cfma = (double)iters * 24 * waves_per_simd;   // complex multiply-adds per SIMD
    printf("%-28s %d waves/SIMD: %.3f ms  -> %.2f ns per complex multiply-add per SIMD (%.1f cycles at 2.4 GHz), %.1f TFLOP/s\n",
           name, waves_per_simd, ms, ms * 1e6 / cfma, ms * 1e6 / cfma * 2.4, 8.0 * 64 * cfma * 1024 / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("pk_fma op_sel (2 instr)", w);
        run<1>("v_fma_f32 (4 instr)", w);
        run<2>("pk_fma plain (2 instr)", w);
    }
    return 0;
}
