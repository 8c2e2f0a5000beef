// mfma_chain_probe.hip -- cost of DEPENDENT v_mfma_f32_32x32x16_bf16 chains with one wave per SIMD: six products into one
// accumulator back to back (what a group of the split kernel does) against the same products alternating between two
// accumulators.  Build: hipcc -O3 --offload-arch=gfx950 mfma_chain_probe.hip -o mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const u32x4 *in, float *out, int iters) {
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, in[threadIdx.x]), a1 = __builtin_bit_cast(bf16x8, in[256 + threadIdx.x]);
    const bf16x8 b0 = __builtin_bit_cast(bf16x8, in[512 + threadIdx.x]), b1 = __builtin_bit_cast(bf16x8, in[768 + threadIdx.x]);
    f32x16 d[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) d[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // four groups of six dependent products, one accumulator after the other
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int p = 0; p < 6; ++p)
                    d[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p & 1 ? a0 : a1, p & 2 ? b0 : b1, d[g], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {           // the same 24 products, alternating between two accumulators
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    d[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p & 1 ? a0 : a1, p & 2 ? b0 : b1, d[g], 0, 0, 0);
                    d[g + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p & 1 ? a0 : a1, p & 2 ? b0 : b1, d[g + 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += d[i][j];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, const u32x4 *in, float *out, int waves) {
    const int iters = 4000;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) k<MODE><<<256 * waves, 256>>>(in, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<MODE><<<256 * waves, 256>>>(in, out, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double mf = (double)iters * 24 * waves;   // MFMAs per SIMD
    printf("%-28s %d wave(s)/SIMD: %.3f ms = %.1f ns per MFMA per SIMD; %.0f TFLOP/s\n", name, waves, ms, ms * 1e6 / mf,
           mf * 1024 * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
    unsigned short h[8192];
    srand(2);
    for (auto &x : h) x = (unsigned short)(0x3f00 + (rand() & 0xff));   // bf16 in [0.5, 1)
    u32x4 *in;
    float *out;
    (void)hipMalloc(&in, sizeof(h));
    (void)hipMalloc(&out, 4096);
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int w : {1, 2}) {
        run<0>("six dependent, in turn", in, out, w);
        run<1>("two accumulators alternating", in, out, w);
    }
    return 0;
}
