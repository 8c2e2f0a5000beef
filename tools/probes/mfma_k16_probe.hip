// mfma_k16_probe.hip -- issue cost of the legacy v_mfma_f32_16x16x16_bf16 (K = 16) against v_mfma_f32_16x16x32_bf16 (K = 32)
// on gfx950, in-kernel cycles (s_memtime) per instruction on one wave per SIMD and wall time at one and two waves per SIMD:
// does half the K cost half the cycles?  (If it does, the half-empty last k block of 12 x 12 atoms -- K = 144 = 4.5 blocks
// of 32 -- can run on it.)   Build: hipcc -O3 --offload-arch=gfx950 mfma_k16_probe.hip -o mfma_k16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int K32>
__global__ __launch_bounds__(256, 1) void k(const u32x4 *in, float *out, unsigned long long *cyc, int iters) {
    const u32x4 ra = in[threadIdx.x], rb = in[256 + threadIdx.x];
    const bf16x8 a8 = __builtin_bit_cast(bf16x8, ra), b8 = __builtin_bit_cast(bf16x8, rb);
    const s16x4 a4 = __builtin_bit_cast(s16x4, u32x2{ra[0], ra[1]}), b4 = __builtin_bit_cast(s16x4, u32x2{rb[0], rb[1]});
    f32x4 d[8];
    for (int i = 0; i < 8; ++i) d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                if (K32)
                    d[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, d[g], 0, 0, 0);
                else
                    d[g] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, d[g], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) s += d[i][j];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int K32>
void run(const char *name, const u32x4 *in, float *out, unsigned long long *cyc, int waves) {
    const int iters = 4000;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) k<K32><<<256 * waves, 256>>>(in, out, cyc, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<K32><<<256 * waves, 256>>>(in, out, cyc, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long c = 0;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mf = (double)iters * 48;   // MFMAs per wave
    printf("%-28s %d wave(s)/SIMD: %.3f ms; s_memtime ticks per MFMA of one wave %.2f (100 MHz ticks x clock ratio: compare the two rows)\n", name,
           waves, ms, (double)c / mf);
}

int main() {
    unsigned short h[4096];
    srand(2);
    for (auto &x : h) x = (unsigned short)(0x3f00 + (rand() & 0xff));   // bf16 in [0.5, 1)
    u32x4 *in;
    float *out;
    unsigned long long *cyc;
    (void)hipMalloc(&in, sizeof(h));
    (void)hipMalloc(&out, 4096);
    (void)hipMalloc(&cyc, 8);
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int w : {1, 2}) {
        run<1>("16x16x32 bf16 (K = 32)", in, out, cyc, w);
        run<0>("16x16x16 bf16 (K = 16)", in, out, cyc, w);
    }
    return 0;
}
