// mfma4x4_probe.hip -- sustained rate of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per instruction: the
// shape of the mixed W gradient's lag accumulation, G[m][a] += T[m] * conj(V[a]) per frequency) against the packed
// multiply-adds fft_mixed.hip uses today.  Same launch shape, operands from memory (random), 1 / 2 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 mfma4x4_probe.hip -o mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int NACC = 24;

template <int MODE>
__global__ __launch_bounds__(256) void k(const float *in, float *out, int iters) {
    const float x0 = in[threadIdx.x], y0 = in[256 + threadIdx.x];
    float s = 0;
    if (MODE == 0) {   // 24 x 4x4x1 MFMA per iteration: 24 * 256 multiply-adds per wave
        f4 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = f4{x0, y0, x0, y0};
        float a = x0, b = y0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
            a = -a;   // (keeps the sums bounded: the operands are not loop invariant)
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {           // 48 packed multiply-adds per iteration: 48 * 128 multiply-adds per wave
        f2 acc[2 * NACC];
#pragma unroll
        for (int i = 0; i < 2 * NACC; ++i) acc[i] = f2{x0, y0};
        f2 t = {x0, y0}, v = {y0, x0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 2 * NACC; ++i)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "+v"(acc[i]) : "v"(t), "v"(v));
            t = -t;
        }
#pragma unroll
        for (int i = 0; i < 2 * NACC; ++i) s += acc[i].x + acc[i].y;
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int waves_per_simd, const float *in, float *out) {
    const int iters = 20000, blocks = 256 * waves_per_simd;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) k<MODE><<<blocks, 256>>>(in, out, iters);   // warm the clock governor
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(in, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double macs = (double)iters * (MODE == 0 ? NACC * 256.0 : 2 * NACC * 128.0) * blocks * 4;   // 4 waves per block
    printf("%-22s %d waves/SIMD: %.3f ms  %.1f TFLOP/s (2 flop per multiply-add)\n", name, waves_per_simd, ms,
           2 * macs / (ms * 1e-3) / 1e12);
}

int main() {
    float h[512];
    srand(1);
    for (float &x : h) x = (float)rand() / RAND_MAX - 0.5f;
    float *in, *out;
    hipMalloc(&in, sizeof(h));
    hipMalloc(&out, 4096);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int w : {1, 2}) {
        run<0>("mfma_f32_4x4x1", w, in, out);
        run<1>("v_pk_fma_f32", w, in, out);
    }
    return 0;
}
