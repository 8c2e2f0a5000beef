// read_probe.hip -- read-only streaming rate of the box: every thread sums 16-byte loads of a 2.5 GB array (far beyond the
// Infinity Cache), U loads in flight per thread, W waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 read_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void k_read(const f4 *src, long n4, float *out) {
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    for (; i < n4; i += stride) acc += src[i];
    const float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 123.456f) out[0] = s;
}

// every workgroup sweeps a private contiguous chunk (as many concurrent streams as workgroups) instead of the grid-stride
// sweep above (the whole chip inside one moving window of a few MB)
template <int U>
__global__ __launch_bounds__(256) void k_read_chunks(const f4 *src, long n4, float *out) {
    const long per = n4 / gridDim.x;
    const f4 *p = src + (long)blockIdx.x * per;
    f4 acc = {0, 0, 0, 0};
    for (long i = threadIdx.x; i + (U - 1) * 256 < per; i += U * 256) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    const float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 123.456f) out[0] = s;
}

template <int U>
void run_chunks(const f4 *src, long n4, float *out, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    k_read_chunks<U><<<blocks, 256>>>(src, n4, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; ++r) k_read_chunks<U><<<blocks, 256>>>(src, n4, out);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    printf("private chunks: U=%d, %2d workgroups per CU (%d streams): %.3f ms  %.2f TB/s\n", U, blocks_per_cu, blocks, ms,
           n4 * 16.0 / ms / 1e9);
}

// 8-byte loads (the row-spectrum entries of the mixed kernels are complex64), U in flight, and F dependent-free packed
// multiply-adds per loaded value: the instruction mix of k_mix_reconstruct (27 loads, 384 v_pk_fma_f32 per atom)
typedef float f2 __attribute__((ext_vector_type(2)));
template <int U, int F>
__global__ __launch_bounds__(128) void k_read8(const f2 *src, long n2, float *out) {
    const long per = n2 / gridDim.x;
    const f2 *p = src + (long)blockIdx.x * per;
    f2 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f2{0, 0};
    for (long i = threadIdx.x; i + (U - 1) * 128 < per; i += U * 128) {
        f2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * 128];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int f = 0; f < F; ++f)
                asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc[(u + f) & 7]) : "v"(v[u]));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    if (s == 123.456f) out[0] = s;
}

template <int U, int F>
void run8(const f4 *src, long n4, float *out, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    k_read8<U, F><<<blocks, 128>>>((const f2 *)src, 2 * n4, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; ++r) k_read8<U, F><<<blocks, 128>>>((const f2 *)src, 2 * n4, out);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    printf("8-byte loads: U=%2d in flight, %2d packed FMAs per value, %2d workgroups of 2 waves per CU: %.3f ms  %.2f TB/s\n", U, F,
           blocks_per_cu, ms, n4 * 16.0 / ms / 1e9);
}

template <int U>
void run(const f4 *src, long n4, float *out, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    k_read<U><<<blocks, 256>>>(src, n4, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; ++r) k_read<U><<<blocks, 256>>>(src, n4, out);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    printf("U=%d loads in flight, %2d workgroups per CU: %.3f ms  %.2f TB/s\n", U, blocks_per_cu, ms, n4 * 16.0 / ms / 1e9);
}

int main() {
    const long n4 = 160L * 1024 * 1024;   // 2.5 GB
    f4 *src;
    float *out;
    (void)hipMalloc(&src, n4 * 16);
    (void)hipMalloc(&out, 64);
    (void)hipMemset(src, 0, n4 * 16);
    for (int bpc : {2, 4, 8}) {
        run<2>(src, n4, out, bpc);
        run<4>(src, n4, out, bpc);
        run<8>(src, n4, out, bpc);
    }
    for (int bpc : {4, 8}) {
        run8<4, 2>(src, n4, out, bpc);
        run8<16, 2>(src, n4, out, bpc);
        run8<27, 2>(src, n4, out, bpc);
        run8<27, 7>(src, n4, out, bpc);
        run8<27, 14>(src, n4, out, bpc);
    }
    for (int bpc : {2, 4, 8}) {
        run_chunks<2>(src, n4, out, bpc);
        run_chunks<4>(src, n4, out, bpc);
    }
    return 0;
}
