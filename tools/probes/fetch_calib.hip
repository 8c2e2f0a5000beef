// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of the H read-modify-write.
// MI355X_MICROARCH.md (HBM): FETCH_SIZE reports half the bytes of a wide coalesced streaming read; other shapes are
// uncalibrated.  Each kernel below reads (and kernel 3 also writes) a KNOWN number of bytes from a buffer far larger
// than the 256 MiB Infinity Cache, every byte exactly once:
//   k_stream     lane i -> 16 bytes at base + 16 i: the wide coalesced stream (control)
//   k_planes32   the shape of k_mfma_corr_W_persist's H accesses (round 1): per instruction 32 planes x 2 pieces of
//                16 bytes, planes 285 KB apart, 4-byte aligned
//   k_planes8    the shape of k_split_corr_W's H accesses (round 2): per instruction 8 planes x 8 adjacent lanes x 16
//                bytes (128 contiguous bytes), 4-byte aligned, read AND written back
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/fetch_calib.hip -o tools/probes/fetch_calib
// Run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- tools/probes/fetch_calib     (and a second pass with WRITE_SIZE)
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                  \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__global__ void k_stream(const f32x4 *__restrict__ p, float *out, size_t n4) {
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}

// H[n][32 planes][Hy][Hx]: a wave takes (n, row u, 32-pixel tile); lane (atom j, h) reads pixels 8q + 4h + {0..3}, q = 0..3
__global__ void k_planes32(const float *__restrict__ H, float *out, int N, int Hy, int Hx) {
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int tiles = Hx / 32;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    f32x4 acc = {0, 0, 0, 0};
    for (size_t t = wave; t < (size_t)N * Hy * tiles; t += nw) {
        const int tx = t % tiles, u = (t / tiles) % Hy, n = t / ((size_t)tiles * Hy);
        const float *row = H + (((size_t)n * 32 + j) * Hy + u) * Hx + tx * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += *reinterpret_cast<const f32x4_u *>(row + 8 * q + 4 * h);
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}

// lane (j = l & 31, h): atom ((j >> 3) & 3) + 4h + 8q, pixels 4 (j & 7) + {0..3}: read, scale, write back
__global__ void k_planes8(float *__restrict__ H, int N, int Hy, int Hx) {
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int tiles = Hx / 32;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t t = wave; t < (size_t)N * Hy * tiles; t += nw) {
        const int tx = t % tiles, u = (t / tiles) % Hy, n = t / ((size_t)tiles * Hy);
        f32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float *p = H + (((size_t)n * 32 + ((j >> 3) & 3) + 4 * h + 8 * q) * Hy + u) * Hx + tx * 32 + 4 * (j & 7);
            v[q] = *reinterpret_cast<const f32x4_u *>(p);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float *p = H + (((size_t)n * 32 + ((j >> 3) & 3) + 4 * h + 8 * q) * Hy + u) * Hx + tx * 32 + 4 * (j & 7);
            *reinterpret_cast<f32x4_u *>(p) = v[q] * 1.0001f;
        }
    }
}

int main() {
    // config-3-like geometry, 64 samples x 32 planes x 267 rows, 256 of the columns of every row touched.  Two row
    // strides: 288 floats (every 32-pixel tile = whole 128-byte lines: the fabric traffic equals the bytes touched) and
    // 267 floats like the real H (tiles straddle lines: what the counters report then, against the aligned case, is
    // the over-fetch of partial lines, not a counter artefact).
    const int N = 64, Hy = 267;
    float *H, *out;
    const size_t elems = (size_t)N * 32 * Hy * 288;
    CK(hipMalloc(&H, elems * 4));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(H, 0, elems * 4));
    hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (const f32x4 *)H, out, elems / 4);
    CK(hipDeviceSynchronize());
    printf("k_stream reads %.0f bytes\n", (double)elems * 4);
    for (int stride : {288, 267}) {
        hipLaunchKernelGGL(k_planes32, dim3(2048), dim3(256), 0, 0, (const float *)H, out, N, Hy, stride);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_planes8, dim3(2048), dim3(256), 0, 0, H, N, Hy, stride);
        CK(hipDeviceSynchronize());
        printf("row stride %d floats: k_planes32 reads %.0f bytes; k_planes8 reads and writes %.0f bytes (tiles = %d)\n", stride,
               (double)N * 32 * Hy * (stride / 32 * 32) * 4, (double)N * 32 * Hy * (stride / 32 * 32) * 4, stride / 32);
    }
    return 0;
}
