// Probe for DESIGN.md section 7: f32-grade products on the bf16 matrix cores.
//
// Every f32 operand is split exactly into three bf16 terms x = hi + mid + lo (8 + 8 + 8 significand bits) and one f32
// MFMA is replaced by the six bf16 MFMAs with i + j <= 2 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid), accumulated
// in f32.  This program measures, for D[32x32] = A[32xK] * B[Kx32] per wave on non-negative random data (like V, R, W):
//   * the error of that scheme and of the plain f32 MFMA against a double reference,
//   * the rate of both inner loops with operands coming from L1/L2 (so: an upper bound on what a kernel could reach).
// Build and run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/probes/bf16x3_probe.hip -o gpurun_out/bf16x3_probe && gpurun_out/bf16x3_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                  \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__device__ __forceinline__ void split8(const float *x, bf16x8 &h, bf16x8 &m, bf16x8 &l) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 hh = (__bf16)x[i];
        const float r1 = x[i] - (float)hh;      // exact
        const __bf16 mm = (__bf16)r1;
        const float r2 = r1 - (float)mm;        // exact
        h[i] = hh;
        m[i] = mm;
        l[i] = (__bf16)r2;
    }
}

// A: [tiles][32 rows][K], B: [tiles][32 cols][K] (k contiguous), D: [tiles][32][32].  One wave per tile; REP repeats the
// K loop (timing).  MODE 0: f32 MFMA 32x32x2;  MODE 1: six bf16 MFMAs 32x32x16 per K block of 16, operands split in
// the loop;  MODE 2: as 1, but the splits are done once per K block pair and reused by REP (timing of the MFMAs alone).
template <int MODE>
__global__ __launch_bounds__(256) void k_gemm(const float *A, const float *B, float *D, int K, int rep) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * 4 + wave;
    const float *a = A + tile * 32 * K, *b = B + tile * 32 * K;
    const int j = lane & 31, h = lane >> 5;
    f32x16 acc = {0};
    for (int r = 0; r < (MODE < 2 ? rep : 0); ++r) {
        if (MODE == 0) {
            for (int k = 0; k < K; k += 2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j * K + k + h], b[j * K + k + h], acc, 0, 0, 0);
        } else {
            for (int k = 0; k < K; k += 16) {
                float xa[8], xb[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    xa[i] = a[j * K + k + 8 * h + i];
                    xb[i] = b[j * K + k + 8 * h + i];
                }
                bf16x8 ah, am, al, bh, bm, bl;
                split8(xa, ah, am, al);
                split8(xb, bh, bm, bl);
                // smallest terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            }
        }
    }
    if (MODE >= 2) {
        // matrix-pipe rate alone: operands of K = 64 held in registers (split once), the MFMAs repeated `rep` times
        float xa[4][8], xb[4][8];
        bf16x8 ah[4], am[4], al[4], bh[4], bm[4], bl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xa[q][i] = a[j * K + 16 * q + 8 * h + i];
                xb[q][i] = b[j * K + 16 * q + 8 * h + i];
            }
            split8(xa[q], ah[q], am[q], al[q]);
            split8(xb[q], bh[q], bm[q], bl[q]);
        }
        for (int r = 0; r < rep * 18; ++r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (MODE == 2) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[q], bh[q], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[q], bl[q], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[q], bm[q], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[q], bh[q], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[q], bm[q], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[q], bh[q], acc, 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[q][i], xb[q][i], acc, 0, 0, 0);
                }
            }
        }
    }
    // D row = (r & 3) + 8 (r >> 2) + 4 h, col = j
#pragma unroll
    for (int r = 0; r < 16; ++r) D[tile * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = acc[r];
}

int main() {
    const int K = 1152, tiles = 256 * 4 * 8;   // 8 blocks per CU
    std::vector<float> A((size_t)tiles * 32 * K), B((size_t)tiles * 32 * K);
    srand(5);
    for (auto &v : A) v = rand() / (float)RAND_MAX;
    for (auto &v : B) v = rand() / (float)RAND_MAX * 0.01f;
    float *dA, *dB, *dD;
    CK(hipMalloc(&dA, A.size() * 4));
    CK(hipMalloc(&dB, B.size() * 4));
    CK(hipMalloc(&dD, (size_t)tiles * 1024 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> D((size_t)tiles * 1024);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        // accuracy (rep = 1), tiles 0..3 against double
        if (mode == 0)
            hipLaunchKernelGGL(k_gemm<0>, dim3(tiles / 4), dim3(256), 0, 0, dA, dB, dD, K, 1);
        else
            hipLaunchKernelGGL(k_gemm<1>, dim3(tiles / 4), dim3(256), 0, 0, dA, dB, dD, K, 1);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, rms = 0;
        long cnt = 0;
        for (int t = 0; t < 4; ++t)
            for (int i = 0; i < 32; ++i)
                for (int jj = 0; jj < 32; ++jj) {
                    double ref = 0;
                    for (int k = 0; k < K; ++k) ref += (double)A[((size_t)t * 32 + i) * K + k] * B[((size_t)t * 32 + jj) * K + k];
                    const double rel = fabs(D[(size_t)t * 1024 + i * 32 + jj] - ref) / ref;
                    worst = fmax(worst, rel);
                    rms += rel * rel;
                    ++cnt;
                }
        // timing
        const int rep = 20;
        CK(hipEventRecord(e0));
        if (mode == 0)
            hipLaunchKernelGGL(k_gemm<0>, dim3(tiles / 4), dim3(256), 0, 0, dA, dB, dD, K, rep);
        else
            hipLaunchKernelGGL(k_gemm<1>, dim3(tiles / 4), dim3(256), 0, 0, dA, dB, dD, K, rep);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = 2.0 * tiles * 32 * 32 * K * rep;
        printf("%-28s max rel err %.3e  rms %.3e   %.3f ms  = %.1f TFLOP/s (f32-equivalent)\n",
               mode == 0 ? "f32 MFMA 32x32x2" : "3 x bf16 split, 6 MFMAs", worst, sqrt(rms / cnt), ms, flops / ms / 1e9);
    }
    // matrix-pipe rates with register-resident operands: K = 64 per pass, rep * 18 passes
    for (int mode = 2; mode < 4; ++mode) {
        const int rep = 20;
        CK(hipEventRecord(e0));
        if (mode == 2)
            hipLaunchKernelGGL(k_gemm<2>, dim3(tiles / 4), dim3(256), 0, 0, dA, dB, dD, K, rep);
        else
            hipLaunchKernelGGL(k_gemm<3>, dim3(tiles / 4), dim3(256), 0, 0, dA, dB, dD, K, rep);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = 2.0 * tiles * 32 * 32 * 64.0 * rep * 18;
        printf("%-28s operands in registers: %.3f ms  = %.1f TFLOP/s (f32-equivalent)\n",
               mode == 3 ? "f32 MFMA 32x32x2" : "3 x bf16 split, 6 MFMAs", ms, flops / ms / 1e9);
    }
    return 0;
}
