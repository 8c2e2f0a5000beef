// inhibit.hip -- the lateral terms of the H half step and the fold + update of the reconstruction modes, as kernels of
// the library (the reference computes them with array arithmetic in its front end:
// tnmf/TransformInvariantNMF.py:253-269 on top of tnmf/backends/_NumPyBackend.py:56-64).
//
//   G[n,m]  = ky (*)_y kx (*)_x H[n,m]            separable 'same' convolution, zeros outside (scipy convolve1d, cval 0)
//   E[n,m]  = inh * (G[n,m] - H[n,m])  +  xc * (sum_m' G[n,m'] - G[n,m]),      xc = cross_inhibition / (M - 1)
//
// E is the extra term of the denominator of the multiplicative update: H <- H * neg / (pos + E + eps + sparsity).  One
// kernel computes it: a workgroup owns a 16 x 64 pixel tile of one sample and walks the atoms; per atom the tile with its
// halo goes through LDS once (x pass into a second LDS array, y pass into registers; every thread slides a register
// window along the convolution axis: one LDS read per 8 / 4 multiply-adds).  With cross inhibition the G of every atom
// is parked in E on the way and a second walk over the atoms (its reads hit in L2) turns it into E.
#include "generic.h"

namespace {

constexpr int kTY = 16, kTX = 64, kThreads = 256;

struct Taps2 {
    double ky[kMaxTaps];
    double kx[kMaxTaps];
};

// H, E: [N][M][Hy][ld] (the same row stride; pad columns beyond the shift width are pixels that hold zeros)
template <typename T>
__global__ __launch_bounds__(kThreads) void k_inhibition(const T *__restrict__ H, T *__restrict__ E, int M, int Hy, int ld,
                                                       Taps2 taps, int ly, int lx, T inh, T xc, int tiles_y,
                                                       int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int ry = (ly - 1) / 2, rx = (lx - 1) / 2;
    const int SH = kTY + 2 * ry, SW = kTX + 2 * rx + 8;   // (+8: the register window of the x pass reads 7 past its last tap)
    T *A0 = reinterpret_cast<T *>(smem_raw);             // [SH][SW]   tile + halo
    T *A1 = A0 + (size_t)SH * SW;                        // [SH][kTX]  after the x pass
    T *ky = A1 + (size_t)SH * kTX;                       // taps in window order: weight of window offset d is k[l-1-d]
    T *kx = ky + ly;
    unsigned bid = blockIdx.x;
    const int txi = bid % tiles_x;
    bid /= tiles_x;
    const int tyi = bid % tiles_y;
    const int n = bid / tiles_y;
    const int u0 = tyi * kTY, v0 = txi * kTX;
    for (int i = threadIdx.x; i < ly; i += kThreads) ky[i] = (T)taps.ky[ly - 1 - i];
    for (int i = threadIdx.x; i < lx; i += kThreads) kx[i] = (T)taps.kx[lx - 1 - i];
    const int col = threadIdx.x & (kTX - 1), rg = threadIdx.x / kTX;   // y pass / output: column, group of 4 rows
    const size_t plane = (size_t)Hy * ld;
    T S[4] = {T(0), T(0), T(0), T(0)};
    const bool cross = xc != T(0);
    for (int m = 0; m < M; ++m) {
        const T *h = H + ((size_t)n * M + m) * plane;
        __syncthreads();   // the previous atom's y pass is done with A1 (first time: the taps are in place)
        for (int i = threadIdx.x; i < SH * SW; i += kThreads) {
            const int r = i / SW, q = i - r * SW;
            const int y = u0 + r - ry, x = v0 + q - rx;
            A0[i] = (y >= 0 && y < Hy && x >= 0 && x < ld) ? h[(size_t)y * ld + x] : T(0);
        }
        __syncthreads();
        // x pass: item = (row, segment of 8 columns); out[c] = sum_d kx[d] * A0[row][c + d]
        for (int it = threadIdx.x; it < SH * (kTX / 8); it += kThreads) {
            const int r = it / (kTX / 8), c0 = (it - r * (kTX / 8)) * 8;
            const T *row = A0 + (size_t)r * SW + c0;
            T win[8], acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] = T(0);
                win[j] = j < 7 ? row[j] : T(0);
            }
            for (int d0 = 0; d0 < lx; d0 += 8) {
#pragma unroll
                for (int dd = 0; dd < 8; ++dd) {
                    const int d = d0 + dd;
                    if (d < lx) {
                        win[(dd + 7) & 7] = row[d + 7];
                        const T k = kx[d];
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] += k * win[(dd + j) & 7];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) A1[(size_t)r * kTX + c0 + j] = acc[j];
        }
        __syncthreads();
        // y pass: rows 4 rg .. 4 rg + 3 of column col; out[r] = sum_d ky[d] * A1[r + d][col]
        T g4[4], win[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g4[j] = T(0);
            win[j] = j < 3 ? A1[(size_t)(4 * rg + j) * kTX + col] : T(0);
        }
        for (int d0 = 0; d0 < ly; d0 += 4) {
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                const int d = d0 + dd;
                if (d < ly) {
                    const int rr = 4 * rg + d + 3;   // (< SH whenever it matters: the last window row of the last output)
                    win[(dd + 3) & 3] = A1[(size_t)(rr < SH ? rr : SH - 1) * kTX + col];
                    const T k = ky[d];
#pragma unroll
                    for (int j = 0; j < 4; ++j) g4[j] += k * win[(dd + j) & 3];
                }
            }
        }
        const int x = v0 + col;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = u0 + 4 * rg + j;
            if (u < Hy && x < ld) {
                const size_t o = ((size_t)n * M + m) * plane + (size_t)u * ld + x;
                if (cross) {
                    S[j] += g4[j];
                    E[o] = g4[j];   // parked; the second walk below turns it into E
                } else {
                    E[o] = inh * (g4[j] - A0[(size_t)(4 * rg + j + ry) * SW + col + rx]);
                }
            }
        }
    }
    if (cross) {
        const int x = v0 + col;
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int u = u0 + 4 * rg + j;
                if (u < Hy && x < ld) {
                    const size_t o = ((size_t)n * M + m) * plane + (size_t)u * ld + x;
                    const T gv = E[o];   // (this thread's own store)
                    E[o] = inh * (gv - H[o]) + xc * (S[j] - gv);
                }
            }
    }
}

// valid mode, kernel families without an extra-term epilogue: H = H * neg / (pos + E + reg); H and E with row stride ld,
// neg / pos C-contiguous
template <typename T>
__global__ void k_mu_update_extra(T *__restrict__ H, const T *__restrict__ neg, const T *__restrict__ pos,
                                  const T *__restrict__ E, size_t rows, int Hx, int ld, T reg) {
    const size_t total = rows * (size_t)Hx;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const size_t r = e / Hx;
        const size_t o = r * ld + (e - r * Hx);
        const T p = pos[e] + (E ? E[o] : T(0)) + reg;
        H[o] = (H[o] * neg[e]) / p;
    }
}

// activation index copied to padded position j of one axis (-1: a zero) and the second padded copy of activation u
// (generic.hip has the same tables for tnmf_hip_pad_H / tnmf_hip_fold_H; reference: backends/_PyTorchBackend.py:42-52)
__device__ __forceinline__ int dup_of(int u, int S, int a, int mode) {
    const int l = a - 1;
    if (mode == TNMF_MODE_CIRCULAR) return u >= S - l ? u - (S - l) : -1;
    if (mode == TNMF_MODE_REFLECT) return (u >= 1 && u <= l) ? l - u : -1;
    return -1;
}

// reconstruction modes: the gradient with respect to an activation is the sum of the gradients at the padded positions
// that copy it (fold = adjoint of the pad), then the multiplicative update -- in one pass:
//   H[u] = H[u] * sum_copies negp / (sum_copies posp + E[u] + reg)
template <typename T>
__global__ void k_fold_update(size_t rows, int Sy, int Sx, int Ay, int Ax, int Py, int Px, int mode, T *__restrict__ H,
                              const T *__restrict__ negp, const T *__restrict__ posp, const T *__restrict__ E, T reg) {
    const size_t total = rows * (size_t)Sy * Sx;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int ux = (int)(e % Sx);
        const size_t rest = e / Sx;
        const int uy = (int)(rest % Sy);
        const size_t r = rest / Sy;
        const int jy[2] = {Py == 1 ? 0 : uy + Ay - 1, Py == 1 ? -1 : dup_of(uy, Sy, Ay, mode)};
        const int jx[2] = {ux + Ax - 1, dup_of(ux, Sx, Ax, mode)};
        T an = T(0), ap = T(0);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (jy[a] >= 0 && jx[b] >= 0) {
                    const size_t o = (r * Py + jy[a]) * Px + jx[b];
                    an += negp[o];
                    ap += posp[o];
                }
        H[e] = (H[e] * an) / (ap + (E ? E[e] : T(0)) + reg);
    }
}

inline int grid_cap(size_t n, const tnmf_hip_ctx *ctx) {
    const size_t want = (n + kThreads - 1) / kThreads, cap = (size_t)ctx->num_cu * 8;
    return (int)(want < cap ? (want ? want : 1) : cap);
}

template <typename T>
int launch_inhibition_t(int N, int M, int Hy, int ld, const void *H, void *E, const Taps2 &taps, int ly, int lx, double inh,
                        double xc, hipStream_t s) {
    const int ry = (ly - 1) / 2, rx = (lx - 1) / 2;
    const size_t SH = kTY + 2 * ry, SW = kTX + 2 * rx + 8;
    const size_t lds = (SH * SW + SH * kTX + ly + lx) * sizeof(T);
    if (lds > 160 * 1024) return TNMF_E_UNSUPPORTED;
    if (lds > 64 * 1024)
        TNMF_HIP_TRY(hipFuncSetAttribute((const void *)k_inhibition<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024));
    const int tiles_y = cdiv(Hy, kTY), tiles_x = cdiv(ld, kTX);
    const size_t blocks = (size_t)N * tiles_y * tiles_x;
    if (blocks > 0x7fffffffull) return TNMF_E_GEOM;
    hipLaunchKernelGGL(k_inhibition<T>, dim3((unsigned)blocks), dim3(kThreads), lds, s, (const T *)H, (T *)E, M, Hy, ld,
                       taps, ly, lx, (T)inh, (T)xc, tiles_y, tiles_x);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

}  // namespace

int launch_inhibition(const tnmf_hip_ctx *, int dtype, int N, int M, int Hy, int ld, const void *H, void *E,
                      const double *ky_host, int ly, const double *kx_host, int lx, double inh, double xc,
                      hipStream_t s) {
    if (ly < 1 || lx < 1 || ly > kMaxTaps || lx > kMaxTaps || !(ly & 1) || !(lx & 1)) return TNMF_E_UNSUPPORTED;
    if (N <= 0) return TNMF_OK;
    Taps2 taps;
    for (int i = 0; i < kMaxTaps; ++i) {
        taps.ky[i] = i < ly ? ky_host[i] : 0.0;
        taps.kx[i] = i < lx ? kx_host[i] : 0.0;
    }
    return dtype == 0 ? launch_inhibition_t<float>(N, M, Hy, ld, H, E, taps, ly, lx, inh, xc, s)
                      : launch_inhibition_t<double>(N, M, Hy, ld, H, E, taps, ly, lx, inh, xc, s);
}

int launch_mu_update_extra(const tnmf_hip_ctx *ctx, int dtype, void *H, const void *neg, const void *pos, const void *E,
                           size_t rows, int Hx, int ld, double reg, hipStream_t s) {
    const size_t total = rows * (size_t)Hx;
    if (total == 0) return TNMF_OK;
    const int grid = grid_cap(total, ctx);
    if (dtype == 0)
        hipLaunchKernelGGL(k_mu_update_extra<float>, dim3(grid), dim3(kThreads), 0, s, (float *)H, (const float *)neg,
                           (const float *)pos, (const float *)E, rows, Hx, ld, (float)reg);
    else
        hipLaunchKernelGGL(k_mu_update_extra<double>, dim3(grid), dim3(kThreads), 0, s, (double *)H, (const double *)neg,
                           (const double *)pos, (const double *)E, rows, Hx, ld, reg);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

int launch_fold_update(const tnmf_hip_ctx *ctx, const Geo &g, int dtype, int mode, int Sy, int Sx, void *H,
                       const void *negp, const void *posp, const void *E, double reg, hipStream_t s) {
    const size_t rows = (size_t)g.N * g.M, total = rows * (size_t)Sy * Sx;
    if (total == 0) return TNMF_OK;
    const int grid = grid_cap(total, ctx);
    if (dtype == 0)
        hipLaunchKernelGGL(k_fold_update<float>, dim3(grid), dim3(kThreads), 0, s, rows, Sy, Sx, g.Ay, g.Ax, g.Hy, g.Hx, mode,
                           (float *)H, (const float *)negp, (const float *)posp, (const float *)E, (float)reg);
    else
        hipLaunchKernelGGL(k_fold_update<double>, dim3(grid), dim3(kThreads), 0, s, rows, Sy, Sx, g.Ay, g.Ax, g.Hy, g.Hx,
                           mode, (double *)H, (const double *)negp, (const double *)posp, (const double *)E, reg);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}
