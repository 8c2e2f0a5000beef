// fft_spectral.hip -- contractions on RESIDENT full spectra of the activations.
//
// Problems the mixed kernels (fft_mixed.hip) do not cover -- several channels, atoms taller than 16 rows -- keep the
// column transforms of H in the workspace: they are computed once per H (k_fft_cols_fwd) and serve both reconstructs
// and the W gradient of an iteration, which then are plain streaming kernels with no transform, no LDS and no barrier:
//
//   R^[n,c,f]  = sum_m H^[n,m,f] * W^[m,c,f]            (reconstruct, NumPy.py:122-132 in the frequency domain)
//   G^[m,c,f]  = sum_n H^[n,m,f] * conj(V^[n,c,f])      (W gradient, NumPy.py:69-91; pos with R^)
//
// f runs over the Ly * KXP entries of a full-spectrum plane; lanes run along f.
#include "fft.h"
#include "fft_engine.h"

namespace {

constexpr int kSpecThreads = 256;
constexpr int kSpecCG = 4;   // channels accumulated per pass at most (instantiated for 1 .. 4: three channels take three)

constexpr int kSpecNS = 4;   // samples per thread in the reconstruct contraction (W^ entries loaded once for all of them)
constexpr int kSpecMS = 4;   // atoms per thread in the W-gradient contraction (V^, R^ entries loaded once for all of them)

// XCD-aware block order: the hardware deals consecutive workgroups round-robin to the eight XCDs, each with an L2 of its
// own.  Both kernels below order their logical grid so that the blocks that share the SMALL operand (W^; V^ and R^) are
// consecutive -- dealt that way they land on eight different L2s and the small operand is fetched eight times (PMC at
// config 5: 23.7 GB fetched by the W gradient for 11.5 GB of activation spectra).  Launched as 1-D grids and remapped so
// that every XCD walks a contiguous chunk of the logical order, the sharers meet in one L2.
__device__ __forceinline__ void xcd_block(int gx, int gy, int gz, int &bx, int &by, int &bz) {
    long lin = blockIdx.x;
    const long total = (long)gx * gy * gz, whole = total / 8 * 8;
    if (lin < whole) lin = (lin & 7) * (whole / 8) + (lin >> 3);
    bx = (int)(lin % gx);
    by = (int)((lin / gx) % gy);
    bz = (int)(lin / ((long)gx * gy));
}

// grid (sample quads, f in blocks of 256, channel groups): every thread owns one f of kSpecNS samples and loops the
// atoms.  The sample index runs fastest over the blocks so that the blocks in flight share the W^ entries of one f
// block (L2).
template <typename T, int CG>
__global__ __launch_bounds__(kSpecThreads) void k_spec_contract_R(const cplx<T> *SH, const cplx<T> *SW, cplx<T> *SR, int N,
                                                                int M, int C, long plane, int KX, int KXP, int gx,
                                                                int gy, int gz) {
    int bx, by, bz;
    xcd_block(gx, gy, gz, bx, by, bz);
    const long f = (long)by * kSpecThreads + threadIdx.x;
    if (f >= plane || (int)(f % KXP) >= KX) return;   // the pad columns of a spectrum row hold nothing
    const int n0 = bx * kSpecNS, c0 = bz * CG;
    cplx<T> acc[kSpecNS][CG];
#pragma unroll
    for (int i = 0; i < kSpecNS; ++i)
#pragma unroll
        for (int c = 0; c < CG; ++c) acc[i][c] = {0, 0};
    for (int m = 0; m < M; ++m) {
        cplx<T> w[CG], hv[kSpecNS];
#pragma unroll
        for (int c = 0; c < CG; ++c) w[c] = SW[((long)m * C + (c0 + c < C ? c0 + c : C - 1)) * plane + f];
#pragma unroll
        for (int i = 0; i < kSpecNS; ++i) hv[i] = SH[((long)(n0 + i < N ? n0 + i : N - 1) * M + m) * plane + f];
#pragma unroll
        for (int i = 0; i < kSpecNS; ++i)
#pragma unroll
            for (int c = 0; c < CG; ++c) cfma(acc[i][c], hv[i], w[c]);
    }
#pragma unroll
    for (int i = 0; i < kSpecNS; ++i)
#pragma unroll
        for (int c = 0; c < CG; ++c)
            if (n0 + i < N && c0 + c < C) SR[((long)(n0 + i) * C + c0 + c) * plane + f] = acc[i][c];
}

// grid (atom quads, f in blocks of 256, sample groups * channel groups): every thread owns one f of kSpecMS atoms and
// sums over the samples of its group; partial sums [group][M*C][plane], added up in group order afterwards.  The atom
// index runs fastest over the blocks so that the blocks in flight share the V^, R^ entries of one f block (L2).
template <typename T, int CG>
__global__ __launch_bounds__(kSpecThreads) void k_spec_grad_W(const cplx<T> *SH, const cplx<T> *SV, const cplx<T> *SR,
                                                            cplx<T> *Gn, cplx<T> *Gp, int N, int M, int C, long plane,
                                                            int ngroups, int nper, int KX, int KXP, int gx, int gy,
                                                            int gz) {
    int bx, by, bz;
    xcd_block(gx, gy, gz, bx, by, bz);
    const long f = (long)by * kSpecThreads + threadIdx.x;
    if (f >= plane || (int)(f % KXP) >= KX) return;
    const int m0 = bx * kSpecMS, grp = bz % ngroups, c0 = (bz / ngroups) * CG;
    cplx<T> an[kSpecMS][CG], ap[kSpecMS][CG];
#pragma unroll
    for (int i = 0; i < kSpecMS; ++i)
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            an[i][c] = {0, 0};
            ap[i][c] = {0, 0};
        }
    const int nbeg = grp * nper, nend = nbeg + nper < N ? nbeg + nper : N;
    for (int n = nbeg; n < nend; ++n) {
        cplx<T> v[CG], r[CG], hv[kSpecMS];
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            const long o = ((long)n * C + (c0 + c < C ? c0 + c : C - 1)) * plane + f;
            v[c] = SV[o];
            r[c] = SR[o];
        }
#pragma unroll
        for (int i = 0; i < kSpecMS; ++i) hv[i] = SH[((long)n * M + (m0 + i < M ? m0 + i : M - 1)) * plane + f];
#pragma unroll
        for (int i = 0; i < kSpecMS; ++i)
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                cfmac(an[i][c], hv[i], v[c]);
                cfmac(ap[i][c], hv[i], r[c]);
            }
    }
    const long gsize = (long)M * C * plane;
#pragma unroll
    for (int i = 0; i < kSpecMS; ++i)
#pragma unroll
        for (int c = 0; c < CG; ++c)
            if (m0 + i < M && c0 + c < C) {
                const long o = (long)grp * gsize + ((long)(m0 + i) * C + c0 + c) * plane + f;
                Gn[o] = an[i][c];
                Gp[o] = ap[i][c];
            }
}

}  // namespace

int spectral_contract_R(const Geo &g, int dtype, const void *SH, const void *SW, void *SR, int Ly, int KX, int KXP,
                        hipStream_t s) {
    const long plane = (long)Ly * KXP;
    const int cg = g.C < kSpecCG ? g.C : kSpecCG;
    const int gx = cdiv(g.N, kSpecNS), gy = (int)((plane + kSpecThreads - 1) / kSpecThreads), gz = cdiv(g.C, cg);
    if ((long)gx * gy * gz > 0x7fffffffL) return TNMF_E_GEOM;
    const dim3 grid((unsigned)((long)gx * gy * gz));
#define SPEC_R(T_, CG_)                                                                                              \
    hipLaunchKernelGGL((k_spec_contract_R<T_, CG_>), grid, dim3(kSpecThreads), 0, s, (const cplx<T_> *)SH,           \
                       (const cplx<T_> *)SW, (cplx<T_> *)SR, g.N, g.M, g.C, plane, KX, KXP, gx, gy, gz)
#define SPEC_R_T(T_)                 \
    switch (cg) {                    \
        case 1: SPEC_R(T_, 1); break; \
        case 2: SPEC_R(T_, 2); break; \
        case 3: SPEC_R(T_, 3); break; \
        default: SPEC_R(T_, 4); break; \
    }
    if (dtype == 0) {
        SPEC_R_T(float)
    } else {
        SPEC_R_T(double)
    }
#undef SPEC_R_T
#undef SPEC_R
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

int spectral_grad_W(const Geo &g, int dtype, const void *SH, const void *SV, const void *SR, void *Gn, void *Gp, int Ly,
                    int KX, int KXP, int ngroups, int nper, hipStream_t s) {
    const long plane = (long)Ly * KXP;
    const int cg = g.C < kSpecCG ? g.C : kSpecCG;
    const int gx = cdiv(g.M, kSpecMS), gy = (int)((plane + kSpecThreads - 1) / kSpecThreads),
              gz = ngroups * cdiv(g.C, cg);
    if ((long)gx * gy * gz > 0x7fffffffL) return TNMF_E_GEOM;
    const dim3 grid((unsigned)((long)gx * gy * gz));
#define SPEC_G(T_, CG_)                                                                                              \
    hipLaunchKernelGGL((k_spec_grad_W<T_, CG_>), grid, dim3(kSpecThreads), 0, s, (const cplx<T_> *)SH,               \
                       (const cplx<T_> *)SV, (const cplx<T_> *)SR, (cplx<T_> *)Gn, (cplx<T_> *)Gp, g.N, g.M, g.C,     \
                       plane, ngroups, nper, KX, KXP, gx, gy, gz)
#define SPEC_G_T(T_)                 \
    switch (cg) {                    \
        case 1: SPEC_G(T_, 1); break; \
        case 2: SPEC_G(T_, 2); break; \
        case 3: SPEC_G(T_, 3); break; \
        default: SPEC_G(T_, 4); break; \
    }
    if (dtype == 0) {
        SPEC_G_T(float)
    } else {
        SPEC_G_T(double)
    }
#undef SPEC_G_T
#undef SPEC_G
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}
