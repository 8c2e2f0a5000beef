// fft_engine.h -- batched mixed-radix complex FFT on a tile held in LDS (or, for the host check, in plain memory).
//
// Used by fft.hip, the FFT formulation of the three primitives (the reference's counterpart is
// backends/NumPy_FFT.py:16-40, which calls scipy.fft -- this is an independent implementation, nothing is shared).
//
// A tile is x[pos * BS + b]: NB independent sequences b of length L = R1*R2*R3, position-major, so that the 16
// batch lanes of one position are contiguous.  The forward transform is decimation in frequency, in place, and
// leaves the spectrum in digit-reversed order (pos_of_k); the inverse is decimation in time from that order back
// to natural order, unscaled.  Pointwise products do not care about the order, so no reordering pass exists.
//
// Everything here is __host__ __device__: tests/native/fft_engine_check.cpp runs the very same stage functions on
// the CPU, task by task, against a naive DFT.
#pragma once

#if defined(__HIPCC__)
#define TNMF_HD __host__ __device__ __forceinline__
#else
#define TNMF_HD inline
#endif

template <typename T>
struct cplx {
    T x, y;
};

template <typename T>
TNMF_HD cplx<T> cadd(cplx<T> a, cplx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T>
TNMF_HD cplx<T> csub(cplx<T> a, cplx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T>
TNMF_HD cplx<T> cmul(cplx<T> a, cplx<T> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
// a * conj(b)
template <typename T>
TNMF_HD cplx<T> cmulc(cplx<T> a, cplx<T> b) { return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }
// acc += a * b
template <typename T>
TNMF_HD void cfma(cplx<T> &acc, cplx<T> a, cplx<T> b) {
    acc.x += a.x * b.x - a.y * b.y;
    acc.y += a.x * b.y + a.y * b.x;
}
// acc += a * conj(b)
template <typename T>
TNMF_HD void cfmac(cplx<T> &acc, cplx<T> a, cplx<T> b) {
    acc.x += a.x * b.x + a.y * b.y;
    acc.y += a.y * b.x - a.x * b.y;
}

template <int N>
TNMF_HD constexpr double tw_cos(int j);
template <int N>
TNMF_HD constexpr double tw_sin(int j);
#include "fft_twiddles.h"

// ---- forward DFTs of length R on registers: v[k] <- sum_n v[n] exp(-2 pi i n k / R) ----------------------------
template <typename T, int R>
struct Dft;

template <typename T>
struct Dft<T, 2> {
    static TNMF_HD void run(cplx<T> *v) {
        const cplx<T> a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

template <typename T>
struct Dft<T, 3> {
    static TNMF_HD void run(cplx<T> *v) {
        const T s = (T)0.86602540378443864676;
        const cplx<T> a = v[0], t1 = cadd(v[1], v[2]), d = csub(v[1], v[2]);
        const cplx<T> t2 = {a.x - (T)0.5 * t1.x, a.y - (T)0.5 * t1.y};
        v[0] = cadd(a, t1);
        v[1] = {t2.x + s * d.y, t2.y - s * d.x};
        v[2] = {t2.x - s * d.y, t2.y + s * d.x};
    }
};

template <typename T>
struct Dft<T, 4> {
    static TNMF_HD void run(cplx<T> *v) {
        const cplx<T> s02 = cadd(v[0], v[2]), d02 = csub(v[0], v[2]);
        const cplx<T> s13 = cadd(v[1], v[3]), d13 = csub(v[1], v[3]);
        v[0] = cadd(s02, s13);
        v[2] = csub(s02, s13);
        v[1] = {d02.x + d13.y, d02.y - d13.x};   // d02 - i d13
        v[3] = {d02.x - d13.y, d02.y + d13.x};   // d02 + i d13
    }
};

template <typename T>
struct Dft<T, 5> {
    static TNMF_HD void run(cplx<T> *v) {
        const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;   // cos(2 pi / 5), cos(4 pi / 5)
        const T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;    // sin(2 pi / 5), sin(4 pi / 5)
        const cplx<T> a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]), b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
        const cplx<T> t1 = {v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y};
        const cplx<T> t2 = {v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y};
        const cplx<T> u1 = {s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y};
        const cplx<T> u2 = {s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y};
        v[0] = cadd(v[0], cadd(a1, a2));
        v[1] = {t1.x + u1.y, t1.y - u1.x};   // t1 - i u1
        v[4] = {t1.x - u1.y, t1.y + u1.x};   // t1 + i u1
        v[2] = {t2.x + u2.y, t2.y - u2.x};
        v[3] = {t2.x - u2.y, t2.y + u2.x};
    }
};

// N = RA*RB by one Cooley-Tukey split with compile-time twiddles (loops unroll, constants fold)
template <typename T, int RA, int RB>
struct DftComposite {
    static TNMF_HD void run(cplx<T> *v) {
        constexpr int N = RA * RB;
        cplx<T> t[RB][RA];
#pragma unroll
        for (int n2 = 0; n2 < RB; ++n2) {
            cplx<T> u[RA];
#pragma unroll
            for (int n1 = 0; n1 < RA; ++n1) u[n1] = v[RB * n1 + n2];
            Dft<T, RA>::run(u);
#pragma unroll
            for (int k1 = 0; k1 < RA; ++k1) {
                const int j = n2 * k1;
                if (j == 0) {
                    t[n2][k1] = u[k1];
                } else if (4 * j == N) {
                    t[n2][k1] = {u[k1].y, -u[k1].x};   // * (-i)
                } else if (2 * j == N) {
                    t[n2][k1] = {-u[k1].x, -u[k1].y};
                } else if (4 * j == 3 * N) {
                    t[n2][k1] = {-u[k1].y, u[k1].x};   // * i
                } else {
                    const cplx<T> w = {(T)tw_cos<N>(j), (T)(-tw_sin<N>(j))};
                    t[n2][k1] = cmul(u[k1], w);
                }
            }
        }
#pragma unroll
        for (int k1 = 0; k1 < RA; ++k1) {
            cplx<T> u[RB];
#pragma unroll
            for (int n2 = 0; n2 < RB; ++n2) u[n2] = t[n2][k1];
            Dft<T, RB>::run(u);
#pragma unroll
            for (int k2 = 0; k2 < RB; ++k2) v[k1 + RA * k2] = u[k2];
        }
    }
};

template <typename T>
struct Dft<T, 6> : DftComposite<T, 2, 3> {};
template <typename T>
struct Dft<T, 8> : DftComposite<T, 2, 4> {};
template <typename T>
struct Dft<T, 9> : DftComposite<T, 3, 3> {};
template <typename T>
struct Dft<T, 10> : DftComposite<T, 2, 5> {};
template <typename T>
struct Dft<T, 12> : DftComposite<T, 3, 4> {};
template <typename T>
struct Dft<T, 16> : DftComposite<T, 4, 4> {};

// inverse (unscaled) through the swap identity  IDFT(x) = swap(DFT(swap(x))),  swap(a + ib) = b + ia
template <typename T, int R>
TNMF_HD void idft(cplx<T> *v) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const T t = v[i].x;
        v[i].x = v[i].y;
        v[i].y = t;
    }
    Dft<T, R>::run(v);
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const T t = v[i].x;
        v[i].x = v[i].y;
        v[i].y = t;
    }
}

// ---- three-stage plan over a tile ------------------------------------------------------------------------------
// tw[t] = exp(-2 pi i t / L), t in [0, L).  x points at (pos 0, this sequence); BS is the position stride.
template <typename T, int L_, int R1_, int R2_, int R3_>
struct FftPlan {
    static constexpr int L = L_, R1 = R1_, R2 = R2_, R3 = R3_;
    static constexpr int M1 = L / R1, M2 = M1 / R2;
    static_assert(R1 * R2 * R3 == L && M2 == R3, "radices must multiply to L");
    static constexpr int tasks1 = M1, tasks2 = R1 * M2, tasks3 = R1 * R2;

    // where frequency k sits after the forward transform / which frequency sits at pos
    static TNMF_HD int pos_of_k(int k) { return (k % R1) * M1 + ((k / R1) % R2) * M2 + k / (R1 * R2); }
    static TNMF_HD int k_of_pos(int pos) {
        const int p1 = pos / M1, r = pos - p1 * M1, p2 = r / M2, p3 = r - p2 * M2;
        return p1 + R1 * (p2 + R2 * p3);
    }

    // forward, decimation in frequency
    template <int BS>
    static TNMF_HD void fwd1(cplx<T> *x, const cplx<T> *tw, int j) {
        cplx<T> v[R1];
#pragma unroll
        for (int q = 0; q < R1; ++q) v[q] = x[(j + M1 * q) * BS];
        Dft<T, R1>::run(v);
#pragma unroll
        for (int p = 1; p < R1; ++p) v[p] = cmul(v[p], tw[j * p]);
#pragma unroll
        for (int p = 0; p < R1; ++p) x[(j + M1 * p) * BS] = v[p];
    }
    template <int BS>
    static TNMF_HD void fwd2(cplx<T> *x, const cplx<T> *tw, int task) {
        const int b = task / M2, j = task - b * M2;
        cplx<T> *xb = x + (b * M1 + j) * BS;
        cplx<T> v[R2];
#pragma unroll
        for (int q = 0; q < R2; ++q) v[q] = xb[M2 * q * BS];
        Dft<T, R2>::run(v);
#pragma unroll
        for (int p = 1; p < R2; ++p) v[p] = cmul(v[p], tw[R1 * j * p]);
#pragma unroll
        for (int p = 0; p < R2; ++p) xb[M2 * p * BS] = v[p];
    }
    template <int BS>
    static TNMF_HD void fwd3(cplx<T> *x, int blk) {
        cplx<T> *xb = x + blk * R3 * BS;
        cplx<T> v[R3];
#pragma unroll
        for (int q = 0; q < R3; ++q) v[q] = xb[q * BS];
        Dft<T, R3>::run(v);
#pragma unroll
        for (int p = 0; p < R3; ++p) xb[p * BS] = v[p];
    }

    // inverse, decimation in time: inv3, inv2, inv1 in this order
    template <int BS>
    static TNMF_HD void inv3(cplx<T> *x, int blk) {
        cplx<T> *xb = x + blk * R3 * BS;
        cplx<T> v[R3];
#pragma unroll
        for (int q = 0; q < R3; ++q) v[q] = xb[q * BS];
        idft<T, R3>(v);
#pragma unroll
        for (int p = 0; p < R3; ++p) xb[p * BS] = v[p];
    }
    template <int BS>
    static TNMF_HD void inv2(cplx<T> *x, const cplx<T> *tw, int task) {
        const int b = task / M2, j = task - b * M2;
        cplx<T> *xb = x + (b * M1 + j) * BS;
        cplx<T> v[R2];
#pragma unroll
        for (int p = 0; p < R2; ++p) v[p] = xb[M2 * p * BS];
#pragma unroll
        for (int p = 1; p < R2; ++p) v[p] = cmulc(v[p], tw[R1 * j * p]);
        idft<T, R2>(v);
#pragma unroll
        for (int q = 0; q < R2; ++q) xb[M2 * q * BS] = v[q];
    }
    template <int BS>
    static TNMF_HD void inv1(cplx<T> *x, const cplx<T> *tw, int j) {
        cplx<T> v[R1];
#pragma unroll
        for (int p = 0; p < R1; ++p) v[p] = x[(j + M1 * p) * BS];
#pragma unroll
        for (int p = 1; p < R1; ++p) v[p] = cmulc(v[p], tw[j * p]);
        idft<T, R1>(v);
#pragma unroll
        for (int q = 0; q < R1; ++q) x[(j + M1 * q) * BS] = v[q];
    }
};

// The supported lengths and their radices (kept short: every length instantiates every kernel of fft.hip).
template <typename T, int L>
struct FftPlanFor;
#define TNMF_FFT_PLAN(L, R1, R2, R3)               \
    template <typename T>                          \
    struct FftPlanFor<T, L> : FftPlan<T, L, R1, R2, R3> {}
TNMF_FFT_PLAN(32, 4, 4, 2);
TNMF_FFT_PLAN(48, 4, 4, 3);
TNMF_FFT_PLAN(64, 4, 4, 4);
TNMF_FFT_PLAN(96, 4, 4, 6);
TNMF_FFT_PLAN(144, 4, 6, 6);
TNMF_FFT_PLAN(192, 4, 6, 8);
TNMF_FFT_PLAN(270, 5, 6, 9);   // (row direction only: 136 stored frequencies = 8.5 segments of 16 where 288 has 145 = 9.06)
TNMF_FFT_PLAN(288, 6, 6, 8);
TNMF_FFT_PLAN(384, 6, 8, 8);
TNMF_FFT_PLAN(540, 6, 9, 10);  // (row direction only: 271 stored frequencies = 17 segments where 576 has 289 = 18.06)
TNMF_FFT_PLAN(576, 8, 8, 9);
#undef TNMF_FFT_PLAN

// Two real sequences a, b transformed as z = a + i b.  After the forward transform:
//   A[k] = (Z[k] + conj(Z[L-k])) / 2,   B[k] = -i (Z[k] - conj(Z[L-k])) / 2.
template <typename T>
TNMF_HD void split_pair(cplx<T> z1, cplx<T> z2, cplx<T> &a, cplx<T> &b) {
    a = {(T)0.5 * (z1.x + z2.x), (T)0.5 * (z1.y - z2.y)};
    b = {(T)0.5 * (z1.y + z2.y), (T)0.5 * (z2.x - z1.x)};
}
// Inverse direction: Z[k] = A[k] + i B[k],  Z[L-k] = conj(A[k]) + i conj(B[k]).
template <typename T>
TNMF_HD void merge_pair(cplx<T> a, cplx<T> b, cplx<T> &zk, cplx<T> &zlk) {
    zk = {a.x - b.y, a.y + b.x};
    zlk = {a.x + b.y, b.x - a.y};
}
