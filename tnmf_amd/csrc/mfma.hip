// mfma.hip -- float32 kernels of the shift-invariant MU path on the gfx950 matrix cores.
//
// All three primitives are written as implicit GEMMs on v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 (exact f32:
// a k-ordered fmaf chain, cdna_hip_programming.md section 3), operands staged through LDS, no im2col in memory.
//
//   corr_W  (H gradient, NumPy.py:101-119)   D[atom m][pixel v]        K = (c, a, b)      A = W          B = X window
//   corr_H  (W gradient, NumPy.py:77-90)     D[atom m][shift (c,a',b')] K = H pixel (r,t)  A = H          B = shifted X
//   reconstruct (NumPy.py:122-132)           D[(c, a)][pixel t]        K = (m, b)         A = flipped W  B = H window
//                                            followed by an in-register/LDS "col2im" along the row axis.
//
// Operand maps used (lane l of the wave64):
//   32x32x2 : A[i = l&31][k = l>>5]   B[k = l>>5][j = l&31]   D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31], r = 0..15
//   16x16x4 : A[i = l&15][k = l>>4]   B[k = l>>4][j = l&15]   D[row = 4*(l>>4) + r][col = l&15],           r = 0..3
#include <cstdio>
#include <cstdlib>

#include "mfma.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access at 4-byte alignment

namespace {

constexpr int kBlock = 256;   // 4 waves

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory, i.e. it waits for every
// outstanding global load/store of the wave (vmcnt(0)): prefetches and epilogue stores that are meant to stay in flight
// across the barrier would be drained there.  Here only the LDS queue is drained; "memory" keeps hipcc from moving
// memory operations across the barrier.
// diagnostic builds of the timing harness only (env TNMF_HIP_STAMPS): cycle stamp that also drains the LDS queue
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define STAMP(k_)                                   \
    do {                                            \
        if (dbg) {                                  \
            const unsigned long long t_ = stamp();  \
            phase[k_] += t_ - tprev;                \
            tprev = t_;                             \
        }                                           \
    } while (0)
// XCD-aware workgroup numbering (speed only, never correctness): workgroups are dealt round-robin over the 8 XCDs, so
// ids b and b+8 share an L2.  The remap gives every XCD a contiguous chunk of the logical grid, so that logically
// adjacent workgroups (neighbouring column blocks of one sample: shared halo, same DRAM pages) run on one XCD at about
// the same time.  Bijective for any grid size (cdna_hip_programming.md, T1).
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ================================================================================================================
// corr_W: one workgroup = 32 atoms x (CW_TY rows x 32 cols) of the shift plane of one sample.
//   wave w owns rows w*CW_RB .. w*CW_RB+CW_RB-1 of the tile, for V and for R: 2*CW_RB accumulators of 32x32.
//   per channel: zero-padded V and R windows (row stride CW_XSTR) and W[32 atoms][Ay][Axp] (Ax padded to even with
//   zero taps, so that the two k of one MFMA sit in the same atom row) are staged in LDS.
// ================================================================================================================
constexpr int CW_TY = 16, CW_TX = 32, CW_RB = 4, CW_XSTR = 64;

template <bool FUSED>
__global__ __launch_bounds__(kBlock, 2) void k_mfma_corr_W(Geo g, int tiles_y, int tiles_x, int MT, int ablate,
                                                        const float *__restrict__ V, const float *__restrict__ Rr,
                                                        const float *__restrict__ W, float *__restrict__ Hio,
                                                        float *__restrict__ neg, float *__restrict__ pos, float reg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Axp = (g.Ax + 1) & ~1;
    const int SH = CW_TY + g.Ay - 1;
    const int KC = g.Ay * Axp;
    float2 *Xs = reinterpret_cast<float2 *>(smem);   // [SH][CW_XSTR] of (V, R): one ds_read_b64 feeds both MFMAs
    float *Ws = smem + 2 * SH * CW_XSTR;             // [KC][32]

    unsigned bid = blockIdx.x;
    const int txi = bid % tiles_x;
    bid /= tiles_x;
    const int tyi = bid % tiles_y;
    bid /= tiles_y;
    const int mt = bid % MT;
    const int n = bid / MT;
    const int u0 = tyi * CW_TY, v0 = txi * CW_TX;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: lives in an SGPR
    const int j = lane & 31, h = lane >> 5;
    const int nA = g.Ay * g.Ax;
    const int need_w = CW_TX + Axp - 1;

    f32x16 an[CW_RB], ap[CW_RB];
#pragma unroll
    for (int rb = 0; rb < CW_RB; ++rb) {
        an[rb] = zero16();
        ap[rb] = zero16();
    }

    const int vv = v0 + j;
    const size_t mstride = (size_t)g.Hy * g.Hx;

    for (int c = 0; c < g.C; ++c) {
        const float *v = V + ((size_t)n * g.C + c) * g.Dy * g.Dx;
        const float *r = Rr + ((size_t)n * g.C + c) * g.Dy * g.Dx;
        __syncthreads();
        if (!(TNMF_ABL(ablate) & 1))
        for (int i = threadIdx.x; i < SH * CW_XSTR; i += kBlock) {
            const int rr = i / CW_XSTR, q = i - rr * CW_XSTR;
            const int y = u0 + rr - (g.Ay - 1), x = v0 + q - (g.Ax - 1);
            const bool in = (q < need_w) && y >= 0 && y < g.Dy && x >= 0 && x < g.Dx;
            const size_t o = (size_t)y * g.Dx + x;
            Xs[i] = in ? float2{v[o], r[o]} : float2{0.f, 0.f};
        }
        if (!(TNMF_ABL(ablate) & 2))
        for (int i = threadIdx.x; i < KC * 32; i += kBlock) {
            const int kk = i >> 5, mi = i & 31;
            const int a = kk / Axp, b = kk - a * Axp;
            const int m = mt * 32 + mi;
            Ws[i] = (b < g.Ax && m < g.M) ? W[((size_t)m * g.C + c) * nA + a * g.Ax + b] : 0.f;
        }
        __syncthreads();

        // flattened k loop over (a, b-pair), unrolled by two with two operand register sets: the LDS reads of step
        // st+1 (one W word + CW_RB (V,R) pairs = 5 reads) are issued before the MFMAs of step st
        const float2 *xb = Xs + (wave * CW_RB) * CW_XSTR + j + h;
        const float *wb = Ws + h * 32 + j;
        const int nsteps = g.Ay * (Axp >> 1);
        int b2 = 0, xo = 0, st = 0;
        float wA, wB;
        float2 xA[CW_RB], xB[CW_RB];
#define CW_LOAD(w_, x_)                                                                      \
    do {                                                                                     \
        w_ = wb[st * 64];                                                                    \
        _Pragma("unroll") for (int rb = 0; rb < CW_RB; ++rb) x_[rb] = xb[rb * CW_XSTR + xo]; \
    } while (0)
#define CW_NEXT()                  \
    do {                           \
        ++st;                      \
        b2 += 2;                   \
        xo += 2;                   \
        if (b2 == Axp) {           \
            b2 = 0;                \
            xo += CW_XSTR - Axp;   \
        }                          \
    } while (0)
#define CW_MMA(w_, x_)                                        \
    do {                                                      \
        _Pragma("unroll") for (int rb = 0; rb < CW_RB; ++rb) { \
            an[rb] = mfma32(w_, x_[rb].x, an[rb]);            \
            ap[rb] = mfma32(w_, x_[rb].y, ap[rb]);            \
        }                                                     \
    } while (0)
        if (TNMF_ABL(ablate) & 4) st = nsteps;
        CW_LOAD(wA, xA);
        while (st + 2 <= nsteps) {
            CW_NEXT();
            CW_LOAD(wB, xB);
            __builtin_amdgcn_sched_barrier(0);
            CW_MMA(wA, xA);
            CW_NEXT();
            if (st < nsteps) CW_LOAD(wA, xA);
            __builtin_amdgcn_sched_barrier(0);
            CW_MMA(wB, xB);
        }
        if (st < nsteps) CW_MMA(wA, xA);
#undef CW_LOAD
#undef CW_NEXT
#undef CW_MMA
    }

    // epilogue.  Lane (j, h) holds pixel column v0+j of 16 atoms per accumulator: for a fixed register the 32 lanes of a
    // half touch 128 contiguous bytes.  FUSED: the 16 H values of one tile row are loaded before the first store so that
    // the loads are independent: one memory round trip per tile row.  (Loading all four rows at once costs a wave of
    // occupancy and measured slower; see DESIGN.md.)
    if (vv < g.Hx && !(TNMF_ABL(ablate) & 8)) {
#pragma unroll
        for (int rb = 0; rb < CW_RB; ++rb) {
            const int u = u0 + wave * CW_RB + rb;
            if (u < g.Hy) {
                const size_t base = (((size_t)n * g.M + mt * 32 + 4 * h) * g.Hy + u) * g.Hx + vv;
                if (FUSED) {
                    float hv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = (r & 3) + 8 * (r >> 2);
                        hv[r] = (mt * 32 + 4 * h + ml < g.M) ? Hio[base + ml * mstride] : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = (r & 3) + 8 * (r >> 2);
                        if (mt * 32 + 4 * h + ml < g.M) Hio[base + ml * mstride] = (hv[r] * an[rb][r]) / (ap[rb][r] + reg);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = (r & 3) + 8 * (r >> 2);
                        if (mt * 32 + 4 * h + ml < g.M) {
                            neg[base + ml * mstride] = an[rb][r];
                            pos[base + ml * mstride] = ap[rb][r];
                        }
                    }
                }
            }
        }
    }
}

// ================================================================================================================
// corr_W, persistent form (used when W[32 atoms][C][Ay][Axp] fits in LDS next to the window): grid (P, MT).
//   A workgroup keeps W resident and walks tiles of 8 rows x 32 cols (wave = 2 rows, 4 accumulators) of the shift
//   plane; stages = (tile, channel).  While the MFMAs of a stage run, the next stage's zero-padded (V,R) window is in
//   flight into registers and -- in the last channel stage of a fused tile -- so are the H values the epilogue will
//   update: the 4.7 GB read-modify-write of H per call then hides under the matrix pipe instead of serialising
//   memory round trips at the end of every tile (measured 1.0 of 3.9 ms in the one-tile-per-workgroup kernel).
// ================================================================================================================
constexpr int CP_TY = 8, CP_RB = 2, CP_XE4 = 3;   // CP_XE4: 4-column window pieces prefetched per thread (39 x 16 / 256)

template <bool FUSED, int NE, int NBP>
__global__ __launch_bounds__(kBlock, 2) void k_mfma_corr_W_persist(Geo g, int tiles_y, int tiles_x, int ablate,
                                                                   const float *__restrict__ V,
                                                                   const float *__restrict__ Rr,
                                                                   const float *__restrict__ W, float *__restrict__ Hio,
                                                                   float *__restrict__ neg, float *__restrict__ pos,
                                                                   float reg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Axp = (g.Ax + 1) & ~1;
    const int SH = CP_TY + g.Ay - 1;
    const int KC = g.Ay * Axp;
    const int need_w = CW_TX + Axp - 1;
    float2 *Xs = reinterpret_cast<float2 *>(smem);   // [SH][CW_XSTR] of (V, R)
    float *Ws = smem + 2 * SH * CW_XSTR;             // [C][KC][32]

    const int mt = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int nA = g.Ay * g.Ax;

    // W of this atom tile, all channels, once
    for (int i = threadIdx.x; i < g.C * KC * 32; i += kBlock) {
        const int mi = i & 31;
        const int kk = (i >> 5) % KC;
        const int c = i / (32 * KC);
        const int a = kk / Axp, b = kk - a * Axp;
        const int m = mt * 32 + mi;
        Ws[i] = (b < g.Ax && m < g.M) ? W[((size_t)m * g.C + c) * nA + a * g.Ax + b] : 0.f;
    }
    for (int i = threadIdx.x; i < SH * CW_XSTR; i += kBlock) Xs[i] = float2{0.f, 0.f};   // columns beyond need_w stay 0

    const int ntiles = g.N * tiles_y * tiles_x;
    const int nstages = ntiles * g.C;
    f32x4 pxv[NE], pxr[NE];

    auto stage_coords = [&](int st, int &n, int &u0, int &v0, int &c) {
        c = st % g.C;
        int t = blockIdx.x + (st / g.C) * gridDim.x;
        const int txi = t % tiles_x;
        t /= tiles_x;
        const int tyi = t % tiles_y;
        n = t / tiles_y;
        u0 = tyi * CP_TY;
        v0 = txi * CW_TX;
    };
    // window pieces (4 columns = one dwordx4 load per array) of this thread: p = tid + 256 e, e < NE (compile time: the
    // loads are unconditional so that hipcc can count what is in flight); (row, first column) never change
    const int wq4 = (need_w + 3) >> 2;
    const int wpieces = SH * wq4;
    int wrc[NE];                                      // row << 16 | first column
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int i = threadIdx.x + e * kBlock;
        const int ic = i < wpieces ? i : 0;
        const int rr = ic / wq4;
        wrc[e] = (rr << 16) | ((ic - rr * wq4) << 2);
    }
    auto prefetch = [&](int st) {
        int n, u0, v0, c;
        stage_coords(st, n, u0, v0, c);
        const float *vp = V + ((size_t)n * g.C + c) * g.Dy * g.Dx;
        const float *rp = Rr + ((size_t)n * g.C + c) * g.Dy * g.Dx;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int y = u0 + (wrc[e] >> 16) - (g.Ay - 1), xa = v0 + (wrc[e] & 0xffff) - (g.Ax - 1);
            const int yc = y < 0 ? 0 : (y < g.Dy ? y : g.Dy - 1);
            int xs = xa < 0 ? 0 : xa;
            xs = xs < g.Dx - 4 ? xs : g.Dx - 4;   // clamped, always legal start column
            pxv[e] = *reinterpret_cast<const f32x4_u *>(vp + (size_t)yc * g.Dx + xs);
            pxr[e] = *reinterpret_cast<const f32x4_u *>(rp + (size_t)yc * g.Dx + xs);
        }
    };
    auto commit = [&](int st) {
        int n, u0, v0, c;
        stage_coords(st, n, u0, v0, c);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int rr = wrc[e] >> 16, q = wrc[e] & 0xffff;
            const int y = u0 + rr - (g.Ay - 1), xa = v0 + q - (g.Ax - 1);
            const bool yok = y >= 0 && y < g.Dy;
            int xs = xa < 0 ? 0 : xa;
            xs = xs < g.Dx - 4 ? xs : g.Dx - 4;
            const f32x4 tv = pxv[e], tr = pxr[e];
            if (threadIdx.x + e * kBlock < wpieces) {
                float2 *dst = Xs + rr * CW_XSTR + q;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int x = xa + k, d = x - xs;   // d in 0..3 whenever x lies inside the row
                    const bool ok = yok && x >= 0 && x < g.Dx;
                    const float vv = d == 0 ? tv[0] : d == 1 ? tv[1] : d == 2 ? tv[2] : tv[3];
                    const float rv = d == 0 ? tr[0] : d == 1 ? tr[1] : d == 2 ? tr[2] : tr[3];
                    dst[k] = ok ? float2{vv, rv} : float2{0.f, 0.f};
                }
            }
        }
    };

    const int my_tiles = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int my_stages = my_tiles * g.C;
    (void)nstages;
    if (my_stages > 0 && !(TNMF_ABL(ablate) & 1)) prefetch(0);

    f32x16 an[CP_RB], ap[CP_RB];
    float hv[CP_RB][16];
    for (int st = 0; st < my_stages; ++st) {
        int n, u0, v0, c;
        stage_coords(st, n, u0, v0, c);
        if (c == 0) {
#pragma unroll
            for (int rb = 0; rb < CP_RB; ++rb) {
                an[rb] = zero16();
                ap[rb] = zero16();
            }
        }
        lds_barrier();   // every wave is done with the previous window (and, first time, W is staged)
        if (!(TNMF_ABL(ablate) & 1)) commit(st);
        lds_barrier();
        if (st + 1 < my_stages && !(TNMF_ABL(ablate) & 1)) prefetch(st + 1);
        // D = [pixel][atom] (A operand = window, B operand = W): lane (atom = lane&31, hl = lane>>5) holds, per
        // accumulator, pixels 8q + 4hl + {0..3} in registers 4q..4q+3: four consecutive pixels of one atom = one 16-byte
        // access.  (The [atom][pixel] orientation needs 4x as many 4-byte store instructions and is store-issue bound.)
        const int atom = mt * 32 + j;
        const int atomc = atom < g.M ? atom : g.M - 1;
        const bool interior = v0 + CW_TX <= g.Hx;   // wave-uniform: whole tile inside the row
        if (FUSED && c == g.C - 1 && !(TNMF_ABL(ablate) & (8 | 128))) {
            // H values of this lane's outputs: clamped (always legal) addresses, consumed only in the epilogue
#pragma unroll
            for (int rb = 0; rb < CP_RB; ++rb) {
                const int u = u0 + wave * CP_RB + rb;
                const int uc = u < g.Hy ? u : g.Hy - 1;
                const float *hp = Hio + (((size_t)n * g.M + atomc) * g.Hy + uc) * g.Hx;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int p0 = v0 + 8 * q + 4 * h;
                    if (interior) {
                        const f32x4 t4 = *reinterpret_cast<const f32x4_u *>(hp + p0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) hv[rb][4 * q + e] = t4[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) hv[rb][4 * q + e] = hp[p0 + e < g.Hx ? p0 + e : g.Hx - 1];
                    }
                }
            }
        }

        const float2 *xb = Xs + (wave * CP_RB) * CW_XSTR + j + h;
        const float *wb = Ws + c * KC * 32 + h * 32 + j;
        if (NBP > 0) {
            // k loop row by row of the atom, the NBP tap pairs of a row fully unrolled (Axp = 2*NBP at compile time: every
            // LDS offset inside a row is an immediate; only one add per row and operand is left -- the loop is bound by
            // instruction issue otherwise).  Two operand register sets, MFMAs and LDS reads alternated by the scheduler.
            constexpr int NB1 = NBP > 0 ? NBP : 1;
            float wA[NB1], wB[NB1];
            float2 xA[NB1][CP_RB], xB[NB1][CP_RB];
#define CPR_LOAD(w_, x_, A_)                                                                         \
    do {                                                                                             \
        const float *wr_ = wb + (A_) * (NBP * 64);                                                   \
        const float2 *xr_ = xb + (A_) * CW_XSTR;                                                     \
        _Pragma("unroll") for (int q = 0; q < NBP; ++q) {                                            \
            w_[q] = wr_[q * 64];                                                                     \
            _Pragma("unroll") for (int rb = 0; rb < CP_RB; ++rb) x_[q][rb] = xr_[rb * CW_XSTR + 2 * q]; \
        }                                                                                            \
    } while (0)
#define CPR_MMA(w_, x_)                                                                              \
    do {                                                                                             \
        _Pragma("unroll") for (int q = 0; q < NBP; ++q)                                              \
            _Pragma("unroll") for (int rb = 0; rb < CP_RB; ++rb) {                                   \
                an[rb] = mfma32(x_[q][rb].x, w_[q], an[rb]);                                         \
                ap[rb] = mfma32(x_[q][rb].y, w_[q], ap[rb]);                                         \
            }                                                                                        \
    } while (0)
            if (!(TNMF_ABL(ablate) & 4)) {
                int a = 0;
                CPR_LOAD(wA, xA, 0);
                while (a + 2 <= g.Ay) {
                    CPR_LOAD(wB, xB, a + 1);
                    CPR_MMA(wA, xA);
                    a += 2;
                    const int an_ = a < g.Ay ? a : g.Ay - 1;   // last round: a redundant (legal) reload
                    CPR_LOAD(wA, xA, an_);
                    CPR_MMA(wB, xB);
#pragma unroll
                    for (int k = 0; k < 2 * NBP * (1 + CP_RB); ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMAs (V and R of one operand pair)
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                    }
                }
                if (a < g.Ay) CPR_MMA(wA, xA);
            }
#undef CPR_LOAD
#undef CPR_MMA
        } else {
        // k loop over (a, b-pair) of channel c, two operand register sets (see k_mfma_corr_W)
        const int nsteps = g.Ay * (Axp >> 1);
        int b2 = 0, xo = 0, k = (TNMF_ABL(ablate) & 4) ? nsteps : 0;
        float wA, wB;
        float2 xA[CP_RB], xB[CP_RB];
#define CP_LOAD(w_, x_)                                                                      \
    do {                                                                                     \
        w_ = wb[k * 64];                                                                     \
        _Pragma("unroll") for (int rb = 0; rb < CP_RB; ++rb) x_[rb] = xb[rb * CW_XSTR + xo]; \
    } while (0)
#define CP_NEXT()                  \
    do {                           \
        ++k;                       \
        b2 += 2;                   \
        xo += 2;                   \
        if (b2 == Axp) {           \
            b2 = 0;                \
            xo += CW_XSTR - Axp;   \
        }                          \
    } while (0)
#define CP_MMA(w_, x_)                                        \
    do {                                                      \
        _Pragma("unroll") for (int rb = 0; rb < CP_RB; ++rb) { \
            an[rb] = mfma32(x_[rb].x, w_, an[rb]);            \
            ap[rb] = mfma32(x_[rb].y, w_, ap[rb]);            \
        }                                                     \
    } while (0)
        CP_LOAD(wA, xA);
        while (k + 2 <= nsteps) {
            CP_NEXT();
            CP_LOAD(wB, xB);
            __builtin_amdgcn_sched_barrier(0);
            CP_MMA(wA, xA);
            CP_NEXT();
            if (k < nsteps) CP_LOAD(wA, xA);
            __builtin_amdgcn_sched_barrier(0);
            CP_MMA(wB, xB);
        }
        if (k < nsteps) CP_MMA(wA, xA);
#undef CP_LOAD
#undef CP_NEXT
#undef CP_MMA
        }

        if (c == g.C - 1 && atom < g.M && !(TNMF_ABL(ablate) & (8 | 64))) {
#pragma unroll
            for (int rb = 0; rb < CP_RB; ++rb) {
                const int u = u0 + wave * CP_RB + rb;
                if (u < g.Hy) {
                    const size_t row = (((size_t)n * g.M + atom) * g.Hy + u) * g.Hx;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int p0 = v0 + 8 * q + 4 * h;
                        f32x4 o4, n4, q4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // H * neg / (pos + reg) with the hardware reciprocal (1 ulp): within the f32 parity budget
                            o4[e] = __fdividef(hv[rb][4 * q + e] * an[rb][4 * q + e], ap[rb][4 * q + e] + reg);
                            n4[e] = an[rb][4 * q + e];
                            q4[e] = ap[rb][4 * q + e];
                        }
                        if (interior) {
                            if (FUSED) {
                                *reinterpret_cast<f32x4_u *>(Hio + row + p0) = o4;
                            } else {
                                *reinterpret_cast<f32x4_u *>(neg + row + p0) = n4;
                                *reinterpret_cast<f32x4_u *>(pos + row + p0) = q4;
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (p0 + e < g.Hx) {
                                    if (FUSED) {
                                        Hio[row + p0 + e] = o4[e];
                                    } else {
                                        neg[row + p0 + e] = n4[e];
                                        pos[row + p0 + e] = q4[e];
                                    }
                                }
                        }
                    }
                }
            }
        }
    }
}

// ================================================================================================================
// corr_H: grid (P, MT, JG).  Work item = (sample n, CH_RH rows x TW cols of the shift plane), TW chosen to divide
//   Hx with little padding.  16x16x4 MFMA: A[atom][k = pixel] = H tile, B[k = pixel (r,t)][j = column J] =
//   X[c][r - a'][t - b'] with J = c*nA + a'*Ax + b' (zero outside the sample; columns beyond J walk a zero strip).
//   The 4 waves are (atom half ah, row pair kh): wave (ah, kh) multiplies atoms 16*ah..16*ah+15 with rows 2*kh, 2*kh+1
//   of the tile for all NT 16-column tiles of its column group, for V and for R: 2*NT accumulators of 16x16.
//   Layout notes: H tile atom stride == 2 (mod 32) and X row stride == Ax + 1 (mod 32) make both operand reads
//   conflict-free (a wave reads 64 consecutive-ish words).  The next item's H tile and X windows are fetched into
//   registers while the current item is multiplied.
//   Output: partials[p][m*C + c][s = a'*Ax + b'][{V, R}] in double (summed in fixed order by k_corr_H_finalize).
// ================================================================================================================
constexpr int CH_RH = 4;
constexpr int CH_NE4 = 9;        // 16-byte staging pieces of the H tile per thread: 128 lines * (TW/4 <= 18) / 256

struct CorrHGeom {
    int TW, AST, XSTW, rblocks, cblocks, P;
};

template <int NT>
__global__ __launch_bounds__(kBlock, 2) void k_mfma_corr_H(Geo g, CorrHGeom cg, int ablate, const float *__restrict__ V,
                                                           const float *__restrict__ Rr,
                                                           const float *__restrict__ H, double *__restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TW = cg.TW, AST = cg.AST;
    const int XST = cg.XSTW + g.Ax + 1;   // == Ax + 1 (mod 32): conflict-free B reads (see tools/lds_conflicts.py)
    const int XR = CH_RH + g.Ay - 1;
    const int plane = XR * XST;              // one channel of one of V / R
    const int ZL = TW + XST + 8;             // zero strip behind the windows
    float *Hs = smem;                        // [32][AST]  (AST even: the float2 region below stays 8-byte aligned)
    float2 *Xs = reinterpret_cast<float2 *>(Hs + 32 * AST);   // [C][XR][XST] of (V, R), then ZL zero pairs

    const int p = blockIdx.x, mt = blockIdx.y, jg = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: lives in an SGPR
    const int ah = wave >> 1, kh = wave & 1;
    const int j = lane & 15, kq = lane >> 4;
    const int nA = g.Ay * g.Ax;
    const int J = g.C * nA;
    // only the channels this column group touches are staged: [c0, c0 + nch)
    const int c0 = (jg * NT * 16) / nA;
    int c1 = ((jg + 1) * NT * 16 - 1) / nA;
    if (c1 > g.C - 1) c1 = g.C - 1;
    const int nch = c1 - c0 + 1;

    int bo[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = (jg * NT + t) * 16 + j;
        if (col < J) {
            const int c = col / nA, s = col - c * nA;
            const int a = s / g.Ax, b = s - a * g.Ax;
            bo[t] = (c - c0) * plane + (2 * kh + (g.Ay - 1) - a) * XST + (g.Ax - 1) - b + kq;
        } else {
            bo[t] = nch * plane;
        }
    }

    f32x4 accv[NT], accr[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        accv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        accr[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int i = threadIdx.x; i < ZL; i += kBlock) Xs[nch * plane + i] = float2{0.f, 0.f};

    // ---- staging in 16-byte pieces (the address unit handles a wave instruction in ~16 cycles whatever its width).
    // H tile: 128 lines (atom mi, row) x TW/4 pieces, piece f = tid + 256 e of thread tid; loads are unconditional on
    // clamped, always legal addresses (start column min(t, Hx-4)), the raw values of the NEXT item stay in registers
    // while the current item is multiplied and are shifted/masked when commit() writes them to LDS.
    const int q4 = TW >> 2;
    const int npiece = 32 * CH_RH * q4;
    int plc[CH_NE4];                                  // line << 8 | piece column, fixed
#pragma unroll
    for (int e = 0; e < CH_NE4; ++e) {
        const int f = threadIdx.x + e * kBlock;
        const int fc = f < npiece ? f : 0;
        const int line = fc / q4;
        plc[e] = (line << 8) | (fc - line * q4);
    }
    f32x4 pm[CH_NE4];

    auto item_coords = [&](int it, int &n, int &r0, int &t0) {
        const int cbi = it % cg.cblocks;
        it /= cg.cblocks;
        const int rbi = it % cg.rblocks;
        n = it / cg.rblocks;
        r0 = rbi * CH_RH;
        t0 = cbi * TW;
    };
    auto prefetch = [&](int it) {
        int n, r0, t0;
        item_coords(it, n, r0, t0);
        const int nat = g.M - mt * 32;   // atoms of this tile that exist (>= 1)
        const float *Hn = H + ((size_t)n * g.M + mt * 32) * g.Hy * g.Hx;
#pragma unroll
        for (int e = 0; e < CH_NE4; ++e) {   // unconditional: pieces beyond the tile are clamped to piece 0
            const int line = plc[e] >> 8, col = (plc[e] & 255) << 2;
            int mi = line >> 2, r = r0 + (line & 3);
            mi = mi < nat ? mi : nat - 1;
            r = r < g.Hy ? r : g.Hy - 1;
            int xs = t0 + col;
            xs = xs < g.Hx - 4 ? xs : g.Hx - 4;
            pm[e] = *reinterpret_cast<const f32x4_u *>(Hn + ((size_t)mi * g.Hy + r) * g.Hx + xs);
        }
    };
    auto commit = [&](int it) {
        int n_, r0_, t0_;
        item_coords(it, n_, r0_, t0_);
        const int nat = g.M - mt * 32;
        const bool edge = t0_ + TW > g.Hx;
#pragma unroll
        for (int e = 0; e < CH_NE4; ++e) {
            const int line = plc[e] >> 8, col = (plc[e] & 255) << 2;
            const bool lok = (line >> 2) < nat && r0_ + (line & 3) < g.Hy;
            f32x4 v = pm[e];
            if (edge) {
                const int xa = t0_ + col;
                const int sh = xa - (xa < g.Hx - 4 ? xa : g.Hx - 4);
                const f32x4 t = v;
                v[0] = sh == 0 ? t[0] : sh == 1 ? t[1] : sh == 2 ? t[2] : t[3];
                v[1] = sh == 0 ? t[1] : sh == 1 ? t[2] : t[3];
                v[2] = sh == 0 ? t[2] : t[3];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (xa + k < g.Hx) ? v[k] : 0.f;
            }
            if (!lok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (threadIdx.x + e * kBlock < npiece) {
                // the atom stride is only 8-byte aligned (== 2 mod 32 words): two 8-byte stores
                float2 *dst = reinterpret_cast<float2 *>(Hs + (line >> 2) * AST + (line & 3) * TW + col);
                dst[0] = float2{v[0], v[1]};
                dst[1] = float2{v[2], v[3]};
            }
        }
        // (V, R) windows of the channels of this column group: rows r0-(Ay-1) .. r0+RH-1, columns t0-(Ax-1) .. in pieces
        // of 4; loaded here (small, L2-resident; the second workgroup of the CU covers the latency)
        const int xq4 = (TW + g.Ax - 1 + 3) >> 2;
        const int nxp = nch * XR * xq4;
        for (int i = threadIdx.x; i < nxp; i += kBlock) {
            const int c4 = i % xq4;
            const int row = (i / xq4) % XR;
            const int cl = i / (xq4 * XR);
            const int y = r0_ - (g.Ay - 1) + row;
            const int xa = t0_ - (g.Ax - 1) + 4 * c4;
            const bool yok = y >= 0 && y < g.Dy;
            const int yc = y < 0 ? 0 : (y < g.Dy ? y : g.Dy - 1);
            int xs = xa < 0 ? 0 : xa;
            xs = xs < g.Dx - 4 ? xs : g.Dx - 4;
            const size_t o = (((size_t)n_ * g.C + c0 + cl) * g.Dy + yc) * g.Dx + xs;
            const f32x4 tv = *reinterpret_cast<const f32x4_u *>(V + o);
            const f32x4 tr = *reinterpret_cast<const f32x4_u *>(Rr + o);
            float2 *dst = Xs + cl * plane + row * XST + 4 * c4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x = xa + k, d = x - xs;   // d in 0..3 whenever x is inside the row
                const bool ok = yok && x >= 0 && x < g.Dx;
                const float vv = d == 0 ? tv[0] : d == 1 ? tv[1] : d == 2 ? tv[2] : tv[3];
                const float rr = d == 0 ? tr[0] : d == 1 ? tr[1] : d == 2 ? tr[2] : tr[3];
                dst[k] = ok ? float2{vv, rr} : float2{0.f, 0.f};
            }
        }
    };

    const int items = g.N * cg.rblocks * cg.cblocks;
    if (p < items && !(TNMF_ABL(ablate) & 1)) prefetch(p);
    for (int it = p; it < items; it += cg.P) {
        __syncthreads();   // every wave is done with the previous item's tiles
        if (!(TNMF_ABL(ablate) & 1)) commit(it);
        __syncthreads();
        if (it + cg.P < items && !(TNMF_ABL(ablate) & 1)) prefetch(it + cg.P);   // in flight under the MFMAs below

        // flattened k loop: 2 rows x TW/4 pixel quads; A offsets are linear, B offsets step by XST at the row change.
        // Fine-grained software pipeline: right after the two MFMAs of column tile t its (V,R) operand pair is reloaded
        // for the NEXT k step (one ds_read_b64), so each LDS read has a whole step of MFMAs to land and at most NT+1
        // reads are outstanding (the lgkm counter saturates at 15).
        const float *ha = Hs + (ah * 16 + j) * AST + (2 * kh) * TW + kq;
        const int nq = TW >> 2, nsteps = 2 * nq;
        int sq = 0, xo = 0;
        float hv = ha[0];
        float2 bx[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bx[t] = Xs[bo[t]];
        for (int st = (TNMF_ABL(ablate) & 4) ? nsteps : 1; st < nsteps; ++st) {
            ++sq;
            xo += 4;
            if (sq == nq) {
                sq = 0;
                xo += XST - TW;
            }
            const float hn = ha[st * 4];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                accv[t] = mfma16(hv, bx[t].x, accv[t]);
                accr[t] = mfma16(hv, bx[t].y, accr[t]);
                bx[t] = Xs[bo[t] + xo];
                __builtin_amdgcn_sched_barrier(0);
            }
            hv = hn;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            accv[t] = mfma16(hv, bx[t].x, accv[t]);
            accr[t] = mfma16(hv, bx[t].y, accr[t]);
        }
    }

    // fold the two row-pair waves of each atom half through LDS (fixed order), write this block's partial in double
    __syncthreads();
    float *red = smem;   // [4 waves][2][4 regs][64 lanes], one column tile at a time
    const int MC = g.M * g.C;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[((wave * 2 + 0) * 4 + r) * 64 + lane] = accv[t][r];
            red[((wave * 2 + 1) * 4 + r) * 64 + lane] = accr[t][r];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 2 * 2 * 4 * 64; e += kBlock) {
            const int ln = e & 63;
            const int r = (e >> 6) & 3;
            const int which = (e >> 8) & 1;
            const int half = e >> 9;
            const float s0 = red[(((half * 2 + 0) * 2 + which) * 4 + r) * 64 + ln];
            const float s1 = red[(((half * 2 + 1) * 2 + which) * 4 + r) * 64 + ln];
            const int col = (jg * NT + t) * 16 + (ln & 15);
            const int m = mt * 32 + half * 16 + 4 * (ln >> 4) + r;
            if (col < J && m < g.M) {
                const int c = col / nA, sft = col - c * nA;
                partials[(((size_t)p * MC + (size_t)m * g.C + c) * nA + sft) * 2 + which] = (double)s0 + (double)s1;
            }
        }
        __syncthreads();
    }
}

// ================================================================================================================
// reconstruct: R[c][y][x] = sum_{m,a,b} H[m][y+a][x+b] * Wf[m][c][a][b]  (Wf = W flipped on both shift axes).
//   The output-channel dimension C is tiny, so the GEMM is turned around: for every row r of the shift plane
//       D_r[a][t] = sum_{(m,b)} Wf[m][c][a][b] * H[m][r][t + b]          (16x16x4 MFMA: rows a, cols 16 pixels t)
//   and R[c][y][t] = sum_a D_{y+a}[a][t] is a "col2im" along the row axis.  A wave owns 16 output columns and sweeps
//   r = 0 .. Dy+3+4*gmax in order, so the col2im never crosses lanes of different t:
//     lane group g = lane>>4 holds rows a = 4g+q (q = 0..3) of D_r.  A 3-register rolling window per channel sums
//     q = 0..3 over four consecutive r; the finished group partial P_g(y), y = r-3-4g, is added to a 16-slot LDS ring
//     (one wave, program order: deterministic); after the last group (gmax = (Ay-1)/4) the row y = r-3-4*gmax is
//     complete, is stored and its slot cleared.  Utilisation of the 16 MFMA rows is Ay/16.
//   One workgroup = 4 waves = 64 output columns of one sample, CB channels; H rows are staged RC_RBK at a time for an
//   atom chunk of MB atoms (zero beyond Hy/Hx/M), Wf for the chunk as [c][k = (m,b)][16 rows], b padded to 4.
// ================================================================================================================
constexpr int RC_RBK = 4;
constexpr int RC_NE4_MAX = 12;   // 16-byte staging pieces per thread: 128 lines * (HST/4 <= 24) / 256

template <int CB, int NB>
__global__ __launch_bounds__(kBlock, 2) void k_mfma_reconstruct(Geo g, int MB, int xblocks, int cgroups, int ablate,
                                                                unsigned long long *dbg,
                                                                const float *__restrict__ W,
                                                                const float *__restrict__ H, float *__restrict__ R) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Axp4 = (g.Ax + 3) & ~3;
    const int HST = 64 + Axp4;
    const int K4 = MB * Axp4;
    float *Wl = smem;                          // [CB][K4][16]
    float *Hs = Wl + CB * K4 * 16;             // [MB][RC_RBK][HST]
    float *ring = Hs + MB * RC_RBK * HST;      // [4 waves][CB][16 slots][16 cols]

#ifndef TNMF_DIAG
    dbg = nullptr;   // product build: every stamp below folds away
#endif
    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int xb = bid % xblocks;
    bid /= xblocks;
    const int cg = bid % cgroups;
    const int n = bid / cgroups;
    const int c0 = cg * CB, x0 = xb * 64;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: lives in an SGPR
    const int j = lane & 15, kq = lane >> 4;
    const int gmax = (g.Ay - 1) >> 2;
    const int nA = g.Ay * g.Ax;
    const int nchunks = (g.M + MB - 1) / MB;
    const int rows_total = g.Dy + 3 + 4 * gmax;
    const int nrb = (rows_total + RC_RBK - 1) / RC_RBK;
    const int nstages = nrb * nchunks;

    // ---- staging.  The stage tile [MB*RC_RBK lines][HST] (line = ml*RC_RBK + row) is walked flat in 16-byte pieces:
    // piece f = tid + 256 e of thread tid is (line, 4 columns).  One dwordx4 load per piece: the address unit handles a
    // wave instruction in ~16 cycles whatever its width, so 4-byte loads would quarter the staging rate (measured: issue
    // of 39 dword loads per thread took 14 % and their landing another 20 % of the kernel).  Loads are unconditional on
    // clamped, always legal addresses (rows are only 4-byte aligned: unaligned dwordx4 is fine on gfx950); the raw
    // values of stage s+1 stay untouched in registers while stage s is multiplied and are masked (zero outside H /
    // beyond M) when commit() writes them to LDS with one ds_write_b128 each.
    // Every thread handles exactly RC_NE4 pieces, unconditionally (pieces beyond the tile are clamped to piece 0 and not
    // written): with conditional loads hipcc cannot count what is in flight and turns later waits into full drains.
    constexpr int RC_NE4 = NB > 0 ? (128 * (16 + NB) + kBlock - 1) / kBlock : RC_NE4_MAX;
    const int q4 = HST >> 2;                          // pieces per line
    const int nlines = MB * RC_RBK;
    const int npiece = nlines * q4;
    const bool edge = x0 + HST > g.Hx;                // this workgroup's window crosses the right border
    const float *Hn = H + (size_t)n * g.M * g.Hy * g.Hx;
    int plc[RC_NE4];                                  // line << 8 | piece column, fixed over the sweep
#pragma unroll
    for (int e = 0; e < RC_NE4; ++e) {
        const int f = threadIdx.x + e * kBlock;
        const int fc = f < npiece ? f : 0;
        const int line = fc / q4;
        plc[e] = (line << 8) | (fc - line * q4);
    }
    f32x4 pm[RC_NE4];

    auto prefetch = [&](int stage) {
        const int rb0 = (stage / nchunks) * RC_RBK;
        const int m0 = (stage % nchunks) * MB;
        int nat = g.M - m0;              // atoms of this chunk that exist (>= 1)
        if (nat > MB) nat = MB;
#pragma unroll
        for (int e = 0; e < RC_NE4; ++e) {
            const int line = plc[e] >> 8, col = (plc[e] & 255) << 2;
            int ml = line >> 2, r = rb0 + (line & 3);
            ml = ml < nat ? ml : nat - 1;
            r = r < g.Hy ? r : g.Hy - 1;
            int xs = x0 + col;
            xs = xs < g.Hx - 4 ? xs : g.Hx - 4;   // keep the 4 columns inside the row
            pm[e] = *reinterpret_cast<const f32x4_u *>(Hn + ((size_t)(m0 + ml) * g.Hy + r) * g.Hx + xs);
        }
    };
    auto commit = [&](int stage) {
        const int rb0 = (stage / nchunks) * RC_RBK;
        int nat = g.M - (stage % nchunks) * MB;
        if (nat > MB) nat = MB;
#pragma unroll
        for (int e = 0; e < RC_NE4; ++e) {
            const int line = plc[e] >> 8, col = (plc[e] & 255) << 2;
            const bool lok = (line >> 2) < nat && rb0 + (line & 3) < g.Hy;
            f32x4 v = pm[e];
            if (edge) {
                // the load started at min(x, Hx-4): shift the components back and zero the columns beyond the row
                const int xa = x0 + col;
                const int sh = xa - (xa < g.Hx - 4 ? xa : g.Hx - 4);
                const f32x4 t = v;
                v[0] = sh == 0 ? t[0] : sh == 1 ? t[1] : sh == 2 ? t[2] : t[3];
                v[1] = sh == 0 ? t[1] : sh == 1 ? t[2] : t[3];
                v[2] = sh == 0 ? t[2] : t[3];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (xa + k < g.Hx) ? v[k] : 0.f;
            }
            if (!lok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (threadIdx.x + e * kBlock < npiece) *reinterpret_cast<f32x4 *>(Hs + line * HST + col) = v;
        }
    };

    for (int i = threadIdx.x; i < 4 * CB * 256; i += kBlock) ring[i] = 0.f;

    float w0[CB], w1[CB], w2[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) w0[c] = w1[c] = w2[c] = 0.f;

    if (!(TNMF_ABL(ablate) & 1)) prefetch(0);
    int stage = 0;
    unsigned long long phase[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = dbg ? stamp() : 0ull;
    f32x4 acc[RC_RBK][CB];
    // col2im along the row axis for the rows rb0 .. rb0+RC_RBK-1 held in acc, rows in order.  The ring is private to the
    // wave and LDS operations of one wave execute in program order, so the accumulate (ds_add_f32, no return) needs no
    // wait before the read of the finished row; only the compiler has to be kept from reordering them.
    auto col2im = [&](int rb0) {
#pragma unroll
        for (int rr = 0; rr < RC_RBK; ++rr) {
            const int r = rb0 + rr;
            const int y = r - 3 - 4 * kq;
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                const f32x4 d = acc[rr][c];
                const float emit = w2[c] + d[3];
                w2[c] = w1[c] + d[2];
                w1[c] = w0[c] + d[1];
                w0[c] = d[0];
                if (kq <= gmax) {
                    float *rp = ring + ((wave * CB + c) * 16 + ((y + 64) & 15)) * 16 + j;
                    __hip_atomic_fetch_add(rp, emit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            asm volatile("" ::: "memory");
            const int yd = r - 3 - 4 * gmax;
            const int cc = lane >> 4;
            if (cc < CB) {
                float *rp = ring + ((wave * CB + cc) * 16 + ((yd + 64) & 15)) * 16 + j;
                const float val = __hip_atomic_exchange(rp, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                const int x = x0 + wave * 16 + j;
                if (yd >= 0 && yd < g.Dy && x < g.Dx && c0 + cc < g.C)
                    R[(((size_t)n * g.C + c0 + cc) * g.Dy + yd) * g.Dx + x] = val;
            }
            asm volatile("" ::: "memory");
        }
    };
    // The col2im of a row block (it ends in conditional R stores) is deferred until the NEXT block's tile has been
    // committed: hipcc cannot count conditional stores, so a vmcnt wait that follows them is a full drain, and the wait
    // for the prefetched tile in commit() would otherwise also wait for R stores issued moments before (measured: 21 %
    // of the kernel).  Deferred, nothing younger than the prefetch is outstanding when commit() waits.
    int pending = -1;
    for (int rb0 = 0; rb0 < rows_total; rb0 += RC_RBK) {
        for (int ch = 0; ch < nchunks; ++ch, ++stage) {
            const int m0 = ch * MB;
            STAMP(0);        // rest of the loop body (accumulator init, ...)
            lds_barrier();   // every wave is done with the previous stage's tile (R stores stay in flight)
            STAMP(1);        // barrier 1
            if (nchunks > 1 || rb0 == 0) {
                for (int i = threadIdx.x; i < CB * K4 * 16; i += kBlock) {
                    const int a = i & 15;
                    const int k = (i >> 4) % K4;
                    const int cc = i / (16 * K4);
                    const int ml = k / Axp4, b = k - ml * Axp4;
                    const int m = m0 + ml, c = c0 + cc;
                    const bool ok = a < g.Ay && b < g.Ax && m < g.M && c < g.C;
                    Wl[i] = ok ? W[((size_t)m * g.C + c) * nA + (g.Ay - 1 - a) * g.Ax + (g.Ax - 1 - b)] : 0.f;
                }
            }
            if (!(TNMF_ABL(ablate) & 1)) commit(stage);
            STAMP(2);        // W staging (first stage) + commit
            lds_barrier();
            STAMP(3);        // barrier 2
            if (ch == 0) {
                if (pending >= 0 && !(TNMF_ABL(ablate) & 2)) col2im(pending);
#pragma unroll
                for (int rr = 0; rr < RC_RBK; ++rr)
#pragma unroll
                    for (int c = 0; c < CB; ++c) acc[rr][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            STAMP(6);        // deferred col2im + R stores of the previous row block
            if (stage + 1 < nstages && !(TNMF_ABL(ablate) & 1)) prefetch(stage + 1);   // in flight under the MFMAs below
            STAMP(4);        // prefetch issue

            const float *wl = Wl + kq * 16 + j;
            const float *hb = Hs + wave * 16 + j + kq;
            if (NB > 0) {
                // k loop atom by atom, the NB b-quads of an atom fully unrolled (compile-time LDS offsets); operands of
                // atom ml+1 (NB*CB + NB*RC_RBK reads) are fetched while the NB*RC_RBK*CB MFMAs of atom ml run
                float aA[NB > 0 ? NB : 1][CB], bA[NB > 0 ? NB : 1][RC_RBK], aB[NB > 0 ? NB : 1][CB], bB[NB > 0 ? NB : 1][RC_RBK];
                // compile-time row stride of the stage tile: every LDS offset inside an atom is an immediate, the only
                // address arithmetic left is one add per atom and operand (the loop is issue-bound otherwise: 0.8 VALU
                // adds + 0.9 LDS reads per 32-cycle MFMA measured before)
                constexpr int HSTc = 64 + 4 * NB;
#define RCA_LOAD(a_, b_, ML)                                                                               \
    do {                                                                                                   \
        const float *wa_ = wl + (ML) * (NB * 64);                                                          \
        const float *ha_ = hb + (ML) * (RC_RBK * HSTc);                                                    \
        _Pragma("unroll") for (int q = 0; q < NB; ++q) {                                                   \
            _Pragma("unroll") for (int c = 0; c < CB; ++c) a_[q][c] = wa_[c * K4 * 16 + q * 64];           \
            _Pragma("unroll") for (int rr = 0; rr < RC_RBK; ++rr) b_[q][rr] = ha_[rr * HSTc + 4 * q];      \
        }                                                                                                  \
    } while (0)
#define RCA_MMA(a_, b_)                                                                                    \
    do {                                                                                                   \
        _Pragma("unroll") for (int q = 0; q < NB; ++q)                                                     \
            _Pragma("unroll") for (int rr = 0; rr < RC_RBK; ++rr)                                          \
                _Pragma("unroll") for (int c = 0; c < CB; ++c)                                             \
                    acc[rr][c] = mfma16(a_[q][c], b_[q][rr], acc[rr][c]);                                  \
    } while (0)
                if (!(TNMF_ABL(ablate) & 4)) {
                    // One scheduling region per two atoms: [load B | multiply A | load A' | multiply B].  The LDS reads
                    // are independent of the MFMAs next to them, and the scheduler is told to alternate them one for
                    // one: a wave issues in order, so reads placed in front of an MFMA block leave the matrix pipe idle
                    // while they issue, whereas a read issued in the shadow of a running MFMA is free.
                    int ml = 0;
                    RCA_LOAD(aA, bA, 0);
                    while (ml + 2 <= MB) {
                        RCA_LOAD(aB, bB, ml + 1);
                        RCA_MMA(aA, bA);
                        ml += 2;
                        const int mn = ml < MB ? ml : MB - 1;   // last round: a redundant (legal) reload
                        RCA_LOAD(aA, bA, mn);
                        RCA_MMA(aB, bB);
#pragma unroll
                        for (int k = 0; k < 2 * NB * RC_RBK * CB; ++k) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                        }
                    }
                    if (ml < MB) RCA_MMA(aA, bA);
                }
#undef RCA_LOAD
#undef RCA_MMA
            } else {
            // flattened k loop over (atom, b-quad): A offsets are linear in the step, B offsets wrap per atom;
            // unrolled by two with two operand register sets (LDS reads of step st+1 fly under the MFMAs of step st)
            const int nb = Axp4 >> 2;
            const int nsteps = MB * nb;
            int bq = 0, ho = 0, st = (TNMF_ABL(ablate) & 4) ? nsteps : 0;
            float aA[CB], bA[RC_RBK], aB[CB], bB[RC_RBK];
#define RC_LOAD(a_, b_)                                                                     \
    do {                                                                                    \
        _Pragma("unroll") for (int c = 0; c < CB; ++c) a_[c] = wl[(c * K4 + st * 4) * 16];  \
        _Pragma("unroll") for (int rr = 0; rr < RC_RBK; ++rr) b_[rr] = hb[rr * HST + ho];   \
    } while (0)
#define RC_NEXT()                           \
    do {                                    \
        ++st;                               \
        ++bq;                               \
        ho += 4;                            \
        if (bq == nb) {                     \
            bq = 0;                         \
            ho += RC_RBK * HST - Axp4;      \
        }                                   \
    } while (0)
#define RC_MMA(a_, b_)                                                                      \
    do {                                                                                    \
        _Pragma("unroll") for (int rr = 0; rr < RC_RBK; ++rr)                               \
            _Pragma("unroll") for (int c = 0; c < CB; ++c)                                  \
                acc[rr][c] = mfma16(a_[c], b_[rr], acc[rr][c]);                             \
    } while (0)
            RC_LOAD(aA, bA);
            while (st + 2 <= nsteps) {
                RC_NEXT();
                RC_LOAD(aB, bB);
                __builtin_amdgcn_sched_barrier(0);
                RC_MMA(aA, bA);
                RC_NEXT();
                if (st < nsteps) RC_LOAD(aA, bA);
                __builtin_amdgcn_sched_barrier(0);
                RC_MMA(aB, bB);
            }
            if (st < nsteps) RC_MMA(aA, bA);
#undef RC_LOAD
#undef RC_NEXT
#undef RC_MMA
            }
        }

        STAMP(5);   // MFMA loop(s)
        pending = rb0;
    }
    if (pending >= 0 && !(TNMF_ABL(ablate) & 2)) col2im(pending);
    if (dbg && lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) dbg[((size_t)blockIdx.x * 4 + wave) * 8 + k] = phase[k];
    }
}

struct ReconPlan {
    int CB, cgroups, xblocks, MB;
    size_t lds;
};

ReconPlan plan_reconstruct(const Geo &g) {
    ReconPlan pl;
    // One channel per workgroup when the per-atom unrolled kernel exists (Ax in 5..16): H is then read once per channel,
    // which is cheap next to the matrix work, and the unrolled single-channel loop is the fastest form (config 4:
    // 6.8 ms instead of 9.6 ms with three channels per workgroup; config 5 with two atom chunks: 63 vs 51 ms, so only
    // for M <= 32).  Otherwise up to 4 channels share the H tile.
    const int nbq = ((g.Ax + 3) & ~3) >> 2;
    pl.CB = (nbq >= 2 && nbq <= 4 && g.M <= 32) ? 1 : (g.C < 4 ? g.C : 4);   // several atom chunks: share the W restaging
    pl.cgroups = cdiv(g.C, pl.CB);
    pl.xblocks = cdiv(g.Dx, 64);
    const int Axp4 = (g.Ax + 3) & ~3;
    const int HST = 64 + Axp4;
    const size_t ring = (size_t)4 * pl.CB * 256 * sizeof(float);
    const size_t per_atom = ((size_t)pl.CB * Axp4 * 16 + (size_t)RC_RBK * HST) * sizeof(float);
    const size_t budget = 72 * 1024;
    long MB = (long)((budget - ring) / per_atom);
    if (MB > 32) MB = 32;
    if (MB > g.M) MB = g.M;
    if (MB < 1) MB = 1;
    pl.MB = (int)MB;
    pl.lds = ring + per_atom * pl.MB;
    return pl;
}

struct CorrHPlan {
    int NT, JG, MT;
    CorrHGeom cg;
    size_t lds;
};

CorrHPlan plan_corr_H(const tnmf_hip_ctx *ctx, const Geo &g) {
    CorrHPlan pl;
    const int J = g.C * g.Ay * g.Ax;
    const int tiles = cdiv(J, 16);
    pl.JG = cdiv(tiles, 12);
    pl.NT = cdiv(tiles, pl.JG);
    pl.MT = cdiv(g.M, 32);
    CorrHGeom &cg = pl.cg;
    cg.cblocks = cdiv(g.Hx, 72);
    cg.TW = (cdiv(g.Hx, cg.cblocks) + 3) & ~3;
    cg.XSTW = (cg.TW + 31) & ~31;
    cg.AST = CH_RH * cg.TW;
    while ((cg.AST & 31) != 2) ++cg.AST;
    cg.rblocks = cdiv(g.Hy, CH_RH);
    const long items = (long)g.N * cg.rblocks * cg.cblocks;
    long P = (2L * ctx->num_cu) / ((long)pl.MT * pl.JG);
    if (P < 1) P = 1;
    // keep one f32 accumulation chain below ~32K terms (K per block and wave = items/P * 2 * TW)
    const long minP = (items * 2 * cg.TW + 32767) / 32768;
    if (P < minP) P = minP;
    if (P > items) P = items;
    if (P > 8192) P = 8192;
    cg.P = (int)P;
    const int XST = cg.XSTW + g.Ax + 1;
    const size_t plane = (size_t)(CH_RH + g.Ay - 1) * XST;
    const size_t ZL = cg.TW + XST + 8;
    int nch_max = 1;   // most channels any column group touches
    for (int jg = 0; jg < pl.JG; ++jg) {
        const int nA = g.Ay * g.Ax;
        const int c0 = (jg * pl.NT * 16) / nA;
        int c1 = ((jg + 1) * pl.NT * 16 - 1) / nA;
        if (c1 > g.C - 1) c1 = g.C - 1;
        if (c1 - c0 + 1 > nch_max) nch_max = c1 - c0 + 1;
    }
    const size_t stage = ((size_t)32 * cg.AST + 2 * ((size_t)nch_max * plane + ZL)) * sizeof(float);   // (V,R) pairs
    const size_t red = (size_t)4 * 2 * 4 * 64 * sizeof(float);
    pl.lds = stage > red ? stage : red;
    return pl;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static bool mfma_common(const Geo &g, int dtype) {
    if (dtype != 0) return false;
    if (g.Dy == 1 || g.Ay == 1) return false;   // 1-D signals run on the generic kernels
    if (g.Ax > 32 || g.Ay > 32) return false;
    return true;
}

bool mfma_has_reconstruct(const Geo &g, int dtype) {
    if (!mfma_common(g, dtype)) return false;
    return g.Ay >= 3 && g.Ay <= 16;   // rows of the 16x16 tile = atom rows (utilisation Ay/16)
}

bool mfma_has_corr_W(const Geo &g, int dtype) {
    if (!mfma_common(g, dtype)) return false;
    const size_t lds_w = ((size_t)2 * (CW_TY + g.Ay - 1) * CW_XSTR + (size_t)g.Ay * ((g.Ax + 1) & ~1) * 32) * sizeof(float);
    return lds_w <= 64 * 1024;
}

bool mfma_has_corr_H(const Geo &g, int dtype) {
    if (!mfma_common(g, dtype)) return false;
    tnmf_hip_ctx fake{};
    fake.num_cu = 256;
    const CorrHPlan pl = plan_corr_H(&fake, g);
    return pl.lds <= 80 * 1024 && pl.NT <= 12 && pl.cg.TW <= 72 && g.Hx >= 4 && g.Dx >= 4;
}

// Opt-in to more than 64 KB of dynamic LDS for every kernel that may ask for it.  The attribute belongs to the
// (function, device) pair, so it is set per context right after hipSetDevice, not behind a process-wide flag.
int mfma_prepare_device() {
#define SET_LDS(K_) TNMF_HIP_TRY(hipFuncSetAttribute((const void *)(K_), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    SET_LDS((k_mfma_reconstruct<1, 0>));
    SET_LDS((k_mfma_reconstruct<1, 2>));
    SET_LDS((k_mfma_reconstruct<1, 3>));
    SET_LDS((k_mfma_reconstruct<1, 4>));
    SET_LDS((k_mfma_reconstruct<2, 0>));
    SET_LDS((k_mfma_reconstruct<3, 0>));
    SET_LDS((k_mfma_reconstruct<4, 0>));
    SET_LDS((k_mfma_corr_H<1>));
    SET_LDS((k_mfma_corr_H<2>));
    SET_LDS((k_mfma_corr_H<3>));
    SET_LDS((k_mfma_corr_H<4>));
    SET_LDS((k_mfma_corr_H<5>));
    SET_LDS((k_mfma_corr_H<6>));
    SET_LDS((k_mfma_corr_H<7>));
    SET_LDS((k_mfma_corr_H<8>));
    SET_LDS((k_mfma_corr_H<9>));
    SET_LDS((k_mfma_corr_H<10>));
    SET_LDS((k_mfma_corr_H<11>));
    SET_LDS((k_mfma_corr_H<12>));
#undef SET_LDS
    return TNMF_OK;
}

int mfma_reconstruct(tnmf_hip_ctx *ctx, const Geo &g, const float *W, const float *H, float *R, hipStream_t s) {
    const ReconPlan pl = plan_reconstruct(g);
    const size_t blocks = (size_t)g.N * pl.cgroups * pl.xblocks;
    if (blocks > 0x7fffffffull) return TNMF_E_GEOM;
    unsigned long long *dbg = nullptr;
    static const bool want_stamps = tnmf_diag_env("TNMF_HIP_STAMPS") != nullptr;   // -DTNMF_DIAG builds only
    if (want_stamps) TNMF_HIP_TRY(hipMalloc(&dbg, blocks * 4 * 8 * sizeof(unsigned long long)));
#define LAUNCH_RC(CB_, NB_)                                                                                        \
    do {                                                                                                           \
        hipLaunchKernelGGL((k_mfma_reconstruct<CB_, NB_>), dim3((unsigned)blocks), dim3(kBlock),                   \
                           pl.lds + ((TNMF_ABL(ctx->ablate) & 4096) ? 40 * 1024 : 0), s, g,                                  \
                           pl.MB, pl.xblocks, pl.cgroups, ctx->ablate, dbg, W, H, R);                              \
    } while (0)
#define LAUNCH_RC_NB(CB_)                 \
    switch (nbq) {                        \
        case 2: LAUNCH_RC(CB_, 2); break; \
        case 3: LAUNCH_RC(CB_, 3); break; \
        case 4: LAUNCH_RC(CB_, 4); break; \
        default: LAUNCH_RC(CB_, 0); break;\
    }
    const int nbq = ((g.Ax + 3) & ~3) >> 2;   // b-quads per atom: unrolled kernels for Ax in 5..16
    switch (pl.CB) {   // the per-atom unrolled loop is instantiated for one channel only (register budget)
        case 1: LAUNCH_RC_NB(1); break;
        case 2: LAUNCH_RC(2, 0); break;
        case 3: LAUNCH_RC(3, 0); break;
        default: LAUNCH_RC(4, 0); break;
    }
#undef LAUNCH_RC_NB
    if (dbg) {
        TNMF_HIP_TRY(hipStreamSynchronize(s));
        const size_t nw = blocks * 4;
        unsigned long long *hbuf = (unsigned long long *)malloc(nw * 8 * sizeof(unsigned long long));
        TNMF_HIP_TRY(hipMemcpy(hbuf, dbg, nw * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double sum[8] = {0};
        for (size_t w = 0; w < nw; ++w)
            for (int k = 0; k < 8; ++k) sum[k] += (double)hbuf[w * 8 + k];
        static const char *names[8] = {"rest", "barrier1", "commit", "barrier2", "prefetch", "mfma", "col2im", "-"};
        double tot = 0;
        for (int k = 0; k < 7; ++k) tot += sum[k];
        fprintf(stderr, "[stamps reconstruct] cycles per wave:");
        for (int k = 0; k < 7; ++k) fprintf(stderr, " %s %.0f (%.1f%%)", names[k], sum[k] / nw, 100.0 * sum[k] / tot);
        fprintf(stderr, " total %.0f\n", tot / nw);
        free(hbuf);
        TNMF_HIP_TRY(hipFree(dbg));
    }
#undef LAUNCH_RC
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

int mfma_corr_W(tnmf_hip_ctx *ctx, const Geo &g, const float *V, const float *R, const float *W, float *H_inout,
                float *neg, float *pos, bool fused, float reg, hipStream_t s) {
    {
        // persistent form: W of all channels resident next to an 8-row window
        const int Axp = (g.Ax + 1) & ~1;
        const int SH = CP_TY + g.Ay - 1;
        const size_t lds_p = ((size_t)2 * SH * CW_XSTR + (size_t)g.C * g.Ay * Axp * 32) * sizeof(float);
        const int wpieces = SH * ((CW_TX + Axp - 1 + 3) / 4);
        if (lds_p <= 52 * 1024 && wpieces <= CP_XE4 * kBlock && g.Dx >= 4 && !(TNMF_ABL(ctx->ablate) & 32)) {
            const int tiles_y = cdiv(g.Hy, CP_TY), tiles_x = cdiv(g.Hx, CW_TX), MT = cdiv(g.M, 32);
            const long ntiles = (long)g.N * tiles_y * tiles_x;
            if (ntiles > 0x7fffffffL) return TNMF_E_GEOM;
            long P = (2L * ctx->num_cu) / MT;   // two workgroups per CU in flight
            if (P < 1) P = 1;
            if (P > ntiles) P = ntiles;
            const dim3 grid((unsigned)P, MT);
            const int ne = (wpieces + kBlock - 1) / kBlock;   // 1..CP_XE4
#define LAUNCH_CP(NE_, NBP_)                                                                                        \
    do {                                                                                                            \
        if (fused)                                                                                                  \
            hipLaunchKernelGGL((k_mfma_corr_W_persist<true, NE_, NBP_>), grid, dim3(kBlock), lds_p, s, g, tiles_y,   \
                               tiles_x, ctx->ablate, V, R, W, H_inout, (float *)nullptr, (float *)nullptr, reg);    \
        else                                                                                                        \
            hipLaunchKernelGGL((k_mfma_corr_W_persist<false, NE_, NBP_>), grid, dim3(kBlock), lds_p, s, g, tiles_y,  \
                               tiles_x, ctx->ablate, V, R, W, (float *)nullptr, neg, pos, 0.f);                     \
    } while (0)
            const int nbp = Axp >> 1;   // tap pairs per atom row; unrolled kernels for the common atom widths
            if (ne <= 1 && nbp == 3)
                LAUNCH_CP(1, 3);
            else if (ne <= 1 && nbp == 4)
                LAUNCH_CP(1, 4);
            else if (ne <= 1 && nbp == 5)
                LAUNCH_CP(1, 5);
            else if (ne <= 1 && nbp == 6)
                LAUNCH_CP(1, 6);
            else if (ne <= 2 && nbp == 8)
                LAUNCH_CP(2, 8);
            else if (ne <= 1)
                LAUNCH_CP(1, 0);
            else if (ne == 2)
                LAUNCH_CP(2, 0);
            else
                LAUNCH_CP(3, 0);
#undef LAUNCH_CP
            TNMF_LAUNCH_CHECK();
            return TNMF_OK;
        }
    }
    const int tiles_y = cdiv(g.Hy, CW_TY), tiles_x = cdiv(g.Hx, CW_TX), MT = cdiv(g.M, 32);
    const size_t lds = ((size_t)2 * (CW_TY + g.Ay - 1) * CW_XSTR + (size_t)g.Ay * ((g.Ax + 1) & ~1) * 32) * sizeof(float);
    const size_t blocks = (size_t)g.N * MT * tiles_y * tiles_x;
    if (blocks > 0x7fffffffull) return TNMF_E_GEOM;
    if (fused)
        hipLaunchKernelGGL((k_mfma_corr_W<true>), dim3((unsigned)blocks), dim3(kBlock), lds, s, g, tiles_y, tiles_x, MT,
                           ctx->ablate, V, R, W, H_inout, (float *)nullptr, (float *)nullptr, reg);
    else
        hipLaunchKernelGGL((k_mfma_corr_W<false>), dim3((unsigned)blocks), dim3(kBlock), lds, s, g, tiles_y, tiles_x,
                           MT, ctx->ablate, V, R, W, (float *)nullptr, neg, pos, 0.f);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

int mfma_corr_H_chunks(const tnmf_hip_ctx *ctx, const Geo &g) {
    if (!mfma_has_corr_H(g, 0)) return 0;
    return plan_corr_H(ctx, g).cg.P;
}

int mfma_corr_H(tnmf_hip_ctx *ctx, const Geo &g, const float *V, const float *R, const float *H, double *partials,
                int P, hipStream_t s) {
    const CorrHPlan pl = plan_corr_H(ctx, g);
    if (P != pl.cg.P) return TNMF_E_WORKSPACE;
    const dim3 grid(pl.cg.P, pl.MT, pl.JG);
#define LAUNCH_CH(NT_)                                                                                             \
    case NT_: {                                                                                                    \
        hipLaunchKernelGGL((k_mfma_corr_H<NT_>), grid, dim3(kBlock), pl.lds, s, g, pl.cg, ctx->ablate, V, R, H,    \
                           partials);                                                                              \
    } break
    switch (pl.NT) {
        LAUNCH_CH(1);
        LAUNCH_CH(2);
        LAUNCH_CH(3);
        LAUNCH_CH(4);
        LAUNCH_CH(5);
        LAUNCH_CH(6);
        LAUNCH_CH(7);
        LAUNCH_CH(8);
        LAUNCH_CH(9);
        LAUNCH_CH(10);
        LAUNCH_CH(11);
        LAUNCH_CH(12);
        default: return TNMF_E_UNSUPPORTED;
    }
#undef LAUNCH_CH
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}
