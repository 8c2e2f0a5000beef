// split.hip -- the H gradient / fused H update (NumPy.py:93-120 + TransformInvariantNMF.py:232-235) on the gfx950 bf16
// matrix cores at float32 accuracy.
//
// Idea.  v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the f32-input MFMA.  Every f32 operand is split EXACTLY
// into three bf16 terms  x = hi + mid + lo  (8 + 8 + 8 significand bits, round-to-nearest at each step, the
// remainders are exact in f32) and a product  x*y  is formed from the six term products with i + j <= 2
//       lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi          (smallest first, f32 accumulation in the MFMA),
// each of which is exact (8 x 8 bits).  What is dropped (mid*lo, lo*mid, lo*lo) is below 2^-23 of the product, i.e. at
// the level of one f32 rounding; measured against a double reference the scheme is slightly MORE accurate than the
// f32 MFMA chain.  Six bf16 MFMAs replace eight f32 MFMAs of the same tile: 16/6 = 2.67x the rate.
//
// GEMM view (as k_mfma_corr_W_persist): D[pixel][atom] = sum_k X[pixel + k] * W[k][atom], K = (c, a, b).
//   A operand (32 pixels x 16 k): lane (i = l & 31, h = l >> 5) holds k = 8h .. 8h+7, i.e. two RUNS of four consecutive
//   taps b0 .. b0+3 of one atom row.  A run of pixel i is 8 contiguous bytes X[row][i + b0 .. i + b0 + 3] of the bf16
//   window -- at a 2-byte alignment that depends on i.  LDS reads wider than 4 bytes must be naturally aligned
//   (cdna_hip_programming.md, Guideline 17), so the window is kept in FOUR copies shifted by 0..3 elements:
//   copy s holds X[. + s], pixel i reads copy i & 3 at element 4 (i >> 2) + b0: 8-byte aligned.  The copies start 64
//   bytes apart (mod 256), so the 32 lanes of a half wave -- 8 lanes per copy, 8 bytes each -- cover all 64 banks once.
//   K order: the runs are enumerated as slots (row pair p, run r); lane half h takes atom row 2p + h, so both halves
//   read at the same compile-time offsets from a lane base that already contains h: no address arithmetic in the loop.
//   B operand (16 k x 32 atoms): W pre-split on the device into the exact register image [k block][term][lane][8 bf16]
//   (k_split_prep_W), copied to LDS once per workgroup (one channel) or per stage (several), read with ds_read_b128.
//
// Workgroup = 4 waves = a tile of 8 rows x 32 columns of the shift plane x 32 atoms; wave w owns rows 2w, 2w+1, for V
// and for R: 4 accumulators of 32 x 32.  Persistent: a workgroup walks its tiles; while the MFMAs of a stage run, the
// next stage's (V, R) window is in flight into registers and so are the H values the epilogue will update.
#include <utility>

#include "split.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int kBlock = 256;
constexpr int SP_TY = 8, SP_TX = 32, SP_RB = 2;

// compile-time geometry of one (atom rows, runs per row) instantiation
template <int AY, int NR4>
struct SplitCfg {
    static constexpr int WSTR = 4 * NR4 + 28;          // window row stride (bf16 elements): 4 (i >> 2) + b0 + 3 <= WSTR - 1
    static constexpr int Q = WSTR / 4;                 // 4-element pieces per window row
    static constexpr int SH = SP_TY + AY - 1;          // window rows that hold data
    static constexpr int SHA = SP_TY + AY;             // + one row of zeros (odd AY: lane half 1 of the last row pair)
    static constexpr int raw = SHA * WSTR * 2;
    static constexpr int planeB = ((raw - 64 + 255) / 256) * 256 + 64;   // bytes per (array, copy): == 64 (mod 256)
    static constexpr int NP = (AY + 1) / 2;            // atom row pairs
    static constexpr int NSLOT = NP * NR4;             // (row pair, run) slots; a k block holds two
    static constexpr int KB = (NSLOT + 1) / 2;
    static constexpr int wimg = KB * 3 * 1024;         // bytes of the W image of one (atom tile, channel)
    static constexpr int win = 24 * planeB;            // 6 arrays x 4 copies
    static constexpr int lds = wimg + win;
    static constexpr int witems = SH * Q;              // staging items (window row, piece) per stage: one per thread
    static_assert(witems <= kBlock, "one staging item per thread");
    static_assert(planeB % 8 == 0 && planeB % 256 == 64, "copy bases 64 bytes apart modulo the bank row");
};

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{}), in order
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// x = hi + mid + lo exactly, each term a bf16 (bit patterns returned)
__device__ __forceinline__ void split3(float x, unsigned &hi, unsigned &mid, unsigned &lo) {
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;   // exact: at most 17 significant bits
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;  // exact: at most 9 significant bits, so lo is exact as well
    const __bf16 l = (__bf16)r2;
    hi = __builtin_bit_cast(unsigned short, h);
    mid = __builtin_bit_cast(unsigned short, m);
    lo = __builtin_bit_cast(unsigned short, l);
}

__device__ __forceinline__ bf16x8 as_bf16x8(u32x2 lo, u32x2 hi) {
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// W[M][C][Ay][Ax] -> register images Wimg[mt][c][kb][term][lane][8 bf16] of the B operand (see the file header):
// element j of lane (n = l & 31, h = l >> 5) of k block kb is tap (a = 2p + h, b = 4r + (j & 3)) of slot 2 kb + (j >> 2)
// = (p, r), zero outside the atom / beyond M.
__global__ void k_split_prep_W(Geo g, int NR4, int NSLOT, int KB, const float *__restrict__ W,
                               u32x4 *__restrict__ Wimg) {
    const int lane = threadIdx.x;
    const int kb = blockIdx.x % KB;
    const int c = (blockIdx.x / KB) % g.C;
    const int mt = blockIdx.x / (KB * g.C);
    const int m = mt * 32 + (lane & 31), h = lane >> 5;
    unsigned t[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int slot = 2 * kb + (j >> 2);
        const int p = slot / NR4, r = slot - p * NR4;
        const int a = 2 * p + h, b = 4 * r + (j & 3);
        const bool ok = slot < NSLOT && a < g.Ay && b < g.Ax && m < g.M;
        const float w = ok ? W[(((size_t)m * g.C + c) * g.Ay + a) * g.Ax + b] : 0.f;
        split3(w, t[0][j], t[1][j], t[2][j]);
    }
#pragma unroll
    for (int term = 0; term < 3; ++term) {
        u32x4 v;
#pragma unroll
        for (int d = 0; d < 4; ++d) v[d] = t[term][2 * d] | (t[term][2 * d + 1] << 16);
        Wimg[(((size_t)(mt * g.C + c) * KB + kb) * 3 + term) * 64 + lane] = v;
    }
}

template <bool FUSED, int AY, int NR4>
__global__ __launch_bounds__(kBlock, 2) void k_split_corr_W(Geo g, int tiles_y, int tiles_x,
                                                            const float *__restrict__ V, const float *__restrict__ Rr,
                                                            const u32x4 *__restrict__ Wimg, float *__restrict__ Hio,
                                                            float *__restrict__ neg, float *__restrict__ pos,
                                                            float reg) {
    using Cfg = SplitCfg<AY, NR4>;
    constexpr int WSTR = Cfg::WSTR, Q = Cfg::Q, planeB = Cfg::planeB, KB = Cfg::KB, NSLOT = Cfg::NSLOT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Wl = smem;                  // [KB][3][64 lanes][16 bytes]
    unsigned char *Xw = smem + Cfg::wimg;      // [6 arrays: V hi, mid, lo, R hi, mid, lo][4 copies][SHA][WSTR] bf16

    const int mt = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;

    // zero the whole window once: the spare row and the slack of every plane stay zero (finite) for good
    for (int i = threadIdx.x; i < Cfg::win / 16; i += kBlock) reinterpret_cast<u32x4 *>(Xw)[i] = u32x4{0, 0, 0, 0};

    auto stage_W = [&](int c) {
        const u32x4 *src = Wimg + (size_t)(mt * g.C + c) * (KB * 3 * 64);
        for (int i = threadIdx.x; i < KB * 3 * 64; i += kBlock) reinterpret_cast<u32x4 *>(Wl)[i] = src[i];
    };
    if (g.C == 1) stage_W(0);

    const int ntiles = g.N * tiles_y * tiles_x;
    auto stage_coords = [&](int st, int &n, int &u0, int &v0, int &c) {
        c = st % g.C;
        int t = blockIdx.x + (st / g.C) * gridDim.x;
        const int txi = t % tiles_x;
        t /= tiles_x;
        const int tyi = t % tiles_y;
        n = t / tiles_y;
        u0 = tyi * SP_TY;
        v0 = txi * SP_TX;
    };

    // staging item of this thread: window row wr, piece wq (elements 4 wq .. 4 wq + 3 of all four copies, which need
    // window elements 4 wq .. 4 wq + 6)
    const int item = threadIdx.x < Cfg::witems ? threadIdx.x : 0;
    const int wr = item / Q, wq = item - wr * Q;
    float pv[7], pr[7];
    auto prefetch = [&](int st) {
        int n, u0, v0, c;
        stage_coords(st, n, u0, v0, c);
        const int y = u0 + wr - (g.Ay - 1);
        const int yc = y < 0 ? 0 : (y < g.Dy ? y : g.Dy - 1);
        const float *vp = V + (((size_t)n * g.C + c) * g.Dy + yc) * g.Dx;
        const float *rp = Rr + (((size_t)n * g.C + c) * g.Dy + yc) * g.Dx;
        const int x0 = v0 + 4 * wq - (g.Ax - 1);
#pragma unroll
        for (int e = 0; e < 7; ++e) {
            const int x = x0 + e;
            const int xc = x < 0 ? 0 : (x < g.Dx ? x : g.Dx - 1);   // clamped: always a legal address
            pv[e] = vp[xc];
            pr[e] = rp[xc];
        }
    };
    auto commit = [&](int st) {
        int n, u0, v0, c;
        stage_coords(st, n, u0, v0, c);
        const int y = u0 + wr - (g.Ay - 1);
        const bool yok = y >= 0 && y < g.Dy;
        const int x0 = v0 + 4 * wq - (g.Ax - 1);
        unsigned t[6][7];   // [V hi, V mid, V lo, R hi, R mid, R lo][element]
#pragma unroll
        for (int e = 0; e < 7; ++e) {
            const int x = x0 + e;
            const bool ok = yok && x >= 0 && x < g.Dx;
            split3(ok ? pv[e] : 0.f, t[0][e], t[1][e], t[2][e]);
            split3(ok ? pr[e] : 0.f, t[3][e], t[4][e], t[5][e]);
        }
        if (threadIdx.x < Cfg::witems) {
            unsigned char *dst = Xw + (wr * WSTR + 4 * wq) * 2;
#pragma unroll
            for (int ar = 0; ar < 6; ++ar)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const u32x2 v = {t[ar][s] | (t[ar][s + 1] << 16), t[ar][s + 2] | (t[ar][s + 3] << 16)};
                    *reinterpret_cast<u32x2 *>(dst + (ar * 4 + s) * planeB) = v;
                }
        }
    };

    const int my_tiles = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int my_stages = my_tiles * g.C;
    if (my_stages > 0) prefetch(0);

    // lane base of the A reads (bytes into Xw): copy (j & 3), window row 2 wave + h, element 4 (j >> 2)
    const unsigned char *abase = Xw + (j & 3) * planeB + ((2 * wave + h) * WSTR + 4 * (j >> 2)) * 2;
    const unsigned char *bbase = Wl + lane * 16;

    f32x16 acc[SP_RB][2];   // [row of the wave][V | R]
    float hv[SP_RB][16];
    for (int st = 0; st < my_stages; ++st) {
        int n, u0, v0, c;
        stage_coords(st, n, u0, v0, c);
        if (c == 0) {
#pragma unroll
            for (int rb = 0; rb < SP_RB; ++rb) acc[rb][0] = acc[rb][1] = zero16();
        }
        lds_barrier();   // every wave is done with the previous window and W image (first time: the zero fill)
        if (g.C > 1) stage_W(c);
        commit(st);
        lds_barrier();
        if (st + 1 < my_stages) prefetch(st + 1);

        const int atom = mt * 32 + j;
        const int atomc = atom < g.M ? atom : g.M - 1;
        const bool interior = v0 + SP_TX <= g.Hx;   // wave-uniform: whole tile inside the row
        if (FUSED && c == g.C - 1) {
            // H values of this lane's outputs (accumulator layout: atom = lane & 31, pixels 8q + 4h + {0..3} in registers
            // 4q .. 4q+3): clamped, always legal addresses, consumed only in the epilogue
#pragma unroll
            for (int rb = 0; rb < SP_RB; ++rb) {
                const int u = u0 + wave * SP_RB + rb;
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

        // ---- MFMA loop: groups g = (k block, row, V | R) of six MFMAs; the operands of group g+1 (six 8-byte reads)
        // and, once per k block, the three W terms of the next k block are fetched under the MFMAs of group g.
        {
            constexpr int G = KB * 4;
            u32x2 a[2][3][2];   // [buffer][term][run of the k block]
            u32x4 b[2][3];      // [buffer][term]
            auto load_a = [&](int buf, int gi) {
                const int kb = gi >> 2, rb = (gi >> 1) & 1, x = gi & 1;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    int slot = 2 * kb + e;
                    if (slot >= NSLOT) slot = NSLOT - 1;   // odd slot count: W is zero there, any legal address serves
                    const int p = slot / NR4, r = slot - p * NR4;
                    const int off = ((rb + 2 * p) * WSTR + 4 * r) * 2;
#pragma unroll
                    for (int term = 0; term < 3; ++term)
                        a[buf][term][e] = *reinterpret_cast<const u32x2 *>(abase + (3 * x + term) * 4 * planeB + off);
                }
            };
            auto load_b = [&](int buf, int kb) {
#pragma unroll
                for (int term = 0; term < 3; ++term)
                    b[buf][term] = *reinterpret_cast<const u32x4 *>(bbase + (kb * 3 + term) * 1024);
            };
            load_b(0, 0);
            load_a(0, 0);
            static_for<G>([&](auto gic) {
                constexpr int gi = decltype(gic)::value;
                constexpr int kb = gi >> 2, rb = (gi >> 1) & 1, x = gi & 1;
                constexpr int ab = gi & 1, bb = kb & 1;
                constexpr bool nextb = (gi & 3) == 0 && kb + 1 < KB;
                if constexpr (gi + 1 < G) load_a(ab ^ 1, gi + 1);
                if constexpr (nextb) load_b(bb ^ 1, kb + 1);
                const bf16x8 ahi = as_bf16x8(a[ab][0][0], a[ab][0][1]);
                const bf16x8 amid = as_bf16x8(a[ab][1][0], a[ab][1][1]);
                const bf16x8 alo = as_bf16x8(a[ab][2][0], a[ab][2][1]);
                const bf16x8 bhi = __builtin_bit_cast(bf16x8, b[bb][0]);
                const bf16x8 bmid = __builtin_bit_cast(bf16x8, b[bb][1]);
                const bf16x8 blo = __builtin_bit_cast(bf16x8, b[bb][2]);
                f32x16 d = acc[rb][x];
                d = mfma_bf16(alo, bhi, d);    // smallest terms first
                d = mfma_bf16(ahi, blo, d);
                d = mfma_bf16(amid, bmid, d);
                d = mfma_bf16(amid, bhi, d);
                d = mfma_bf16(ahi, bmid, d);
                d = mfma_bf16(ahi, bhi, d);
                acc[rb][x] = d;
                if constexpr (gi + 1 < G) {
                    // one LDS read (three while the next k block's W terms are fetched too) in the shadow of every MFMA
                    if constexpr (nextb) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                }
            });
        }

        if (c == g.C - 1 && atom < g.M) {
#pragma unroll
            for (int rb = 0; rb < SP_RB; ++rb) {
                const int u = u0 + wave * SP_RB + rb;
                if (u < g.Hy) {
                    const size_t row = (((size_t)n * g.M + atom) * g.Hy + u) * g.Hx;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int p0 = v0 + 8 * q + 4 * h;
                        f32x4 o4, n4, q4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            n4[e] = acc[rb][0][4 * q + e];
                            q4[e] = acc[rb][1][4 * q + e];
                            // H * neg / (pos + reg) with the hardware reciprocal (1 ulp): within the f32 parity budget
                            if (FUSED) o4[e] = __fdividef(hv[rb][4 * q + e] * n4[e], q4[e] + reg);
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

struct SplitShape {
    int AY, NR4;
};

// instantiated (atom rows, runs per atom row = ceil(Ax / 4)) pairs
constexpr SplitShape kShapes[] = {{12, 3}, {9, 3}, {16, 4}, {7, 2}, {8, 2}, {5, 2}};

bool shape_ok(const Geo &g) {
    const int nr4 = (g.Ax + 3) / 4;
    for (const SplitShape &s : kShapes)
        if (s.AY == g.Ay && s.NR4 == nr4) return true;
    return false;
}

template <int AY, int NR4>
int launch(tnmf_hip_ctx *ctx, const Geo &g, const float *V, const float *R, const float *W, float *H_inout,
           float *neg, float *pos, bool fused, float reg, hipStream_t s) {
    using Cfg = SplitCfg<AY, NR4>;
    const int MT = cdiv(g.M, 32);
    const size_t wbytes = (size_t)MT * g.C * Cfg::wimg;
    if (wbytes > ctx->wimg_bytes) {
        if (ctx->wimg) {
            TNMF_HIP_TRY(hipDeviceSynchronize());
            TNMF_HIP_TRY(hipFree(ctx->wimg));
            ctx->wimg = nullptr;
            ctx->wimg_bytes = 0;
        }
        if (hipMalloc(&ctx->wimg, wbytes) != hipSuccess) {
            (void)hipGetLastError();
            return TNMF_E_WORKSPACE;
        }
        ctx->wimg_bytes = wbytes;
    }
    hipLaunchKernelGGL(k_split_prep_W, dim3(MT * g.C * Cfg::KB), dim3(64), 0, s, g, NR4, Cfg::NSLOT, Cfg::KB, W,
                       (u32x4 *)ctx->wimg);
    const int tiles_y = cdiv(g.Hy, SP_TY), tiles_x = cdiv(g.Hx, SP_TX);
    const long ntiles = (long)g.N * tiles_y * tiles_x;
    if (ntiles > 0x7fffffffL) return TNMF_E_GEOM;
    const int per_cu = Cfg::lds <= 80 * 1024 ? 2 : 1;
    long P = ((long)per_cu * ctx->num_cu) / MT;
    if (P < 1) P = 1;
    if (P > ntiles) P = ntiles;
    const dim3 grid((unsigned)P, MT);
    if (fused)
        hipLaunchKernelGGL((k_split_corr_W<true, AY, NR4>), grid, dim3(kBlock), Cfg::lds, s, g, tiles_y, tiles_x, V, R,
                           (const u32x4 *)ctx->wimg, H_inout, (float *)nullptr, (float *)nullptr, reg);
    else
        hipLaunchKernelGGL((k_split_corr_W<false, AY, NR4>), grid, dim3(kBlock), Cfg::lds, s, g, tiles_y, tiles_x, V, R,
                           (const u32x4 *)ctx->wimg, (float *)nullptr, neg, pos, 0.f);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

template <int AY, int NR4>
int prepare_one() {
    TNMF_HIP_TRY(hipFuncSetAttribute((const void *)k_split_corr_W<true, AY, NR4>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    TNMF_HIP_TRY(hipFuncSetAttribute((const void *)k_split_corr_W<false, AY, NR4>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return TNMF_OK;
}

}  // namespace

bool split_has_corr_W(const Geo &g, int dtype) {
    if (dtype != 0) return false;
    if (g.Dy == 1 || g.Ay == 1) return false;   // 1-D signals: other kernels
    if (g.Ax > 16 || g.Ay > 16) return false;
    return shape_ok(g);
}

int split_prepare_device() {
    int rc;
#define PREP(AY_, NR4_) if ((rc = prepare_one<AY_, NR4_>()) != TNMF_OK) return rc
    PREP(12, 3);
    PREP(9, 3);
    PREP(16, 4);
    PREP(7, 2);
    PREP(8, 2);
    PREP(5, 2);
#undef PREP
    return TNMF_OK;
}

int split_corr_W(tnmf_hip_ctx *ctx, const Geo &g, const float *V, const float *R, const float *W, float *H_inout,
                 float *neg, float *pos, bool fused, float reg, hipStream_t s) {
    const int nr4 = (g.Ax + 3) / 4;
#define DISPATCH(AY_, NR4_) \
    if (g.Ay == AY_ && nr4 == NR4_) return launch<AY_, NR4_>(ctx, g, V, R, W, H_inout, neg, pos, fused, reg, s)
    DISPATCH(12, 3);
    DISPATCH(9, 3);
    DISPATCH(16, 4);
    DISPATCH(7, 2);
    DISPATCH(8, 2);
    DISPATCH(5, 2);
#undef DISPATCH
    return TNMF_E_UNSUPPORTED;
}

void split_release(tnmf_hip_ctx *ctx) {
    if (ctx->wimg) (void)hipFree(ctx->wimg);
    ctx->wimg = nullptr;
    ctx->wimg_bytes = 0;
}
