// split_kernels.h -- the H gradient / fused H update (NumPy.py:93-120 + TransformInvariantNMF.py:232-235) on the gfx950 bf16
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
// GEMM view: D[atom][pixel] = sum_k W[atom][k] * X[pixel + k], K = (c, a, b): atoms on the MFMA rows, 32 consecutive
// pixels of one row of the shift plane on the columns = on the lanes.  Two cheap register<->lane bit exchanges in the
// epilogue (8 v_permlane16_swap + 16 DPP moves per accumulator) then give every lane four consecutive pixels of one atom
// and eight adjacent lanes 128 contiguous bytes: the read-modify-write of H (4.7 GB per call at config 3, as much time
// as the MFMAs) is made of fully used cache lines and 16-byte accesses.  (Straight out of either MFMA orientation the
// same traffic ran at 3.3-3.6 TB/s: 32 atom planes per instruction with pixels in the registers, four times as many
// instructions with pixels on the lanes.)
//   X operand (16 k x 32 pixels): lane (i = l & 31, h = l >> 5) holds k = 8h .. 8h+7, i.e. two RUNS of four consecutive
//   taps b0 .. b0+3 of one atom row.  A run of pixel i is 8 contiguous bytes X[row][i + b0 .. i + b0 + 3] of the bf16
//   window -- at a 2-byte alignment that depends on i.  LDS reads wider than 4 bytes must be naturally aligned
//   (cdna_hip_programming.md, Guideline 17), so the window is kept in FOUR copies shifted by 0..3 elements:
//   copy s holds X[. + s], pixel i reads copy i & 3 at element 4 (i >> 2) + b0: 8-byte aligned.  The copies start 64
//   bytes apart (mod 256), so the 32 lanes of a half wave -- 8 lanes per copy, 8 bytes each -- cover all 64 banks once.
//   K order: the runs are enumerated as slots (run r, row pair p), p fastest; lane half h takes atom row 2p + h, so both
//   halves read at the same compile-time offsets from a lane base that already contains h: no address arithmetic in the
//   loop.  The two slots of a k block are never adjacent in memory, and the kernel is compiled without hipcc's DS pairing
//   (target feature load-store-opt): a paired ds_read2_b64 moves 128 bytes per clock where ds_read_b64 moves 256
//   (MI355X_MICROARCH.md, LDS table), and the LDS pipe is the busiest unit of this kernel after the matrix pipe.
//   W operand (32 atoms x 16 k): W pre-split on the device into the exact register image [k block][term][lane][8 bf16]
//   (k_split_prep_W), copied to LDS once per workgroup (one channel) or per stage (several), read with ds_read_b128.
//
// Workgroup = 4 waves = a tile of 8 rows x 32 columns of the shift plane x 32 atoms (8 waves and 16 rows where four waves
// would need more than half a CU's LDS: SplitCfg::WAVES); wave w owns rows 2w, 2w+1, for V and for R: 4 accumulators of
// 32 x 32.  Persistent: a workgroup walks its tiles; while the MFMAs of a stage run, the
// next stage's (V, R) window is in flight into registers and so are the H values the epilogue will update.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <utility>

#include "split.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// device pass only: compile the kernel without hipcc's pairing of DS accesses (see the K-order note above)
#if defined(__HIP_DEVICE_COMPILE__)
#define TNMF_NO_DS_PAIRING __attribute__((target("no-load-store-opt")))
#else
#define TNMF_NO_DS_PAIRING
#endif

namespace {

constexpr int SP_TX = 32, SP_RB = 2;

constexpr int split_plane_bytes(int rows, int wstr) { return ((rows * wstr * 2 - 64 + 255) / 256) * 256 + 64; }

// compile-time geometry of one (atom rows, runs per row) instantiation
// AY == 1 is the 1-D instantiation (signals: one row per sample).  There the ROWS of a tile are eight consecutive SAMPLES
// (no vertical halo), the atom is one row of up to 4 NR4 taps, and the two lane halves of a k block take the two halves
// of 16 consecutive taps (instead of the two rows of an atom row pair): NR4 is a multiple of 4.
template <int AY, int NR4>
struct SplitCfg {
    static constexpr bool ONE_D = AY == 1;
    static_assert(!ONE_D || NR4 % 4 == 0, "1-D: whole k blocks of 16 taps");
    static constexpr int WSTR = 4 * NR4 + 28;          // window row stride (bf16 elements): 4 (i >> 2) + b0 + 3 <= WSTR - 1
    static constexpr int Q = WSTR / 4;                 // 4-element pieces per window row
    static constexpr int NP = (AY + 1) / 2;            // atom row pairs
    static constexpr int NSLOT = ONE_D ? NR4 / 2 : NP * NR4;   // (row pair, run) slots -- 1-D: pairs of runs; a k block holds two
    static constexpr int KB = (NSLOT + 1) / 2;
    // The 16x16x32 form (see "The 16x16x32 form" in front of the kernel): 16 x 16 atoms always (M16AFF: four runs per atom
    // row = the four 16-lane groups of a wave, all offsets affine); 12 x 12 atoms (three runs per row: the (row, run) slots in
    // row-major order, K = 144 padded to 160) only with SEVERAL channels -- measured in same-box A/Bs, config-4 shard
    // 4.69 -> 4.48 ms and 4.91 -> 4.79 ms on two devices, config 3 (one channel: the kernel is co-limited by its memory
    // side and pays the padded half k block in full) 1.675 -> 1.710 ms and a tie (profiles/r04_ab_split12_16x16x32_*.txt).
#ifdef TNMF_SPLIT_NO_M16
    static constexpr bool m16(bool) { return false; }
#elif defined(TNMF_SPLIT_NO_M16_12)
    static constexpr bool m16(bool) { return AY == 16 && NR4 == 4; }
#else
    static constexpr bool m16(bool multi) { return (AY == 16 && NR4 == 4) || (AY == 12 && NR4 == 3 && multi); }
#endif
    static constexpr bool M16AFF = NR4 == 4;
    static constexpr int NSLOT16 = AY * NR4;           // (atom row, run) slots of the 16x16x32 form, row-major
    static constexpr int NKB16 = (NSLOT16 + 7) / 8;    // k blocks of 32 = eight slots
    static constexpr int wimg16 = NKB16 * 6 * 1024, wimg32 = KB * 3 * 1024;   // bytes of the W image of one (atom tile, channel)
    static constexpr int wimg = (m16(true) || m16(false)) && wimg16 > wimg32 ? wimg16 : wimg32;   // (room for either form)
    // Waves per workgroup: four (a tile of 8 rows, two workgroups per CU); EIGHT (16 rows, one W image for twice the pixels)
    // where a four-wave workgroup needs more than 80 KB of LDS and would sit alone on its CU with ONE wave per SIMD (16 x 16
    // atoms, the config-5 shard).  A lone wave leaves the matrix pipe idle in every bubble of its own instruction stream --
    // in-kernel stamps: 47 cycles per MFMA inside the loop where the issue rate is 32, and a fifth of the kernel outside the
    // loop with nothing running -- and a schedule built for the lone wave (second window buffer filled from inside the
    // loop, a k block's 24 MFMAs round robin over the four accumulators with the LDS reads dealt between them, 300 registers)
    // recovered 4 % (30.8 -> 29.5 ms); the eight-wave workgroup, on the plain two-wave schedule, 9.5 % (27.9 ms).
    static constexpr int lds4 = wimg + 24 * split_plane_bytes(8 + AY, WSTR);
    static constexpr int lds8 = wimg + 24 * split_plane_bytes(16 + AY, WSTR);
    static constexpr int WAVES = (!ONE_D && lds4 > 80 * 1024 && lds8 <= 160 * 1024) ? 8 : 4;
    static constexpr int kBlock = 64 * WAVES, TY = SP_RB * WAVES;
    static constexpr int SH = TY + AY - 1;             // window rows that hold data
    static constexpr int SHA = TY + AY;                // + one row of zeros (odd AY: lane half 1 of the last row pair)
    static constexpr int raw = SHA * WSTR * 2;
    static constexpr int planeB = split_plane_bytes(SHA, WSTR);   // bytes per (array, copy): == 64 (mod 256)
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

// -DTNMF_DIAG builds only: cycle stamp (drains the LDS queue: the stamp itself counts on lgkmcnt)
__device__ __forceinline__ unsigned long long split_stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

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

// the same for two values at once, each result a packed pair (first value in the low half): one v_cvt_pk_bf16_f32 per
// term instead of two conversions and a pack
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
    const float r0 = x0 - __builtin_bit_cast(float, hi << 16), r1 = x1 - __builtin_bit_cast(float, hi & 0xffff0000u);
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
    const float s0 = r0 - __builtin_bit_cast(float, mid << 16), s1 = r1 - __builtin_bit_cast(float, mid & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}

__device__ __forceinline__ bf16x8 as_bf16x8(u32x2 lo, u32x2 hi) {
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 mfma16_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// The 16x16x32 form (SplitCfg::m16(): 16 x 16 atoms -- K = 256 per channel, no padding of K).  Under an MFMA-dense loop the
// chip holds a higher clock on v_mfma_f32_16x16x32_bf16 than on 32x32x16 at equal cycles per flop (MI355X_MICROARCH.md,
// DVFS item 7: 1.12-1.15x the FLOP/s), and this instantiation is at its matrix ceiling.
//   K order: k block kb = atom row pair (2 kb, 2 kb + 1); the four 16-lane groups kg of a wave take the four RUNS of an
//   atom row (taps 4 kg .. 4 kg + 3), a lane's 8 bf16 are the run of row 2 kb (elements 0..3) and of row 2 kb + 1 (4..7).
//   All 64 lanes of an X read sit in ONE window row; the groups' pieces overlap (group kg of pixel i and group kg - 1 of
//   pixel i + 4 read the same 8 bytes: broadcast), a 32-lane half covers 72 contiguous bytes in each of two window
//   copies 128 bytes apart: conflict-free on the present row stride.
//   D[atom][pixel] tiles of 16 x 16: tile (ah, ph) = atoms 16 ah .. + 15, tile column j (0..15) = pixel
//   4 (j & 7) + 2 (j >> 3) + ph of the 32-pixel tile row.  Out of the MFMA lane (j, g = l >> 4) holds atoms 4 g + r,
//   r = register 0..3.  ONE exchange -- register bit 0 <-> lane bit 3, two DPP row_ror:8 moves per register pair --
//   leaves lane l with pixels 4 (l & 7) + e (e = 2 s + ph: s = register bit 0 AFTER the exchange, ph = the tile) of atom
//   16 ah + 4 (l >> 4) + 2 r1 + ((l >> 3) & 1) (r1 = register bit 1): four consecutive pixels of one atom per lane, eight
//   adjacent lanes 128 contiguous bytes -- the same store shape as the 32x32 form, with one exchange instead of two.
//   W image: [mt][c][kb (8)][ah (2)][term (3)][lane][8 bf16]: lane (i = l & 15, kg) of (kb, ah) holds taps
//   (a = 2 kb + (j >> 2), b = 4 kg + (j & 3)) of atom 32 mt + 16 ah + i -- the same 48 KB per (atom tile, channel).
//   Three runs per atom row (12 x 12 atoms, K = 144 = 4.5 k blocks of 32: the last half block is zeros in the image): the
//   36 (row, run) slots in row-major order, slot 8 kb + kg + 4 e for lane group kg and lane half-vector e.  The two groups
//   of a 32-lane half then read CONSECUTIVE slots -- the next run of the same row (8 bytes on: the pieces overlap,
//   broadcast) or run 0 of the next row (80 - 16 = 64 bytes on: the other half of the 128-byte bank window) -- conflict-free
//   again; their window offsets are not affine in the lane: three per-lane base registers (see `ap` in the kernel).
__global__ void k_split_prep_W16(Geo g, int NKB, int NR4, int NSLOT, const float *__restrict__ W, u32x4 *__restrict__ Wimg) {
    const int lane = threadIdx.x;
    const int ah = blockIdx.x & 1, kb = (blockIdx.x >> 1) % NKB;
    const int c = (blockIdx.x / (2 * NKB)) % g.C;
    const int mt = blockIdx.x / (2 * NKB * g.C);
    const int m = mt * 32 + 16 * ah + (lane & 15), kg = lane >> 4;
    unsigned t[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int slot = 8 * kb + kg + 4 * (j >> 2);
        const int a = slot / NR4, b = 4 * (slot - a * NR4) + (j & 3);   // (four runs per row: a = 2 kb + (j >> 2), run kg)
        const bool ok = slot < NSLOT && a < g.Ay && b < g.Ax && m < g.M;
        const float w = ok ? W[(((size_t)m * g.C + c) * g.Ay + a) * g.Ax + b] : 0.f;
        split3(w, t[0][j], t[1][j], t[2][j]);
    }
#pragma unroll
    for (int term = 0; term < 3; ++term) {
        u32x4 v;
#pragma unroll
        for (int d = 0; d < 4; ++d) v[d] = t[term][2 * d] | (t[term][2 * d + 1] << 16);
        Wimg[((((size_t)(mt * g.C + c) * NKB + kb) * 2 + ah) * 3 + term) * 64 + lane] = v;
    }
}

// W[M][C][Ay][Ax] -> register images Wimg[mt][c][kb][term][lane][8 bf16] of the B operand (see the file header):
// element j of lane (n = l & 31, h = l >> 5) of k block kb is tap (a = 2p + h, b = 4r + (j & 3)) of slot 2 kb + (j >> 2)
// = (r, p) with p fastest, zero outside the atom / beyond M.
__global__ void k_split_prep_W(Geo g, int NP, int NSLOT, int KB, int one_d, const float *__restrict__ W,
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
        const int r = slot / NP, p = slot - r * NP;
        // 1-D: lane half h of k block kb holds taps 16 kb + 8 h + j
        const int a = one_d ? 0 : 2 * p + h, b = one_d ? 16 * kb + 8 * h + j : 4 * r + (j & 3);
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

// EXTRA (fused updates on row-padded activations only): a further term of the denominator, laid out like H -- the lateral
// inhibition terms of TransformInvariantNMF.py:253-269 (inhibit.hip computes them) -- is loaded behind the MFMA loop,
// into the registers the operand buffers have just left, and added to pos in the epilogue.
template <bool FUSED, bool MULTI, int AY, int NR4, bool EXTRA = false>
__global__ __launch_bounds__((SplitCfg<AY, NR4>::kBlock), 2) TNMF_NO_DS_PAIRING void k_split_corr_W(Geo g, int tiles_y, int tiles_x, int ablate,
                                                            unsigned long long *dbg,
                                                            const float *__restrict__ V, const float *__restrict__ Rr,
                                                            const u32x4 *__restrict__ Wimg, float *__restrict__ Hio,
                                                            float *__restrict__ neg, float *__restrict__ pos,
                                                            float reg, const float *__restrict__ Ex) {
    static_assert(!EXTRA || FUSED, "the extra denominator term belongs to the fused update");
    using Cfg = SplitCfg<AY, NR4>;
    constexpr int WSTR = Cfg::WSTR, Q = Cfg::Q, planeB = Cfg::planeB, KB = Cfg::KB, NSLOT = Cfg::NSLOT, NP = Cfg::NP;
    // (the extra-term epilogue keeps 32 more registers alive: with several channels it stays on the 32x32x16 form, where it
    // fits without spills)
    constexpr bool ONE_D = Cfg::ONE_D, M16 = Cfg::m16(MULTI) && !(EXTRA && MULTI);
    constexpr int kBlock = Cfg::kBlock, SP_TY = Cfg::TY;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Wl = smem;                  // [KB][3][64 lanes][16 bytes]
    unsigned char *Xw = smem + Cfg::wimg;      // [6 arrays: V hi, mid, lo, R hi, mid, lo][4 copies][SHA][WSTR] bf16

    const int mt = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
#ifndef TNMF_DIAG
    dbg = nullptr;   // product build: every stamp below folds away
#endif
    unsigned long long phase[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = dbg ? split_stamp() : 0ull;
#define SP_STAMP(k_)                                      \
    do {                                                  \
        if (dbg) {                                        \
            __builtin_amdgcn_sched_barrier(0);            \
            const unsigned long long t_ = split_stamp();  \
            __builtin_amdgcn_sched_barrier(0);            \
            phase[k_] += t_ - tprev;                      \
            tprev = t_;                                   \
        }                                                 \
    } while (0)

    // zero the whole window once: the spare row and the slack of every plane stay zero (finite) for good
    for (int i = threadIdx.x; i < Cfg::win / 16; i += kBlock) reinterpret_cast<u32x4 *>(Xw)[i] = u32x4{0, 0, 0, 0};

    // W image of one channel: global (L2-resident, pre-split by k_split_prep_W) -> LDS, 16 bytes per thread and piece, in
    // chunks of up to 8 pieces whose loads are all issued before the first store (one memory round trip per chunk, not
    // one per piece).  MULTI (several channels, compile time): the image changes every stage; the first chunk is loaded
    // BEFORE the barrier that frees the previous image and stored behind it.
    constexpr int WPIECES = (M16 ? Cfg::wimg16 : Cfg::wimg32) / 16;   // 16-byte pieces of one W image
    constexpr int NPW = (WPIECES + kBlock - 1) / kBlock, WCH = 8, NWCH = (NPW + WCH - 1) / WCH;
    u32x4 wtmp[WCH];
    auto load_W = [&](int c, int chunk) {
        const u32x4 *src = Wimg + (size_t)(mt * g.C + c) * WPIECES;
#pragma unroll
        for (int k = 0; k < WCH; ++k) {
            const int i = threadIdx.x + (chunk * WCH + k) * kBlock;
            if (chunk * WCH + k < NPW) wtmp[k] = src[i < WPIECES ? i : 0];
        }
    };
    auto store_W = [&](int chunk) {
#pragma unroll
        for (int k = 0; k < WCH; ++k) {
            const int i = threadIdx.x + (chunk * WCH + k) * kBlock;
            if (chunk * WCH + k < NPW && i < WPIECES) reinterpret_cast<u32x4 *>(Wl)[i] = wtmp[k];
        }
    };
    if (!MULTI) {
#pragma unroll
        for (int ch = 0; ch < NWCH; ++ch) {
            load_W(0, ch);
            store_W(ch);
        }
    }

    // Work assignment: a workgroup walks row blocks (8 rows x the full width of the shift plane of one sample) column tile
    // by column tile, left to right.  Rows of H are Hx floats long -- not a multiple of the 128-byte cache line --
    // so every 32-pixel tile shares its first and last line of each row with its neighbours: walked by ONE workgroup the
    // shared lines are read and written within one L2 (the row block of a plane is one contiguous 8.5 KB region);
    // dealt to different workgroups they were fetched twice and written back as partial lines from two XCDs (measured
    // 1.7x the algorithmic H traffic).
    // (1-D: tiles_y = row blocks of eight SAMPLES; n stays 0 and u0 is the first sample of the block)
    // Row blocks are dealt round robin (workgroup b takes b, b + P, ...) as long as every workgroup gets one; the row
    // blocks left over are dealt TILE by tile -- whole blocks would leave some workgroups a block ahead of the others:
    // at BASELINE config 2 (1088 row blocks of five tiles on 512 workgroups) three rounds where 2.1 do (0.131 -> 0.117 ms).
    // (Contiguous runs of tiles per workgroup balance as well, and measured 0.5 % slower at config 3.)
    const int nblocks = ONE_D ? tiles_y : g.N * tiles_y;   // row blocks
    const int P = gridDim.x, full = nblocks / P;           // whole rounds of row blocks
    const int left = (nblocks - full * P) * tiles_x;       // tiles of the last, partial round
    // Coordinates of a stage.  The map from the stage index is full of integer divisions by run-time values (~35 scalar
    // instructions each, a dozen per stage between stage(), the prefetch set-up and the conversion): the coordinates of the
    // CURRENT and the NEXT stage are kept in scalar registers and advanced incrementally -- next channel, next column tile --
    // with the divisions only where a new row block starts.
    struct SCoord {
        int st, n, u0, v0, c, tl, txi;
    };
    auto coords_div = [&](int st) {
        SCoord k;
        k.st = st;
        k.c = st % g.C;
        k.tl = st / g.C;                                             // tile index within this workgroup's walk
        int rbk;
        if (k.tl < full * tiles_x) {
            k.txi = k.tl % tiles_x;
            rbk = blockIdx.x + (k.tl / tiles_x) * P;                 // row block
        } else {
            const int t = blockIdx.x + (k.tl - full * tiles_x) * P;  // tile of the partial round
            k.txi = t % tiles_x;
            rbk = full * P + t / tiles_x;
        }
        const int tyi = rbk % tiles_y;
        k.n = ONE_D ? 0 : rbk / tiles_y;
        k.u0 = tyi * SP_TY;
        k.v0 = k.txi * SP_TX;
        return k;
    };
    auto coords_next = [&](const SCoord &a) {
        SCoord k = a;
        k.st = a.st + 1;
        if (a.c + 1 < g.C) {   // next channel of the same tile
            k.c = a.c + 1;
            return k;
        }
        k.c = 0;
        k.tl = a.tl + 1;
        if (k.tl < full * tiles_x && a.txi + 1 < tiles_x) {   // next column tile of the same row block
            k.txi = a.txi + 1;
            k.v0 = a.v0 + SP_TX;
            return k;
        }
        return coords_div(k.st);   // a new row block, or the partial round
    };
    SCoord ck[2];   // stage s and stage s + 1 of the walk
    ck[0] = coords_div(0);
    ck[1] = coords_next(ck[0]);
    auto stage_coords = [&](int st, int &n, int &u0, int &v0, int &c) {
        const SCoord k = st == ck[0].st ? ck[0] : (st == ck[1].st ? ck[1] : coords_div(st));
        n = k.n;
        u0 = k.u0;
        v0 = k.v0;
        c = k.c;
    };

    // staging item of this thread: window row wr, piece wq (elements 4 wq .. 4 wq + 3 of all four copies, which need
    // window elements 4 wq .. 4 wq + 6)
    const int item = threadIdx.x < Cfg::witems ? threadIdx.x : 0;
    const int wr = item / Q, wq = item - wr * Q;
    // The window of the next stage is prefetched with FOUR 16-byte loads per thread (elements 0..3 and 4..7 of V and of
    // R) on clamped, always legal start columns; they are issued one at a time INSIDE the MFMA loop (mem_slot below).
    // Issued as a burst in front of the loop, the memory instructions of a stage (8 waves of a CU at once) fill the
    // address FIFO of the texture unit until their misses come back -- the L1 keeps ~128 lines in flight -- and every
    // wave sits at its next load instead of starting its MFMAs (measured: a third of the kernel).
    f32x4 pw[4];            // V[0..3], V[4..7], R[0..3], R[4..7] as loaded
    const float *prow[2];   // row of V / R the loads go to
    int pxs[2];             // clamped start columns of the two pieces
    auto prefetch_setup = [&](int st) {
        int n, u0, v0, c;
        stage_coords(st, n, u0, v0, c);
        if (ONE_D) {   // window row wr = sample u0 + wr (clamped; rows beyond the last sample are masked in convert)
            const int sn = u0 + wr < g.N ? u0 + wr : g.N - 1;
            prow[0] = V + ((size_t)sn * g.C + c) * g.Dx;
            prow[1] = Rr + ((size_t)sn * g.C + c) * g.Dx;
        } else {
            const int y = u0 + wr - (g.Ay - 1);
            const int yc = y < 0 ? 0 : (y < g.Dy ? y : g.Dy - 1);
            prow[0] = V + (((size_t)n * g.C + c) * g.Dy + yc) * g.Dx;
            prow[1] = Rr + (((size_t)n * g.C + c) * g.Dy + yc) * g.Dx;
        }
        const int x0 = v0 + 4 * wq - (g.Ax - 1);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int x = x0 + 4 * k;
            pxs[k] = x < 0 ? 0 : (x < g.Dx - 4 ? x : g.Dx - 4);
        }
    };
    auto prefetch_issue = [&](int k) {   // k = 0..3
        pw[k] = *reinterpret_cast<const f32x4_u *>(prow[k >> 1] + pxs[k & 1]);
    };
    // convert(): prefetched f32 window values -> the six bf16 term arrays, packed, still in registers.  It runs right
    // after the MFMA loop (the loads have long landed) and BEFORE the epilogue's stores: hipcc cannot count conditional
    // stores, so a wait for the prefetch that came after them would be a full drain of the H stores (2-3 us each stage).
    // commit(): registers -> the four shifted LDS copies; touches no global memory, so it waits for nothing.
    unsigned tp[6][4];   // [V hi, V mid, V lo, R hi, R mid, R lo][element pair (0,1) (2,3) (4,5) (6,-)]
    unsigned tq[6][3];   // the odd pairs (1,2) (3,4) (5,6) for the copies shifted by 1 and 3
    auto convert = [&](int st) {
        int n, u0, v0, c;
        stage_coords(st, n, u0, v0, c);
        const int y = u0 + wr - (g.Ay - 1);
        const bool yok = ONE_D ? u0 + wr < g.N : (y >= 0 && y < g.Dy);
        const int x0 = v0 + 4 * wq - (g.Ax - 1);
        float fv[7], fr[7];
        // wave-uniform: every 4-column piece of this tile's window lies inside the image row (the second piece of the
        // last item ends at window column WSTR + 3) -> no start column was clamped, the pieces sit where they belong
        const bool xin = v0 - (g.Ax - 1) >= 0 && v0 - (g.Ax - 1) + WSTR + 4 <= g.Dx;
        if (xin) {
#pragma unroll
            for (int e = 0; e < 7; ++e) {
                fv[e] = yok ? pw[e >> 2][e & 3] : 0.f;
                fr[e] = yok ? pw[2 + (e >> 2)][e & 3] : 0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 7; ++e) {
                const int x = x0 + e;
                const bool ok = yok && x >= 0 && x < g.Dx;
                const int xs = x0 + 4 * (e >> 2);
                const int d = x - (xs < 0 ? 0 : (xs < g.Dx - 4 ? xs : g.Dx - 4));   // 0..3 whenever x is inside the row
                const f32x4 a = pw[e >> 2], b = pw[2 + (e >> 2)];
                const float va = d == 0 ? a[0] : d == 1 ? a[1] : d == 2 ? a[2] : a[3];
                const float vb = d == 0 ? b[0] : d == 1 ? b[1] : d == 2 ? b[2] : b[3];
                fv[e] = ok ? va : 0.f;
                fr[e] = ok ? vb : 0.f;
            }
        }
        // element pairs (0,1) (2,3) (4,5) (6,-) of the six term arrays, converted two at a time
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            split3_pair(fv[2 * k], k < 3 ? fv[2 * k + 1] : 0.f, tp[0][k], tp[1][k], tp[2][k]);
            split3_pair(fr[2 * k], k < 3 ? fr[2 * k + 1] : 0.f, tp[3][k], tp[4][k], tp[5][k]);
        }
        // the odd pairs (1,2) (3,4) (5,6): one funnel shift each
#pragma unroll
        for (int ar = 0; ar < 6; ++ar)
#pragma unroll
            for (int k = 0; k < 3; ++k) tq[ar][k] = (tp[ar][k] >> 16) | (tp[ar][k + 1] << 16);
    };
    auto commit = [&]() {
        if (threadIdx.x < Cfg::witems) {
            unsigned char *dst = Xw + (wr * WSTR + 4 * wq) * 2;
#pragma unroll
            for (int ar = 0; ar < 6; ++ar) {
                // copy s holds window elements s .. s+3 of this piece
                *reinterpret_cast<u32x2 *>(dst + (ar * 4 + 0) * planeB) = u32x2{tp[ar][0], tp[ar][1]};
                *reinterpret_cast<u32x2 *>(dst + (ar * 4 + 1) * planeB) = u32x2{tq[ar][0], tq[ar][1]};
                *reinterpret_cast<u32x2 *>(dst + (ar * 4 + 2) * planeB) = u32x2{tp[ar][1], tp[ar][2]};
                *reinterpret_cast<u32x2 *>(dst + (ar * 4 + 3) * planeB) = u32x2{tq[ar][1], tq[ar][2]};
            }
        }
    };

    const int my_tiles = full * tiles_x + ((int)blockIdx.x < left ? (left - blockIdx.x + P - 1) / P : 0);
    const int my_stages = my_tiles * g.C;
    if (my_stages > 0) {
        prefetch_setup(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) prefetch_issue(k);
        convert(0);
    }

    // lane base of the A reads (bytes into Xw): copy (j & 3), window row 2 wave + h, element 4 (j >> 2)
    // Column j of the MFMA tile is pixel pj = 4 (j & 7) + (j >> 3) of the 32-pixel tile row, not pixel j: with this
    // permutation the epilogue's two register<->lane bit exchanges (see there) leave every lane with FOUR CONSECUTIVE
    // pixels of one atom and eight adjacent lanes with 128 contiguous bytes.  Copy pj & 3 = j >> 3, element 4 (pj >> 2):
    // the 32 lanes of a half wave still cover the 64 banks exactly once.
    // (1-D: lane half h is 8 taps further along the row, not one row further down)
    // (M16: tile column jc = lane & 15 is pixel 4 (jc & 7) + 2 (jc >> 3) + ph: window copy 2 (jc >> 3) + ph, the ph in the
    // compile-time offset; lane group kg = lane >> 4 is 4 kg taps further along the row)
    const unsigned char *abase =
        M16 ? Xw + 2 * ((lane >> 3) & 1) * planeB +
                  ((2 * wave) * WSTR + 4 * (lane & 7) + (Cfg::M16AFF ? 4 * (lane >> 4) : 0)) * 2
            : Xw + (j >> 3) * planeB +
                  (ONE_D ? (2 * wave) * WSTR + 4 * (j & 7) + 8 * h : (2 * wave + h) * WSTR + 4 * (j & 7)) * 2;
    const unsigned char *bbase = Wl + lane * 16;
    // 16x16x32 form, three runs per row: with c = 8 kb + 4 e = 3 c3 + cr the slot of lane group kg is (atom row
    // c3 + (cr + kg) / 3, run (cr + kg) % 3): the row c3 goes into the immediate offset of the read and only THREE
    // lane-dependent bases remain, one per cr.
    const unsigned char *ap[3];
    if constexpr (M16 && !Cfg::M16AFF) {
#pragma unroll
        for (int cr = 0; cr < 3; ++cr) {
            const int t = cr + (lane >> 4);
            ap[cr] = abase + ((t / 3) * WSTR + 4 * (t % 3)) * 2;
        }
    } else {
        ap[0] = ap[1] = ap[2] = abase;
    }

    f32x16 acc[SP_RB][2];   // [row of the wave][V | R]
    // M16: the same 64 registers as sixteen 16 x 16 tiles [row of the wave][V | R][atom half][pixel half]; acc[][] is
    // the view the epilogue works on (filled behind the exchange)
    f32x4 acc16[SP_RB][2][2][2];
    // atom of this lane's outputs within the atom tile, and the atom step between the four register groups q:
    // 32x32 form: ((j >> 3) & 3) + 4 h, groups 8 atoms apart; 16x16x32 form: 4 (lane >> 4) + ((lane >> 3) & 1), group
    // q = (atom half, register bit 1) at 16 (q >> 1) + 2 (q & 1)
    const int lane_atom = M16 ? 4 * (lane >> 4) + ((lane >> 3) & 1) : ((j >> 3) & 3) + 4 * h;
    auto q_atoms = [](int q) { return M16 ? 16 * (q >> 1) + 2 * (q & 1) : 8 * q; };
    float hv[SP_RB][16];
    // One stage = (tile, channel).  LAST (compile time) marks the last channel of a tile, the stage that loads the H
    // values and runs the epilogue: as straight-line code, so that hipcc sees every H load consumed on every path (with a
    // run-time `c == C-1` around both, the loads count as pending at the loop back edge and the next H loads into the same
    // registers wait for the whole window prefetch issued in between).
    auto stage = [&](auto lastc, int st) {
        constexpr bool LAST = decltype(lastc)::value;
        int n, u0, v0, c;
        stage_coords(st, n, u0, v0, c);
        if (MULTI && c == 0) {   // (one channel: the first MFMA of every accumulator takes a zero C operand instead)
#pragma unroll
            for (int rb = 0; rb < SP_RB; ++rb) {
                if constexpr (M16) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) acc16[rb][t >> 2][(t >> 1) & 1][t & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
                    acc[rb][0] = acc[rb][1] = zero16();
                }
            }
        }
        SP_STAMP(0);     // stores of the previous epilogue issued, loop overhead
        if (MULTI) load_W(c, 0);
        lds_barrier();   // every wave is done with the previous window and W image (first time: the zero fill)
        SP_STAMP(1);     // barrier 1
        if (MULTI) {
            store_W(0);
#pragma unroll
            for (int ch = 1; ch < NWCH; ++ch) {
                load_W(c, ch);
                store_W(ch);
            }
        }
        if (!(TNMF_ABL(ablate) & 1)) commit();
        lds_barrier();
        SP_STAMP(2);     // commit + barrier 2
        const bool more = st + 1 < my_stages && !(TNMF_ABL(ablate) & 1);
        if (more) prefetch_setup(st + 1);

        // Accumulator layout out of the 32x32 MFMA (atoms on the rows): lane (column j, h), register r <-> atom
        // (r & 3) + 8 (r >> 2) + 4h, pixel pj.  The epilogue exchanges register bit 1 with lane bit 4 (v_permlane16_swap)
        // and register bit 0 with lane bit 3 (DPP row_ror:8 under bank masks); AFTERWARDS register 4q + e of lane (j, h) is
        // pixel 4 (j & 7) + e of atom ((j >> 3) & 3) + 4h + 8q: 16 contiguous bytes per lane and register group, eight
        // adjacent lanes = 128 contiguous bytes of one atom plane, eight atoms per instruction -- the read-modify-write of H
        // goes through fully used cache lines with the fewest possible instructions (8 loads + 8 stores per wave and stage).
        // H of sample n is addressed through a buffer descriptor (base = first plane of the sample, size = M planes) with
        // 32-bit byte offsets; its range check drops the atoms beyond M of a partial atom tile.
        // Row stride: H itself may have padded rows (g.Hs > g.Hx: every 32-pixel tile is then one whole 128-byte line and
        // the pad columns are part of the last tile -- they are read, updated and written like pixels, 0 stays 0, nobody
        // else looks at them); the separate neg / pos outputs of the unfused call are C-contiguous.
        const int hs = FUSED ? g.Hs : g.Hx;
        const unsigned plane4 = (unsigned)g.Hy * hs * 4;     // bytes of one atom plane (1-D: of one atom's row)
        const int p0 = v0 + 4 * (j & 7);                     // first of this lane's four pixels
        const int p0c = p0 < hs - 4 ? p0 : hs - 4;
        const bool interior = v0 + SP_TX <= hs;              // wave-uniform: whole tile inside the (padded) row
        // 2-D: one descriptor per sample (M planes); a row is an offset inside it.  1-D: the rows of the tile are samples, so
        // the descriptor covers the M rows of ONE sample and each of the wave's two rows gets its own (rsrc_of below): the
        // range check drops the atoms beyond M either way.
        const int nrows = ONE_D ? g.N : g.Hy;
        auto rsrc_of = [&](const float *base, int rb) {
            const int u = u0 + wave * SP_RB + rb;
            const size_t first = ONE_D ? (size_t)(u < nrows ? u : nrows - 1) * g.M * hs : (size_t)n * g.M * g.Hy * hs;
            return __builtin_amdgcn_make_buffer_rsrc((void *)(base + first), 0, (int)(g.M * plane4), 0x00020000);
        };
        const __amdgpu_buffer_rsrc_t hrsrc2[SP_RB] = {rsrc_of(Hio, 0), rsrc_of(Hio, 1)};
        unsigned hoff[SP_RB];   // byte offset of (atom of register group 0, row, first pixel), start column clamped
#pragma unroll
        for (int rb = 0; rb < SP_RB; ++rb) {
            const int u = u0 + wave * SP_RB + rb;
            hoff[rb] = (unsigned)(mt * 32 + lane_atom) * plane4 +
                       ((ONE_D ? 0u : (unsigned)(u < g.Hy ? u : g.Hy - 1) * hs) + p0c) * 4;
        }
        // H values of this lane's outputs, consumed only in the epilogue: UNCONDITIONAL 16-byte loads on clamped, always
        // legal addresses, issued one per MFMA group from inside the loop (mem_slot)
        const bool hload = FUSED && LAST && !(TNMF_ABL(ablate) & 128);
        auto h_issue = [&](int k) {   // k = 0..7 = (row of the wave, register group)
            const int rb = k >> 2, q = k & 3;
            const u32x4 t4 = __builtin_amdgcn_raw_buffer_load_b128(hrsrc2[rb], (int)(hoff[rb] + (unsigned)q_atoms(q) * plane4), 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned w = t4[e];
                hv[rb][4 * q + e] = __builtin_bit_cast(float, w);
            }
        };
        // memory slot k of the stage: 0..3 the window prefetch (consumed first: the conversion runs right behind the loop),
        // 4..11 the H loads (last channel of a fused update; consumed by the arithmetic behind the exchanges)
        auto mem_slot = [&](int k) {
            if (k < 4) {
                if (more) prefetch_issue(k);
            } else if (hload) {
                h_issue(k - 4);
            }
        };
        if (TNMF_ABL(ablate) & 4) {   // (diagnostic builds: no MFMA loop to hide them in)
#pragma unroll
            for (int k = 0; k < 12; ++k) mem_slot(k);
        }

        SP_STAMP(3);     // issue of the window prefetch and the H loads
        // ---- MFMA loop: groups g = (k block, row, V | R) of six MFMAs; the operands of group g+1 (six 8-byte reads)
        // and, once per k block, the three W terms of the next k block are fetched under the MFMAs of group g.
        if constexpr (M16) {
          if (!(TNMF_ABL(ablate) & 4)) {
            // groups gi = (k block = atom row pair, row of the wave, V | R, pixel half) of TWELVE MFMAs (six products x two
            // atom halves) on one X operand; the X operand of group gi + 1 (six 8-byte reads) is fetched under group gi.  The
            // W operand (two atom halves x three terms, 24 registers) serves the eight groups of a k block and is single
            // buffered: the next k block's first half is re-read in the shadow of the last group's second six MFMAs, its
            // second half under the first six MFMAs of the next group (a second register set does not fit: 238 of 256).
            constexpr int NKB = Cfg::NKB16, G = NKB * 8;
            u32x2 a[2][3][2];   // [buffer][term][atom row of the pair]
            u32x4 b[2][3];      // [atom half][term]
            auto load_a = [&](int buf, int gi) {
                const int kb = gi >> 3, rb = (gi >> 2) & 1, x = (gi >> 1) & 1, ph = gi & 1;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    // four runs per row: everything but the lane base is an immediate; three: the slot's base + immediate
                    // (slots beyond the atom -- the second half of the last k block of 12 x 12 atoms -- read the last real
                    // rows: W is zero there)
                    const int cslot = 8 * kb + 4 * e, c3 = cslot / 3 < AY - 2 ? cslot / 3 : AY - 2;
                    const unsigned char *base = Cfg::M16AFF ? abase : ap[cslot % 3];
                    const int off = ph * planeB + ((rb + (Cfg::M16AFF ? 2 * kb + e : c3)) * WSTR) * 2;
#pragma unroll
                    for (int term = 0; term < 3; ++term)
                        a[buf][term][e] = *reinterpret_cast<const u32x2 *>(base + (3 * x + term) * 4 * planeB + off);
                }
            };
            auto load_b = [&](int ah, int kb) {
#pragma unroll
                for (int term = 0; term < 3; ++term)
                    b[ah][term] = *reinterpret_cast<const u32x4 *>(bbase + ((kb * 2 + ah) * 3 + term) * 1024);
            };
            load_b(0, 0);
            load_b(1, 0);
            load_a(0, 0);
            __builtin_amdgcn_s_setprio(1);
            static_for<G>([&](auto gic) {
                constexpr int gi = decltype(gic)::value;
                constexpr int kb = gi >> 3, rb = (gi >> 2) & 1, x = (gi >> 1) & 1, ph = gi & 1;
                constexpr int ab = gi & 1;
                constexpr bool nextb = (gi & 7) == 7 && kb + 1 < NKB;
                constexpr int MSTRIDE = G >= 24 ? G / 24 : 1;
                if constexpr (gi + 1 < G) load_a(ab ^ 1, gi + 1);
                if constexpr (gi % MSTRIDE == 0 && gi / MSTRIDE < 12) mem_slot(gi / MSTRIDE);
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 xhi = as_bf16x8(a[ab][0][0], a[ab][0][1]);
                const bf16x8 xmid = as_bf16x8(a[ab][1][0], a[ab][1][1]);
                const bf16x8 xlo = as_bf16x8(a[ab][2][0], a[ab][2][1]);
#pragma unroll
                for (int ah = 0; ah < 2; ++ah) {
                    const bf16x8 whi = __builtin_bit_cast(bf16x8, b[ah][0]);
                    const bf16x8 wmid = __builtin_bit_cast(bf16x8, b[ah][1]);
                    const bf16x8 wlo = __builtin_bit_cast(bf16x8, b[ah][2]);
                    f32x4 d = (!MULTI && kb == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc16[rb][x][ah][ph];
                    d = mfma16_bf16(whi, xlo, d);    // smallest terms first
                    d = mfma16_bf16(wlo, xhi, d);
                    d = mfma16_bf16(wmid, xmid, d);
                    d = mfma16_bf16(whi, xmid, d);
                    d = mfma16_bf16(wmid, xhi, d);
                    d = mfma16_bf16(whi, xhi, d);
                    acc16[rb][x][ah][ph] = d;
                    if constexpr (nextb) {
                        __builtin_amdgcn_sched_barrier(0);
                        load_b(ah, kb + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            __builtin_amdgcn_s_setprio(0);
          }
        } else if (!(TNMF_ABL(ablate) & 4)) {
            constexpr int G = KB * 4;
            u32x2 a[2][3][2];   // [buffer][term][run of the k block]
            u32x4 b[2][3];      // [buffer][term]
            auto load_a = [&](int buf, int gi) {
                const int kb = gi >> 2, rb = (gi >> 1) & 1, x = gi & 1;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    int slot = 2 * kb + e;
                    if (slot >= NSLOT) slot = NSLOT - 1;   // odd slot count: W is zero there, any legal address serves
                    const int r = slot / NP, p = slot - r * NP;   // consecutive slots: consecutive row pairs, same run
                    // (1-D: slot = run pair: taps 16 kb + 4 e + 8 h, the 8 h sits in the lane base)
                    const int off = ONE_D ? (rb * WSTR + 16 * kb + 4 * e) * 2 : ((rb + 2 * p) * WSTR + 4 * r) * 2;
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
            // the wave inside its MFMA loop outranks its SIMD partner, which is in its epilogue most of that time (VALU issue is
            // arbitrated by priority, then age; measured 1.725 -> 1.713 ms at config 3, the opposite assignment 1.74)
            __builtin_amdgcn_s_setprio(1);
            static_for<G>([&](auto gic) {
                constexpr int gi = decltype(gic)::value;
                constexpr int kb = gi >> 2, rb = (gi >> 1) & 1, x = gi & 1;
                constexpr int ab = gi & 1, bb = kb & 1;
                constexpr bool nextb = (gi & 3) == 0 && kb + 1 < KB;
                constexpr int MSTRIDE = G >= 24 ? G / 24 : 1;
                if constexpr (gi + 1 < G) load_a(ab ^ 1, gi + 1);
                if constexpr (nextb) load_b(bb ^ 1, kb + 1);
                // one memory instruction of the stage per group, over the first part of the loop (mem_slot: the next window
                // first, then the H values of the epilogue)
                if constexpr (gi % MSTRIDE == 0 && gi / MSTRIDE < 12) mem_slot(gi / MSTRIDE);
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 ahi = as_bf16x8(a[ab][0][0], a[ab][0][1]);
                const bf16x8 amid = as_bf16x8(a[ab][1][0], a[ab][1][1]);
                const bf16x8 alo = as_bf16x8(a[ab][2][0], a[ab][2][1]);
                const bf16x8 bhi = __builtin_bit_cast(bf16x8, b[bb][0]);
                const bf16x8 bmid = __builtin_bit_cast(bf16x8, b[bb][1]);
                const bf16x8 blo = __builtin_bit_cast(bf16x8, b[bb][2]);
                // D[atom][pixel] += W[atom][k] * X[k][pixel]; one channel: the first k block starts from zero (an inline
                // constant C operand: no 64 register moves per stage to clear the accumulators)
                f32x16 d = (!MULTI && kb == 0) ? zero16() : acc[rb][x];
                // Two waves per SIMD: the reads of the next group go out in one burst in the shadow of the previous group's
                // last MFMA (an MFMA holds the wave's issue port for 8 of its 32 cycles), then the six MFMAs run back to
                // back; the partner wave fills what the burst and the dependent chain leave open.  (sched_group_barrier
                // pipelines over this fully unrolled loop cost hipcc minutes per instantiation; plain scheduling fences
                // give the instruction stream written here.)
                d = mfma_bf16(bhi, alo, d);    // smallest terms first
                d = mfma_bf16(blo, ahi, d);
                d = mfma_bf16(bmid, amid, d);
                d = mfma_bf16(bhi, amid, d);
                d = mfma_bf16(bmid, ahi, d);
                d = mfma_bf16(bhi, ahi, d);
                acc[rb][x] = d;
                __builtin_amdgcn_sched_barrier(0);
            });
            __builtin_amdgcn_s_setprio(0);
            // short loops (the 1-D instantiations with few taps: fewer than 12 groups): the memory slots the loop had no
            // group for are issued behind it
            static_for<12>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int MSTRIDE = G >= 24 ? G / 24 : 1;
                if constexpr (k >= (G + MSTRIDE - 1) / MSTRIDE) mem_slot(k);
            });
        }

        SP_STAMP(4);     // MFMA loop
        float ev[SP_RB][16];
        if constexpr (EXTRA && LAST) {
            static_assert(!EXTRA || !ONE_D, "the extra-term epilogue is instantiated for 2-D problems only");
            const __amdgpu_buffer_rsrc_t ersrc = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(Ex + (size_t)n * g.M * g.Hy * hs), 0, (int)(g.M * plane4), 0x00020000);
#pragma unroll
            for (int rb = 0; rb < SP_RB; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x4 t4 = __builtin_amdgcn_raw_buffer_load_b128(ersrc, (int)(hoff[rb] + (unsigned)q_atoms(q) * plane4), 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned w = t4[e];
                        ev[rb][4 * q + e] = __builtin_bit_cast(float, w);
                    }
                }
        }
        if (more) convert(st + 1);   // before the stores below (see convert)
        SP_STAMP(5);     // convert (the prefetched window has landed under the loop)

        if (LAST && !(TNMF_ABL(ablate) & 8)) {
            // ---- register <-> lane bit exchanges (see the layout note at the H prefetch)
            // (a) register bit 1 <-> lane bit 4: v_permlane16_swap x, y exchanges the odd 16-lane rows of x with the even
            //     rows of y; x = register r (bit 1 clear), y = register r + 2.
            // (b) register bit 0 <-> lane bit 3: lanes 8..15 of every row take the partner register from the lane 8 below,
            //     lanes 0..7 from the lane 8 above: two DPP moves (row_ror:8) under bank masks, no select.
            // In-place inline asm on scalar copies of the accumulators: through the builtins hipcc spends two register
            // moves per exchange on keeping operands it no longer needs.  (s_nop 1: two wait states between a VALU write
            // of an operand and the permlane / DPP read, which hipcc does not insert inside asm.)
#pragma unroll
            for (int rb = 0; rb < SP_RB; ++rb)
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    float w[16];
                    if constexpr (M16) {
                        // tile (ah, ph) register r -> w[4 (2 ah + ph) + r]; only exchange (b) below (register bit 0 <-> lane
                        // bit 3); afterwards register 4 q + e of the 32x32 numbering -- q = 2 ah + r1, e = 2 s + ph -- is
                        // w[4 (2 ah + ph) + 2 r1 + s]
#pragma unroll
                        for (int r = 0; r < 16; ++r) w[r] = acc16[rb][x][r >> 3][(r >> 2) & 1][r & 3];
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) w[r] = acc[rb][x][r];
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (!(r & 2)) asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(w[r]), "+v"(w[r + 2]));
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (!(r & 1)) {
                            const float a0 = w[r];
                            asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xc"   // lanes 8..15: a1 of lane-8
                                : "+v"(w[r])
                                : "v"(w[r + 1]));
                            asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0x3"   // lanes 0..7: a0 of lane+8
                                : "+v"(w[r + 1])
                                : "v"(a0));
                        }
                    if constexpr (M16) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {   // r = 4 q + e of the epilogue's numbering
                            const int q = r >> 2, e = r & 3;
                            acc[rb][x][r] = w[4 * (2 * (q >> 1) + (e & 1)) + 2 * (q & 1) + (e >> 1)];
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[rb][x][r] = w[r];
                    }
                }
            SP_STAMP(6);   // register <-> lane exchanges
            if (FUSED && !interior) {
                // lanes whose four pixels straddle the end of the row loaded from Hx-4: move element e + d to e (d = 1..3);
                // groups entirely beyond the row keep values nobody uses
                const int d = p0 - p0c;
#pragma unroll
                for (int rb = 0; rb < SP_RB; ++rb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float t0 = hv[rb][4 * q], t1 = hv[rb][4 * q + 1], t2 = hv[rb][4 * q + 2], t3 = hv[rb][4 * q + 3];
                        hv[rb][4 * q] = d == 0 ? t0 : d == 1 ? t1 : d == 2 ? t2 : t3;
                        hv[rb][4 * q + 1] = d == 0 ? t1 : d == 1 ? t2 : t3;
                        hv[rb][4 * q + 2] = d == 0 ? t2 : t3;
                    }
            }
            // The arithmetic consumes every prefetched H value UNCONDITIONALLY (only the stores are predicated): a load
            // whose result is used on some paths only stays "pending" for hipcc's wait-count pass at the loop back edge,
            // and the next stage's H loads into the same registers would then wait for the window prefetch in between.
            const __amdgpu_buffer_rsrc_t nrsrc2[SP_RB] = {rsrc_of(FUSED ? Hio : neg, 0), rsrc_of(FUSED ? Hio : neg, 1)};
            const __amdgpu_buffer_rsrc_t prsrc2[SP_RB] = {rsrc_of(FUSED ? Hio : pos, 0), rsrc_of(FUSED ? Hio : pos, 1)};
#pragma unroll
            for (int rb = 0; rb < SP_RB; ++rb) {
                const int u = u0 + wave * SP_RB + rb;
                const bool urow = u < nrows;
                const __amdgpu_buffer_rsrc_t hrsrc = hrsrc2[rb], nrsrc = nrsrc2[rb], prsrc = prsrc2[rb];
                // un-clamped offset of the first pixel (the stores of a border tile go element by element)
                const unsigned soff = hoff[rb] + (unsigned)(p0 - p0c) * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u32x4 o4, n4, q4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float nv = acc[rb][0][4 * q + e], pv_ = acc[rb][1][4 * q + e];
                        // H * neg / (pos + reg) with the hardware reciprocal (v_rcp_f32, 1 ulp; pos + reg > 0): ~2 ulp, inside
                        // the f32 parity budget.  (__fdividef compiles to the full IEEE division sequence here: ten
                        // instructions per element.)
                        float den = pv_ + reg;
                        if constexpr (EXTRA) den = (pv_ + ev[rb][4 * q + e]) + reg;
                        const float ov = FUSED ? hv[rb][4 * q + e] * nv * __builtin_amdgcn_rcpf(den) : 0.f;
                        o4[e] = __builtin_bit_cast(unsigned, ov);
                        n4[e] = __builtin_bit_cast(unsigned, nv);
                        q4[e] = __builtin_bit_cast(unsigned, pv_);
                    }
                    const int off = (int)(soff + (unsigned)q_atoms(q) * plane4);
                    if (interior) {
                        if (urow) {
                            if (FUSED) {
                                __builtin_amdgcn_raw_buffer_store_b128(o4, hrsrc, off, 0, 0);
                            } else {
                                __builtin_amdgcn_raw_buffer_store_b128(n4, nrsrc, off, 0, 0);
                                __builtin_amdgcn_raw_buffer_store_b128(q4, prsrc, off, 0, 0);
                            }
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (urow && p0 + e < hs) {
                                const unsigned ow = o4[e], nw = n4[e], qw = q4[e];
                                if (FUSED) {
                                    __builtin_amdgcn_raw_buffer_store_b32(ow, hrsrc, off + 4 * e, 0, 0);
                                } else {
                                    __builtin_amdgcn_raw_buffer_store_b32(nw, nrsrc, off + 4 * e, 0, 0);
                                    __builtin_amdgcn_raw_buffer_store_b32(qw, prsrc, off + 4 * e, 0, 0);
                                }
                            }
                    }
                }
            }
        }
    };
    int st = 0;
    auto advance = [&]() {
        ck[0] = ck[1];
        ck[1] = coords_next(ck[1]);
        ++st;
    };
    for (int t = 0; t < my_tiles; ++t) {
        for (int c = 0; c + 1 < g.C; ++c) {
            stage(std::false_type{}, st);
            advance();
        }
        stage(std::true_type{}, st);
        advance();
    }
    if (dbg) {
        SP_STAMP(7);   // wait for H, update arithmetic, store issue of the last stage (the others land in phase 0)
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) dbg[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * Cfg::WAVES + wave) * 8 + k] = phase[k];
        }
    }
#undef SP_STAMP
}

template <int AY, int NR4>
int launch(tnmf_hip_ctx *ctx, const Geo &g, const float *V, const float *R, const float *W, float *H_inout,
           float *neg, float *pos, bool fused, float reg, hipStream_t s, const float *extra) {
    using Cfg = SplitCfg<AY, NR4>;
    // the extra-term epilogue loads whole 16-byte groups at H's own offsets: row-padded activations only
    if (extra && (!fused || g.Hs % SP_TX != 0 || Cfg::ONE_D)) return TNMF_E_UNSUPPORTED;
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
    if (Cfg::m16(g.C > 1) && !(fused && extra && g.C > 1))   // (the form the kernel instantiation below runs on)
        hipLaunchKernelGGL(k_split_prep_W16, dim3(MT * g.C * Cfg::NKB16 * 2), dim3(64), 0, s, g, Cfg::NKB16, NR4,
                           Cfg::NSLOT16, W, (u32x4 *)ctx->wimg);
    else
        hipLaunchKernelGGL(k_split_prep_W, dim3(MT * g.C * Cfg::KB), dim3(64), 0, s, g, Cfg::NP, Cfg::NSLOT, Cfg::KB,
                           Cfg::ONE_D ? 1 : 0, W, (u32x4 *)ctx->wimg);
    // (1-D: the rows of a tile are samples: row blocks of eight samples, one "plane")
    constexpr int kBlock = Cfg::kBlock, SP_TY = Cfg::TY;
    const int tiles_y = Cfg::ONE_D ? cdiv(g.N, SP_TY) : cdiv(g.Hy, SP_TY), tiles_x = cdiv(g.Hx, SP_TX);
    const long nrowblocks = Cfg::ONE_D ? tiles_y : (long)g.N * tiles_y;
    const long ntiles = nrowblocks * tiles_x;
    if (ntiles > 0x7fffffffL) return TNMF_E_GEOM;
    const int per_cu = Cfg::lds <= 80 * 1024 ? 2 : 1;
    long P = ((long)per_cu * ctx->num_cu) / MT;
    if (P < 1) P = 1;
    if (P > ntiles) P = ntiles;   // (row blocks round robin, the last partial round tile by tile)
    const dim3 grid((unsigned)P, MT);
    unsigned long long *dbg = nullptr;
    const size_t nw = (size_t)P * MT * Cfg::WAVES;
    static const bool want_stamps = tnmf_diag_env("TNMF_HIP_STAMPS") != nullptr;   // -DTNMF_DIAG builds only
    if (want_stamps) TNMF_HIP_TRY(hipMalloc(&dbg, nw * 8 * sizeof(unsigned long long)));
#define SPLIT_LAUNCH(FUSED_, MULTI_, EXTRA_, H_, NEG_, POS_, REG_)                                                   \
    hipLaunchKernelGGL((k_split_corr_W<FUSED_, MULTI_, AY, NR4, EXTRA_>), grid, dim3(kBlock), Cfg::lds, s, g, tiles_y,   \
                       tiles_x, ctx->ablate, dbg, V, R, (const u32x4 *)ctx->wimg, H_, NEG_, POS_, REG_, extra)
    if (fused && extra) {
        if constexpr (!Cfg::ONE_D) {
            if (g.C > 1)
                SPLIT_LAUNCH(true, true, true, H_inout, (float *)nullptr, (float *)nullptr, reg);
            else
                SPLIT_LAUNCH(true, false, true, H_inout, (float *)nullptr, (float *)nullptr, reg);
        }
    } else if (fused) {
        if (g.C > 1)
            SPLIT_LAUNCH(true, true, false, H_inout, (float *)nullptr, (float *)nullptr, reg);
        else
            SPLIT_LAUNCH(true, false, false, H_inout, (float *)nullptr, (float *)nullptr, reg);
    } else {
        if (g.C > 1)
            SPLIT_LAUNCH(false, true, false, (float *)nullptr, neg, pos, 0.f);
        else
            SPLIT_LAUNCH(false, false, false, (float *)nullptr, neg, pos, 0.f);
    }
#undef SPLIT_LAUNCH
    TNMF_LAUNCH_CHECK();
    if (dbg) {
        TNMF_HIP_TRY(hipStreamSynchronize(s));
        unsigned long long *hbuf = (unsigned long long *)malloc(nw * 8 * sizeof(unsigned long long));
        TNMF_HIP_TRY(hipMemcpy(hbuf, dbg, nw * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double sum[8] = {0};
        for (size_t w = 0; w < nw; ++w)
            for (int k = 0; k < 8; ++k) sum[k] += (double)hbuf[w * 8 + k];
        static const char *names[8] = {"epilogue(H wait, update, stores)", "barrier1", "commit+barrier2", "load issue",
                                       "mfma", "prefetch wait+convert", "exchange", "tail"};
        double tot = 0;
        for (int k = 0; k < 8; ++k) tot += sum[k];
        fprintf(stderr, "[stamps split] cycles per wave:");
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.0f (%.1f%%)", names[k], sum[k] / nw, 100.0 * sum[k] / tot);
        fprintf(stderr, " total %.0f\n", tot / nw);
        free(hbuf);
        TNMF_HIP_TRY(hipFree(dbg));
    }
    return TNMF_OK;
}

template <int AY, int NR4>
int prepare_one() {
#define SPLIT_ATTR(FUSED_, MULTI_, EXTRA_)                                                          \
    TNMF_HIP_TRY(hipFuncSetAttribute((const void *)k_split_corr_W<FUSED_, MULTI_, AY, NR4, EXTRA_>, \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    SPLIT_ATTR(true, true, false);
    SPLIT_ATTR(true, false, false);
    SPLIT_ATTR(false, true, false);
    SPLIT_ATTR(false, false, false);
    if constexpr (!SplitCfg<AY, NR4>::ONE_D) {
        SPLIT_ATTR(true, true, true);
        SPLIT_ATTR(true, false, true);
    }
#undef SPLIT_ATTR
    return TNMF_OK;
}

}  // namespace
