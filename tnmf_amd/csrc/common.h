// Shared host/device declarations of libtnmf_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "tnmf_hip.h"

// Internal 2-D geometry: 1-D problems run as Dy = Ay = 1.
struct Geo {
    int N, M, C;
    int Dy, Dx;  // sample shape
    int Ay, Ax;  // atom shape
    int Hy, Hx;  // shift (activation) shape: D + A - 1
    int Hs;      // row stride of H in elements (>= Hx; == Hx: C-contiguous)
};

// state of the FFT kernel family (fft.hip): its own workspace and the cache of the spectra of H and V.
//
// The workspace mirrors ONE resident problem -- the "binding": base pointers of the activations H and the samples V of
// `geo.N` samples (tnmf_hip_ctx_bind, or implicitly the operands of the last call).  A call whose H pointer is a whole
// number of samples into the bound H (a mini-batch slice) works on the matching sample range of every per-sample array
// of the workspace, and validity is kept PER SAMPLE: a Cyclic-MU epoch transforms every batch once, exactly like a
// full-batch iteration (the reference's per-slice caches: NumPy_CachingFFT.py:143-158).
#include <vector>
struct FftState {
    void *ws = nullptr;
    size_t ws_bytes = 0;
    size_t failed_bytes = 0;      // smallest workspace request that hipMalloc has refused (0: none); until fft_reserve / release
    bool cache_enabled = false;   // the caller vouches that H and V only change through this library or are announced
                                  // with tnmf_hip_ctx_invalidate (tnmf_hip_ctx_set_cache)
    bool bound = false;           // H_base / V_base / geo / dtype describe the resident problem
    bool explicit_bind = false;   // ... as told by tnmf_hip_ctx_bind (kept across foreign calls), not taken from a call
    const void *H_base = nullptr;
    const void *V_base = nullptr;
    Geo geo = {};                 // geo.N = samples of the binding
    int dtype = 0;
    // per sample of the binding: the workspace holds ...
    std::vector<unsigned char> T_ok;    // the row spectra of H[n]
    std::vector<unsigned char> SH_ok;   // their column transforms (full spectra) as well
    std::vector<unsigned char> V_ok;    // the row spectra of V[n]
    std::vector<unsigned char> SV_ok;   // their full spectra as well
    // the spectra of the dictionary used by reconstruct (row spectra TWr for the mixed kernels, full spectra SW
    // otherwise): W stays the same between the W updates of a mini-batch schedule, every batch reuses them.  Dropped by
    // every entry point that writes W (apply_W, normalize_W, mu_update) and by tnmf_hip_ctx_invalidate.
    const void *W_owner = nullptr;
    bool W_ok = false;
    Geo W_geo = {};
    int W_dtype = 0;
    // row-transform passes over H / V: run, and skipped because the cache held every sample of the call
    // (tnmf_hip_ctx_cache_counters)
    unsigned long long h_runs = 0, h_hits = 0, v_runs = 0, v_hits = 0;
};

struct tnmf_hip_ctx {
    int device;
    int num_cu;
    int path;               // TNMF_PATH_*
    const char *last_path;  // "generic" | "mfma" | "fft"
    int ablate;             // always 0 in the product build; -DTNMF_DIAG builds (tools/probes) read env TNMF_HIP_ABLATE
    void *ws;               // scratch: [R | split-K partials | reduction words]
    size_t ws_bytes;
    int split;              // 1: the H gradient may run on the bf16 matrix cores with 3 x bf16 operand splits (split.hip)
    void *wimg;             // pre-split register images of W for split.hip
    size_t wimg_bytes;
    void *hw = nullptr;     // activation-sized work arrays of tnmf_hip_update_H_ex (lateral terms, padded H, its gradients)
    size_t hw_bytes = 0;
    // persistent schedule kernel: the operation list travels host -> device through a ring of PINNED staging slots (an
    // asynchronous copy from the caller's pageable array could still be reading it after the call has returned); a slot is
    // reused only after the copy that read it has completed (one event per slot)
    static constexpr int kOpSlots = 4;
    void *ops_pinned[kOpSlots] = {nullptr, nullptr, nullptr, nullptr};
    size_t ops_cap[kOpSlots] = {0, 0, 0, 0};
    hipEvent_t ops_done[kOpSlots] = {nullptr, nullptr, nullptr, nullptr};
    int ops_next = 0;
    // persistent schedule kernel (generic.hip: k_schedule): 0 = never, 1 = plain launch of a grid sized by the occupancy
    // query, 2 = the same grid through hipLaunchCooperativeKernel (tnmf_hip_ctx_set_persistent)
    int persistent = 1;
    bool last_schedule_persistent = false;   // the last tnmf_hip_run_schedule ran as ONE launch of k_schedule
    FftState fft;
};

// Diagnostic switches (phase ablation, cycle stamps, forced kernel variants) exist only in -DTNMF_DIAG builds, which
// the timing probes under tools/probes/ make for themselves (`make DIAG=1`, output libtnmf_hip_diag.so).  The product
// library reads no environment variable: TNMF_ABL() folds every ablation test to a compile-time 0 and
// tnmf_diag_env() to "unset".
#ifdef TNMF_DIAG
#include <stdlib.h>
#define TNMF_ABL(x) (x)
static inline const char *tnmf_diag_env(const char *name) { return getenv(name); }
#else
#define TNMF_ABL(x) 0
static inline const char *tnmf_diag_env(const char *) { return nullptr; }
#endif

#define TNMF_HIP_TRY(expr)                          \
    do {                                            \
        hipError_t _e = (expr);                     \
        if (_e != hipSuccess) return (int)_e;       \
    } while (0)

#define TNMF_LAUNCH_CHECK()                         \
    do {                                            \
        hipError_t _e = hipGetLastError();          \
        if (_e != hipSuccess) return (int)_e;       \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

