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

// state of the FFT kernel family (fft.hip): its own workspace and the cache of the row spectra of H
struct FftState {
    void *ws;
    size_t ws_bytes;
    size_t failed_bytes;     // smallest workspace request that hipMalloc has refused (0: none); sticky until release
    bool cache_enabled;      // the caller vouches that H and V only change through this library or are announced
                             // with tnmf_hip_ctx_invalidate (tnmf_hip_ctx_set_cache)
    bool T_valid;            // the workspace holds the row spectra of T_owner for T_geo / T_dtype
    bool SH_valid;           // ... and their column transforms (full spectra of H) as well
    const void *T_owner;
    Geo T_geo;
    int T_dtype;
    bool V_valid;            // ... and the row spectra of the samples V_owner (same geometry rules)
    bool SV_valid;           // ... and their full spectra as well
    const void *V_owner;
    Geo V_geo;
    int V_dtype;
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

