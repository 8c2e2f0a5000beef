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
};

struct tnmf_hip_ctx {
    int device;
    int num_cu;
    int path;               // TNMF_PATH_*
    const char *last_path;  // "generic" | "mfma"
    int ablate;             // diagnostic only (env TNMF_HIP_ABLATE at ctx creation): kernels skip phases; results wrong
    void *ws;               // scratch: [R | split-K partials | reduction words]
    size_t ws_bytes;
};

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

