/*
 * CPU oracle, C flavour: plain nested loops for the three shift-invariant primitives.
 *
 * >>> TEST INFRASTRUCTURE ONLY <<<  (the checker; never linked into or called from the product path).
 * Parity status: PINNED -- tests/test_oracle_pinning.py checks these loops against the golden vectors of the
 * genuine reference PyTorch backend and, through OracleNMF(impl='c'), against the reference's hard-coded energies.
 *
 * Index forms restated from /root/reference/tnmf/backends/NumPy.py (1-D signals are passed as Dy = Ay = 1):
 *   reconstruct   (:122-132)  R[n,c,y,x]   = sum_m sum_{a,b} H[n,m,y+a,x+b] * W[m,c,Ay-1-a,Ax-1-b]
 *   corr_with_W   (:101-109)  O[n,m,u,v]   = sum_c sum_{a,b} W[m,c,a,b] * X[n,c,u+a-(Ay-1),v+b-(Ax-1)]   (X zero outside)
 *   corr_H_with   (:77-90)    G[m,c,a,b]   = sum_n sum_{y,x} H[n,m,y+Ay-1-a,x+Ax-1-b] * X[n,c,y,x]
 * Accumulation is in double for both element types.
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC tnmf_oracle_c.c -o _build/libtnmf_oracle.so
 */
#include <omp.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int N, M, C, Dy, Dx, Ay, Ax;
} geom_t;

void oracle_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }

#define HY(g) ((g)->Dy + (g)->Ay - 1)
#define HX(g) ((g)->Dx + (g)->Ax - 1)

#define DEFINE_ORACLE(T, SUF)                                                                                   \
    void oracle_reconstruct_##SUF(const geom_t *g, const T *W, const T *H, T *R) {                              \
        const int Hy = HY(g), Hx = HX(g);                                                                       \
        _Pragma("omp parallel for collapse(2) schedule(static)")                                                \
        for (int n = 0; n < g->N; ++n)                                                                          \
            for (int c = 0; c < g->C; ++c) {                                                                    \
                double *acc = (double *)calloc((size_t)g->Dy * g->Dx, sizeof(double));                          \
                for (int m = 0; m < g->M; ++m) {                                                                \
                    const T *h = H + ((size_t)n * g->M + m) * Hy * Hx;                                          \
                    const T *w = W + ((size_t)m * g->C + c) * g->Ay * g->Ax;                                    \
                    for (int a = 0; a < g->Ay; ++a)                                                             \
                        for (int b = 0; b < g->Ax; ++b) {                                                       \
                            const double wv = (double)w[(g->Ay - 1 - a) * g->Ax + (g->Ax - 1 - b)];             \
                            for (int y = 0; y < g->Dy; ++y) {                                                   \
                                const T *hr = h + (size_t)(y + a) * Hx + b;                                     \
                                double *ar = acc + (size_t)y * g->Dx;                                           \
                                for (int x = 0; x < g->Dx; ++x) ar[x] += wv * (double)hr[x];                    \
                            }                                                                                   \
                        }                                                                                       \
                }                                                                                               \
                T *r = R + ((size_t)n * g->C + c) * g->Dy * g->Dx;                                              \
                for (size_t i = 0; i < (size_t)g->Dy * g->Dx; ++i) r[i] = (T)acc[i];                            \
                free(acc);                                                                                      \
            }                                                                                                   \
    }                                                                                                           \
                                                                                                                \
    void oracle_corr_with_W_##SUF(const geom_t *g, const T *W, const T *X, T *O) {                              \
        const int Hy = HY(g), Hx = HX(g);                                                                       \
        _Pragma("omp parallel for collapse(2) schedule(static)")                                                \
        for (int n = 0; n < g->N; ++n)                                                                          \
            for (int m = 0; m < g->M; ++m) {                                                                    \
                double *acc = (double *)calloc((size_t)Hy * Hx, sizeof(double));                                \
                for (int c = 0; c < g->C; ++c) {                                                                \
                    const T *xp = X + ((size_t)n * g->C + c) * g->Dy * g->Dx;                                   \
                    const T *w = W + ((size_t)m * g->C + c) * g->Ay * g->Ax;                                    \
                    for (int a = 0; a < g->Ay; ++a)                                                             \
                        for (int b = 0; b < g->Ax; ++b) {                                                       \
                            const double wv = (double)w[a * g->Ax + b];                                         \
                            /* u + a - (Ay-1) = y in [0,Dy)  ->  u = y + Ay-1-a ; same for v */                 \
                            for (int y = 0; y < g->Dy; ++y) {                                                   \
                                const T *xr = xp + (size_t)y * g->Dx;                                           \
                                double *ar = acc + (size_t)(y + g->Ay - 1 - a) * Hx + (g->Ax - 1 - b);          \
                                for (int x = 0; x < g->Dx; ++x) ar[x] += wv * (double)xr[x];                    \
                            }                                                                                   \
                        }                                                                                       \
                }                                                                                               \
                T *o = O + ((size_t)n * g->M + m) * Hy * Hx;                                                    \
                for (size_t i = 0; i < (size_t)Hy * Hx; ++i) o[i] = (T)acc[i];                                  \
                free(acc);                                                                                      \
            }                                                                                                   \
    }                                                                                                           \
                                                                                                                \
    void oracle_corr_H_with_##SUF(const geom_t *g, const T *H, const T *X, T *G) {                              \
        const int Hy = HY(g), Hx = HX(g);                                                                       \
        _Pragma("omp parallel for collapse(2) schedule(static)")                                                \
        for (int m = 0; m < g->M; ++m)                                                                          \
            for (int c = 0; c < g->C; ++c)                                                                      \
                for (int a = 0; a < g->Ay; ++a)                                                                 \
                    for (int b = 0; b < g->Ax; ++b) {                                                           \
                        double tot = 0.0;                                                                       \
                        for (int n = 0; n < g->N; ++n) {                                                        \
                            const T *h = H + ((size_t)n * g->M + m) * Hy * Hx;                                  \
                            const T *xp = X + ((size_t)n * g->C + c) * g->Dy * g->Dx;                           \
                            double sn = 0.0;                                                                    \
                            for (int y = 0; y < g->Dy; ++y) {                                                   \
                                const T *hr = h + (size_t)(y + g->Ay - 1 - a) * Hx + (g->Ax - 1 - b);           \
                                const T *xr = xp + (size_t)y * g->Dx;                                           \
                                double s = 0.0;                                                                 \
                                for (int x = 0; x < g->Dx; ++x) s += (double)hr[x] * (double)xr[x];             \
                                sn += s;                                                                        \
                            }                                                                                   \
                            tot += sn;                                                                          \
                        }                                                                                       \
                        G[(((size_t)m * g->C + c) * g->Ay + a) * g->Ax + b] = (T)tot;                           \
                    }                                                                                           \
    }

DEFINE_ORACLE(double, f64)
DEFINE_ORACLE(float, f32)
