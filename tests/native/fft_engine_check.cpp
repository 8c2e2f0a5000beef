// Host check of tnmf_amd/csrc/fft_engine.h: the stage functions the HIP kernels run per thread are executed here
// task by task and compared with a naive O(L^2) DFT in double.  Built and run by tests/test_fft_engine_cpu.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft_engine.h"

static double urand() { return rand() / (double)RAND_MAX - 0.5; }

template <typename T, int L>
static double check_len() {
    using P = FftPlanFor<T, L>;
    constexpr int NB = 3, BS = 5;
    std::vector<cplx<T>> x(L * BS), x0;
    std::vector<cplx<T>> tw(L);
    for (int t = 0; t < L; ++t) tw[t] = {(T)cos(2 * M_PI * t / L), (T)(-sin(2 * M_PI * t / L))};
    for (auto &v : x) v = {(T)urand(), (T)urand()};
    x0 = x;
    double worst = 0;
    for (int pos = 0; pos < L; ++pos)
        if (P::pos_of_k(P::k_of_pos(pos)) != pos) return 1e9;
    // forward
    for (int b = 0; b < NB; ++b) {
        for (int t = 0; t < P::tasks1; ++t) P::template fwd1<BS>(&x[b], tw.data(), t);
        for (int t = 0; t < P::tasks2; ++t) P::template fwd2<BS>(&x[b], tw.data(), t);
        for (int t = 0; t < P::tasks3; ++t) P::template fwd3<BS>(&x[b], t);
        for (int k = 0; k < L; ++k) {
            double re = 0, im = 0;
            for (int n = 0; n < L; ++n) {
                const double a = -2 * M_PI * ((long)n * k % L) / L;
                re += x0[n * BS + b].x * cos(a) - x0[n * BS + b].y * sin(a);
                im += x0[n * BS + b].x * sin(a) + x0[n * BS + b].y * cos(a);
            }
            const cplx<T> got = x[P::pos_of_k(k) * BS + b];
            worst = fmax(worst, fmax(fabs(got.x - re), fabs(got.y - im)) / sqrt((double)L));
        }
        // inverse
        for (int t = 0; t < P::tasks3; ++t) P::template inv3<BS>(&x[b], t);
        for (int t = 0; t < P::tasks2; ++t) P::template inv2<BS>(&x[b], tw.data(), t);
        for (int t = 0; t < P::tasks1; ++t) P::template inv1<BS>(&x[b], tw.data(), t);
        for (int n = 0; n < L; ++n) {
            worst = fmax(worst, fabs(x[n * BS + b].x / L - x0[n * BS + b].x));
            worst = fmax(worst, fabs(x[n * BS + b].y / L - x0[n * BS + b].y));
        }
    }
    // two real rows as one complex sequence: split after forward, merge before inverse
    std::vector<double> ra(L), rb(L);
    std::vector<cplx<T>> z(L), za(L / 2 + 1), zb(L / 2 + 1);
    for (int n = 0; n < L; ++n) {
        ra[n] = n < L - 7 ? urand() : 0;
        rb[n] = n < L - 7 ? urand() : 0;
        z[n] = {(T)ra[n], (T)rb[n]};
    }
    for (int t = 0; t < P::tasks1; ++t) P::template fwd1<1>(z.data(), tw.data(), t);
    for (int t = 0; t < P::tasks2; ++t) P::template fwd2<1>(z.data(), tw.data(), t);
    for (int t = 0; t < P::tasks3; ++t) P::template fwd3<1>(z.data(), t);
    for (int k = 0; k <= L / 2; ++k) {
        split_pair(z[P::pos_of_k(k)], z[P::pos_of_k((L - k) % L)], za[k], zb[k]);
        double are = 0, aim = 0, bre = 0, bim = 0;
        for (int n = 0; n < L; ++n) {
            const double a = -2 * M_PI * ((long)n * k % L) / L;
            are += ra[n] * cos(a); aim += ra[n] * sin(a);
            bre += rb[n] * cos(a); bim += rb[n] * sin(a);
        }
        worst = fmax(worst, fmax(fabs(za[k].x - are), fabs(za[k].y - aim)) / sqrt((double)L));
        worst = fmax(worst, fmax(fabs(zb[k].x - bre), fabs(zb[k].y - bim)) / sqrt((double)L));
    }
    for (int k = 0; k <= L / 2; ++k) {
        cplx<T> zk, zlk;
        if (k == 0 || 2 * k == L) {
            z[P::pos_of_k(k)] = {za[k].x, zb[k].x};
        } else {
            merge_pair(za[k], zb[k], zk, zlk);
            z[P::pos_of_k(k)] = zk;
            z[P::pos_of_k(L - k)] = zlk;
        }
    }
    for (int t = 0; t < P::tasks3; ++t) P::template inv3<1>(z.data(), t);
    for (int t = 0; t < P::tasks2; ++t) P::template inv2<1>(z.data(), tw.data(), t);
    for (int t = 0; t < P::tasks1; ++t) P::template inv1<1>(z.data(), tw.data(), t);
    for (int n = 0; n < L; ++n) {
        worst = fmax(worst, fabs(z[n].x / L - ra[n]));
        worst = fmax(worst, fabs(z[n].y / L - rb[n]));
    }
    return worst;
}

template <int L>
static int report() {
    const double ef = check_len<float, L>(), ed = check_len<double, L>();
    printf("L=%d  float err %.3e  double err %.3e\n", L, ef, ed);
    return (ef < 2e-6 && ed < 1e-14) ? 0 : 1;
}

int main() {
    srand(7);
    int bad = 0;
    bad += report<32>();
    bad += report<48>();
    bad += report<64>();
    bad += report<96>();
    bad += report<144>();
    bad += report<192>();
    bad += report<270>();
    bad += report<288>();
    bad += report<384>();
    bad += report<540>();
    bad += report<576>();
    printf(bad ? "FAILED\n" : "OK\n");
    return bad;
}
