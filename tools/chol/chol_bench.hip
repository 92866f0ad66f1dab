// Stand-alone check + timing of the dense solve (csrc/satba_chol*.h) without the Python side:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I ../../sat-bundleadjust_amd/csrc chol_bench.hip -o chol_bench
//   ./chol_bench [reps] n1 n2 ...        (SATBA_STAMPS=1: per-step time stamps of the persistent kernel)
// For every n: a random SPD system, the host's Cholesky as the reference, then factor / forward substitution / L^T / block inverses /
// solution of the persistent tile kernel against it ("tiles": the tile kernel at every size; "driver": cholesky_solve as the library
// calls it, panel steps up to 64 unknowns), the not-SPD flag, and the time per solve (HIP events around the launches only).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "satba_chol.h"
using namespace satba;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static double urand(unsigned long long& s) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return ((s >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0; }

int main(int argc, char** argv) {
    int reps = argc > 1 ? atoi(argv[1]) : 20;
    std::vector<int> sizes;
    for (int i = 2; i < argc; ++i) sizes.push_back(atoi(argv[i]));
    if (sizes.empty()) sizes = {1000};
    const bool stamps = getenv("SATBA_STAMPS") != nullptr;
    const bool check_all = getenv("SATBA_CHECK_ALL") != nullptr;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int worst = 0;
    for (int n : sizes) {
        const int T = (n + 63) / 64;
        unsigned long long seed = 1234567ULL + n;
        // A = B B^T / n + I  (B: n x min(n, 96) random: rank-deficient part lifted by the identity; plenty of off-diagonal weight)
        const int kb = std::min(n, 96);
        std::vector<double> B((size_t)n * kb), A((size_t)n * n, 0.0), b(n);
        for (auto& v : B) v = urand(seed);
        for (int c = 0; c < n; ++c)
            for (int r = c; r < n; ++r) {
                double s = 0.0;
                for (int k = 0; k < kb; ++k) s += B[(size_t)r * kb + k] * B[(size_t)c * kb + k];
                A[(size_t)r + (size_t)c * n] = s / kb + (r == c ? 1.0 : 0.0);
            }
        for (auto& v : b) v = urand(seed);
        // strict upper triangle: garbage on purpose (the kernels must not read it)
        for (int c = 1; c < n; ++c) for (int r = 0; r < c; ++r) A[(size_t)r + (size_t)c * n] = 1e30;
        // host reference
        std::vector<double> L(A), y(b), z(n);
        for (int j = 0; j < n; ++j) {
            double d = L[(size_t)j + (size_t)j * n];
            for (int k = 0; k < j; ++k) d -= L[(size_t)j + (size_t)k * n] * L[(size_t)j + (size_t)k * n];
            d = std::sqrt(d);
            L[(size_t)j + (size_t)j * n] = d;
            for (int r = j + 1; r < n; ++r) {
                double s = L[(size_t)r + (size_t)j * n];
                for (int k = 0; k < j; ++k) s -= L[(size_t)r + (size_t)k * n] * L[(size_t)j + (size_t)k * n];
                L[(size_t)r + (size_t)j * n] = s / d;
            }
        }
        for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[(size_t)i + (size_t)k * n] * y[k]; y[i] = s / L[(size_t)i + (size_t)i * n]; }
        for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < n; ++k) s -= L[(size_t)k + (size_t)i * n] * z[k]; z[i] = s / L[(size_t)i + (size_t)i * n]; }

        double *dA0, *dA, *db0, *db, *dinv;
        int* dfail;
        CK(hipMalloc(&dA0, sizeof(double) * n * n)); CK(hipMalloc(&dA, sizeof(double) * n * n));
        CK(hipMalloc(&db0, sizeof(double) * n)); CK(hipMalloc(&db, sizeof(double) * n));
        CK(hipMalloc(&dinv, sizeof(double) * ((n + 31) / 32) * 1024));
        CK(hipMalloc(&dfail, sizeof(int) * (1 + CH_MAX_STEPS)));
        CK(hipMemcpy(dA0, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
        CK(hipMemcpy(db0, b.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        CholWork w;
        if (chol_work_alloc(w, n) != hipSuccess) { printf("alloc failed\n"); return 2; }
        long long* dts = nullptr;
        if (stamps) { CK(hipMalloc(&dts, sizeof(long long) * C3_TS * T)); CK(hipMemset(dts, 0, sizeof(long long) * C3_TS * T)); }
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        int bad_reps = 0;
        auto run = [&](int which, std::vector<float>& times) {
            for (int r = 0; r < reps; ++r) {
                CK(hipMemcpyAsync(dA, dA0, sizeof(double) * n * n, hipMemcpyDeviceToDevice, st));
                CK(hipMemcpyAsync(db, db0, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
                CK(hipMemsetAsync(dfail, 0, sizeof(int) * (1 + CH_MAX_STEPS), st));
                CK(hipEventRecord(e0, st));
                if (which == 0) {   // the tile kernel and the backward substitution, launched by hand (time stamps)
                    cholesky_init();
                    const bool mw = n <= 1024;
                    cholesky_tiles(dA, n, db, dfail, st, w, mw ? dinv : nullptr, mw, nullptr, dts);
                    if (mw) hipLaunchKernelGGL(k_trsv_back_mw, dim3((n + CH_SB - 1) / CH_SB), dim3(512), 0, st, dA, dinv, n, db, dfail + 1 + CH_TRSV_FLAGS, (const int*)nullptr);
                    else hipLaunchKernelGGL(k_trsv_back, dim3(1), dim3(1024), sizeof(double) * n, st, dA, n, db);
                } else cholesky_solve(dA, n, db, dfail, dfail + 1, st, w, dinv, true, nullptr);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                times.push_back(ms);
                if (check_all) {  // every repetition's solution, not only the last one's (a rare hand-over race shows as one bad solve in many)
                    std::vector<double> zr(n);
                    CK(hipMemcpy(zr.data(), db, sizeof(double) * n, hipMemcpyDeviceToHost));
                    double ez = 0.0, nz = 0.0; int wi = -1;
                    for (int i = 0; i < n; ++i) { const double d = std::fabs(zr[i] - z[i]); if (!(d <= ez)) { ez = d; wi = i; } nz = std::max(nz, std::fabs(z[i])); }
                    if (!(ez <= 1e-10 * nz)) {
                        ++bad_reps; worst = 1;
                        if (bad_reps <= 5) {
                            printf("   n %d %s rep %d: |dz|/|z| %.2e worst row %d; per 64-row block:", n, which ? "driver" : "tiles", r, ez / nz, wi);
                            for (int b0 = 0; b0 < n; b0 += 64) { double e = 0.0; for (int i = b0; i < std::min(n, b0 + 64); ++i) e = std::max(e, std::fabs(zr[i] - z[i])); printf(" %.1e", e / nz); }
                            // which of the factor's parts is off: L (lower), L^T (upper), the 32 x 32 inverses
                            std::vector<double> Lr((size_t)n * n);
                            CK(hipMemcpy(Lr.data(), dA, sizeof(double) * n * n, hipMemcpyDeviceToHost));
                            for (int tj = 0; tj * 64 < n; ++tj) for (int ti = 0; ti * 64 < n; ++ti) {
                                double e = 0.0;
                                for (int c = 64 * tj; c < std::min(n, 64 * tj + 64); ++c) for (int rr = 64 * ti; rr < std::min(n, 64 * ti + 64); ++rr) {
                                    if (rr == c) continue;
                                    const double want = rr > c ? L[(size_t)rr + (size_t)c * n] : L[(size_t)c + (size_t)rr * n];  // upper = L^T
                                    e = std::max(e, std::fabs(Lr[(size_t)rr + (size_t)c * n] - want));
                                }
                                if (e > 1e-9) {
                                    printf(" tile(%d,%d) off by %.1e", ti, tj, e);
                                    int nbad = 0, shown = 0;
                                    for (int c = 64 * tj; c < std::min(n, 64 * tj + 64); ++c) for (int rr = 64 * ti; rr < std::min(n, 64 * ti + 64); ++rr) {
                                        if (rr == c) continue;
                                        const double want = rr > c ? L[(size_t)rr + (size_t)c * n] : L[(size_t)c + (size_t)rr * n];
                                        const double got = Lr[(size_t)rr + (size_t)c * n];
                                        if (std::fabs(got - want) > 1e-9) {
                                            ++nbad;
                                            if (shown < 3) { ++shown; printf(" [(%d,%d) got %.6f want %.6f input(lower twin) %.6f]", rr, c, got, want, A[(size_t)c + (size_t)rr * n]); }
                                        }
                                    }
                                    printf(" %d entries", nbad);
                                }
                            }
                            printf("\n");
                        }
                    }
                }
            }
        };
        {   // the tile kernel alone: forward-substituted right-hand side, mirror and 32 x 32 block inverses against the host
            CK(hipMemcpyAsync(dA, dA0, sizeof(double) * n * n, hipMemcpyDeviceToDevice, st));
            CK(hipMemcpyAsync(db, db0, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
            CK(hipMemsetAsync(dfail, 0, sizeof(int) * (1 + CH_MAX_STEPS), st));
            cholesky_init();
            C3Args g;
            g.A = dA; g.n = n; g.b = db; g.fail = dfail; g.flags = w.flags; g.epoch = ++w.epoch; g.Linv = w.Linv; g.Cc = w.Cc; g.ctr = w.ctr;
            g.dinv = dinv; g.ts = nullptr; g.mirror = getenv("SATBA_NO_MIRROR") ? 0 : 1;
            hipLaunchKernelGGL(k_chol_tiles, dim3(chol_tiles_grid(n, g.mirror)), dim3(1024), c3_lds_bytes(), st, g, (const int*)nullptr);
            CK(hipStreamSynchronize(st));
            std::vector<double> Lg((size_t)n * n), yg(n), dg((size_t)((n + 31) / 32) * 1024);
            CK(hipMemcpy(Lg.data(), dA, sizeof(double) * n * n, hipMemcpyDeviceToHost));
            CK(hipMemcpy(yg.data(), db, sizeof(double) * n, hipMemcpyDeviceToHost));
            CK(hipMemcpy(dg.data(), dinv, sizeof(double) * dg.size(), hipMemcpyDeviceToHost));
            double ey = 0.0, ny = 0.0, em = 0.0, ed = 0.0;
            int wy = -1;
            for (int i = 0; i < n; ++i) { const double d = std::fabs(yg[i] - y[i]); if (!(d <= ey)) { ey = d; wy = i; } ny = std::max(ny, std::fabs(y[i])); }
            int mr = -1, mc = -1, nbad_m = 0;
            for (int c = 0; c < n; ++c) for (int r = c + 1; r < n; ++r) { const double d = std::fabs(Lg[(size_t)c + (size_t)r * n] - L[(size_t)r + (size_t)c * n]); if (d > 1e-11) ++nbad_m; if (!(d <= em)) { em = d; mr = r; mc = c; } }
            if (em > 1e-11) printf("   mirror: %d entries off, worst: upper (%d,%d) holds %.6f, L(%d,%d) = %.6f, input there %.6f, L as stored below %.6f\n", nbad_m, mc, mr,
                                   Lg[(size_t)mc + (size_t)mr * n], mr, mc, L[(size_t)mr + (size_t)mc * n], A[(size_t)mc + (size_t)mr * n], Lg[(size_t)mr + (size_t)mc * n]);
            for (int kb = 0; kb * 32 < n; ++kb) {   // D_kb * dinv_kb = I
                const int nb = std::min(32, n - 32 * kb);
                for (int r = 0; r < nb; ++r) for (int c = 0; c < nb; ++c) {
                    double s = 0.0;
                    for (int m = 0; m < nb; ++m) s += L[(size_t)(32 * kb + r) + (size_t)(32 * kb + m) * n] * dg[((size_t)kb * 32 + m) * 32 + c] * (m <= r ? 1.0 : 0.0);
                    const double d = std::fabs(s - (r == c ? 1.0 : 0.0)); if (!(d <= ed)) ed = d;
                }
            }
            printf("n %5d kernel alone: |dy|/|y| %.2e (worst row %d)  mirror max err %.2e  |D dinv - I| %.2e\n", n, ey / ny, wy, em, ed);
            if (!(ey <= 1e-11 * ny) || !(em <= 1e-11) || !(ed <= 1e-11)) worst = 1;
        }
        for (int which = 0; which < 2; ++which) {
            std::vector<float> times;
            run(which, times);
            CK(hipGetLastError());
            std::vector<double> Lg((size_t)n * n), zg(n);
            int fail = 0;
            CK(hipMemcpy(Lg.data(), dA, sizeof(double) * n * n, hipMemcpyDeviceToHost));
            CK(hipMemcpy(zg.data(), db, sizeof(double) * n, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&fail, dfail, sizeof(int), hipMemcpyDeviceToHost));
            double eL = 0.0, nL = 0.0, ez = 0.0, nz = 0.0;
            int wr = -1, wc = -1; double we = 0.0;
            for (int c = 0; c < n; ++c)
                for (int r = c; r < n; ++r) {
                    const double d = Lg[(size_t)r + (size_t)c * n] - L[(size_t)r + (size_t)c * n];
                    if (!(std::fabs(d) <= we)) { we = std::fabs(d); wr = r; wc = c; }
                    eL = std::max(eL, std::fabs(d)); nL = std::max(nL, std::fabs(L[(size_t)r + (size_t)c * n]));
                }
            for (int i = 0; i < n; ++i) { ez = std::max(ez, std::fabs(zg[i] - z[i])); nz = std::max(nz, std::fabs(z[i])); }
            if (!(eL <= 1e-11 * nL) || !(ez <= 1e-10 * nz) || fail) worst = 1;
            std::sort(times.begin(), times.end());
            printf("n %5d %s  fail %d  |dL|/|L| %.2e (worst at %d,%d)  |dz|/|z| %.2e   solve us: min %.1f median %.1f max %.1f\n", n,
                   which == 0 ? "tiles " : "driver", fail, eL / nL, wr, wc, ez / nz, times.front() * 1e3, times[times.size() / 2] * 1e3, times.back() * 1e3);
            if (which == 0 && stamps) {
                std::vector<long long> ts((size_t)C3_TS * T);
                CK(hipMemcpy(ts.data(), dts, sizeof(long long) * C3_TS * T, hipMemcpyDeviceToHost));
                const long long t0 = ts[0];
                printf("   step: start | D done  R done  I done  Tb ready  published   (us since the start of step 0; 100 MHz clock)\n");
                for (int k = 0; k < T; ++k) {
                    printf("   %3d:", k);
                    for (int s = 0; s < 6; ++s) printf(" %8.2f", ts[k * C3_TS + s] ? (ts[k * C3_TS + s] - t0) * 0.01 : -1.0);
                    printf("\n");
                    {
                        auto rel = [&](int idx) { return ts[k * C3_TS + idx] ? (ts[k * C3_TS + idx] - ts[k * C3_TS]) * 0.01 : -1.0; };
                        printf("        rel. to start: D p0 %.2f p7 %.2f | R p0 %.2f p7 %.2f | I p0 %.2f p7 %.2f | I end %.2f | D has Tb %.2f | mirror done %.2f | Tb(aux) %.2f | pub %.2f\n", rel(8), rel(15), rel(16), rel(23), rel(24), rel(31), rel(3), rel(6), rel(7), rel(4), rel(5));
                    }
                    if (k == 9) {   // micro-panel flags of the three row sets, relative to the start of the step
                        const char* nm[3] = {"D", "R", "I"};
                        for (int st_ = 0; st_ < 3; ++st_) {
                            printf("        %s micro-panels:", nm[st_]);
                            for (int p = 0; p < 8; ++p) printf(" %6.2f", ts[k * C3_TS + 8 + 8 * st_ + p] ? (ts[k * C3_TS + 8 + 8 * st_ + p] - ts[k * C3_TS]) * 0.01 : -1.0);
                            printf("\n");
                        }
                    }
                }
            }
        }
        // not positive definite: the flag, no hang
        {
            std::vector<double> A2(A);
            A2[(size_t)(n / 2) + (size_t)(n / 2) * n] = -1.0;
            CK(hipMemcpy(dA, A2.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
            CK(hipMemcpy(db, b.data(), sizeof(double) * n, hipMemcpyHostToDevice));
            CK(hipMemsetAsync(dfail, 0, sizeof(int) * (1 + CH_MAX_STEPS), st));
            cholesky_solve(dA, n, db, dfail, dfail + 1, st, w, dinv, true, nullptr);
            CK(hipStreamSynchronize(st));
            int fail = 0;
            CK(hipMemcpy(&fail, dfail, sizeof(int), hipMemcpyDeviceToHost));
            printf("n %5d not-SPD input: fail flag %d (expect 1)\n", n, fail);
            if (fail != 1) worst = 1;
        }
        chol_work_free(w);
        CK(hipFree(dA0)); CK(hipFree(dA)); CK(hipFree(db0)); CK(hipFree(db)); CK(hipFree(dinv)); CK(hipFree(dfail));
        if (dts) CK(hipFree(dts));
    }
    printf(worst ? "FAILED\n" : "all ok\n");
    return worst;
}
