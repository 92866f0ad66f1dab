// satba_lm.h -- scalar side of the trust-region loop: scipy's `trf_no_bounds` control flow
// (scipy:optimize/_lsq/trf.py:401-560) and its helpers (scipy:optimize/_lsq/common.py:171-245, 302-322, 705-717), restated
// in C++ so that a caller of the C ABI can solve without the Python loop of satba/trf.py (which stays the reference
// implementation: tests/test_gpu_parity.py runs both on the same problems and compares nfev, status and x).
// Everything here works on the handful of scalars the phases return in the exchange header.
#pragma once
#include <cmath>
#include <cstdint>

// the same scalar code runs on the host (satba_solve_lm's host loop, the tests' view of it) and in the one-thread decision kernels
// of the device-resident loop (satba_lmdev.h)
#ifdef __HIPCC__
#define SATBA_HD __host__ __device__
#else
#define SATBA_HD
#endif

namespace satba_lm {

SATBA_HD inline double norm2(double a, double b) {
#pragma clang fp contract(off)
    return sqrt(a * a + b * b);
}

// min 0.5 p^T B p + g^T p  s.t. |p| <= Delta in two dimensions (scipy common.py:171-219): the Newton step when B is positive
// definite and the step fits, otherwise the minimiser on the boundary, found as the root of the secular equation
// |(B + mu I)^-1 g| = Delta, mu >= max(0, -lambda_min(B)), in the eigenbasis of B with a safeguarded Newton iteration on
// 1 / |p(mu)| - 1 / Delta (scipy takes the real roots of a quartic: same point).  Mirrors satba/trf.py:solve_trust_region_2d.
SATBA_HD inline bool solve_trust_region_2d(double a, double b, double c, double g0, double g1, double Delta, double& p0, double& p1) {
    // no fused multiply-adds and only correctly rounded operations (+ - * / sqrt): the host loop and the device's decision kernels
    // then take bit-identical decisions
#pragma clang fp contract(off)
    const double det = a * c - b * b;
    if (a > 0 && det > 0) {
        p0 = -(c * g0 - b * g1) / det;
        p1 = -(a * g1 - b * g0) / det;
        if (p0 * p0 + p1 * p1 <= Delta * Delta) return true;
    }
    const double h = 0.5 * (a + c), d = 0.5 * (a - c);
    const double r = norm2(d, b);
    const double l1 = h - r, l2 = h + r;
    double v1x, v1y;
    if (r == 0.0) { v1x = 1.0; v1y = 0.0; }
    else if (d > 0) { const double n = norm2(d + r, b); v1x = -b / n; v1y = (d + r) / n; }
    else { const double n = norm2(d - r, b); v1x = (d - r) / n; v1y = b / n; }
    const double v2x = -v1y, v2y = v1x;
    const double c1 = g0 * v1x + g1 * v1y, c2 = g0 * v2x + g1 * v2y;
    const double gn = norm2(c1, c2);
    if (gn == 0.0) { p0 = Delta * v1x; p1 = Delta * v1y; return false; }
    const double lo = fmax(0.0, -l1);
    const double tiny = 1e-14 * gn;
    if (fabs(c1) <= tiny && l2 + lo > 0 && fabs(c2) / (l2 + lo) < Delta) {
        const double q2 = -c2 / (l2 + lo);
        const double q1 = sqrt(fmax(Delta * Delta - q2 * q2, 0.0));
        p0 = q1 * v1x + q2 * v2x; p1 = q1 * v1y + q2 * v2y;
        return false;
    }
    double mu_lo = lo, mu_hi = fmax(lo, gn / Delta - l1);
    double mu = mu_hi;
    for (int it = 0; it < 100; ++it) {
        const double d1 = l1 + mu, d2 = l2 + mu;
        if (d1 <= 0.0) { mu = 0.5 * (mu_lo + mu_hi); continue; }
        const double q1 = c1 / d1, q2 = c2 / d2;
        const double n2 = q1 * q1 + q2 * q2;
        const double nrm = sqrt(n2);
        if (nrm > Delta) mu_lo = mu; else mu_hi = mu;
        if (fabs(nrm - Delta) <= 4e-16 * Delta) break;
        const double dphi = (q1 * q1 / d1 + q2 * q2 / d2) / (n2 * nrm);
        const double step = (1.0 / nrm - 1.0 / Delta) / dphi;
        double nw = mu - step;
        if (!(mu_lo <= nw && nw <= mu_hi) || nw == mu) {
            nw = 0.5 * (mu_lo + mu_hi);
            if (nw == mu_lo || nw == mu_hi) break;
        }
        mu = nw;
    }
    const double d1 = l1 + mu, d2 = l2 + mu;
    double q1 = -c1 / d1, q2 = -c2 / d2;
    const double s = Delta / norm2(q1, q2);
    q1 *= s; q2 *= s;
    p0 = q1 * v1x + q2 * v2x; p1 = q1 * v1y + q2 * v2y;
    return false;
}

// scipy common.py:222-245
SATBA_HD inline double update_tr_radius(double Delta, double actual, double predicted, double step_norm, bool bound_hit, double& ratio) {
#pragma clang fp contract(off)
    if (predicted > 0) ratio = actual / predicted;
    else if (predicted == 0 && actual == 0) ratio = 1;
    else ratio = 0;
    if (ratio < 0.25) Delta = 0.25 * step_norm;
    else if (ratio > 0.75 && bound_hit) Delta *= 2.0;
    return Delta;
}

// scipy common.py:705-717; 0: none
SATBA_HD inline int check_termination(double dF, double F, double dx_norm, double x_norm, double ratio, double ftol, double xtol) {
#pragma clang fp contract(off)
    const bool f_ok = dF < ftol * F && ratio > 0.25;
    const bool x_ok = dx_norm < xtol * (xtol + x_norm);
    if (f_ok && x_ok) return 4;
    if (f_ok) return 2;
    if (x_ok) return 3;
    return 0;
}

}  // namespace satba_lm
