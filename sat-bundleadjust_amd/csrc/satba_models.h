// satba_models.h -- device-side camera models: projection and analytic Jacobian blocks of one observation.
//
// Math: SURVEY.md appendix B.  Reference statements of the three projections:
//   affine       ref:bundle_adjust/ba_core.py:59-81
//   perspective  ref:bundle_adjust/ba_core.py:84-107
//   rpc          ref:bundle_adjust/ba_core.py:110-154 -> ref:bundle_adjust/cam_utils.py:217-231
//                -> ref:bundle_adjust/geo_utils.py:236-255 -> rpcm.RPCModel.projection
//                (polynomial order ref:bundle_adjust/ba_rpcfit.py:17-44, ref:c/rpc.c:279-298)
// The reference differentiates these numerically (scipy 2-point differences); the derivatives here are
// analytic and are validated against 3-point differences of the reference's `fun` (tests/golden/fun_*.npz).
//
// Rotation convention R = Rz(g) Ry(b) Rx(a) (ref:bundle_adjust/ba_rotate.py:85-94), applied as three plane
// rotations in the order of ref:bundle_adjust/ba_core.py:47-55 so the residuals round like the reference's.
#pragma once
#include <hip/hip_runtime.h>

namespace satba {

enum { AFFINE = 0, PERSPECTIVE = 1, RPC = 2 };

// per-camera constant record, rebuilt from x before every pass over the observations
//   [0..5]   cos a, sin a, cos b, sin b, cos g, sin g
//   [6..14]  R row-major
//   [15]     pad
//   [16..]   affine: t0 t1 fx fy skew | perspective: t0 t1 t2 fx fy skew cx cy | rpc: T(3) C(3)
// Rows are 13 sixteen-byte slots: every pair (2 k, 2 k + 1) of a row is one aligned ds_read_b128 (256 B / clk, four 16-lane groups),
// and an odd slot count starts the rows of consecutive cameras in all 16 slots of the LDS's 256-byte bank row.  Round 6: with 25
// doubles per row (8-byte alignment) the compiler paired the reads as ds_read2_b64 -- half the rate and banks modulo 32: the gathers
// of a wave's ~50 different cameras were most of the LDS time of the lane = point kernels.
constexpr int CAMC = 26;
constexpr int CAMX = 16;  // first model-specific entry
static_assert(CAMC % 2 == 0 && (CAMC / 2) % 2 == 1 && CAMX % 2 == 0 && CAMX + 8 <= CAMC, "camera records: an odd number of 16-byte slots, model entries on a slot boundary");

constexpr double WGS84_A = 6378137.0;
constexpr double WGS84_E = 8.1819190842622e-2;  // as hard-coded at ref:bundle_adjust/geo_utils.py:241

__device__ inline void cam_constants(int model, const double* full, double* cc) {
    double sa, ca, sb, cb, sg, cg;
    sincos(full[0], &sa, &ca);
    sincos(full[1], &sb, &cb);
    sincos(full[2], &sg, &cg);
    cc[0] = ca; cc[1] = sa; cc[2] = cb; cc[3] = sb; cc[4] = cg; cc[5] = sg;
    cc[6] = cg * cb;  cc[7] = cg * sb * sa - sg * ca;   cc[8] = cg * sb * ca + sg * sa;
    cc[9] = sg * cb;  cc[10] = sg * sb * sa + cg * ca;  cc[11] = sg * sb * ca - cg * sa;
    cc[12] = -sb;     cc[13] = cb * sa;                 cc[14] = cb * ca;
    const int n_extra = model == AFFINE ? 5 : (model == PERSPECTIVE ? 8 : 6);
    for (int i = 0; i < n_extra; ++i) cc[CAMX + i] = full[3 + i];
    cc[15] = 0.0;
    for (int i = CAMX + n_extra; i < CAMC; ++i) cc[i] = 0.0;
}

// y3 = Rz Ry Rx X and the three angle derivatives of y3
struct Rot {
    double y3[3], da[3], db[3], dg[3];
};

template <bool JAC>
__device__ inline void rotate(const double* __restrict__ cc, double X, double Y, double Z, Rot& r) {
    const double ca = cc[0], sa = cc[1], cb = cc[2], sb = cc[3], cg = cc[4], sg = cc[5];
    const double y1y = ca * Y - sa * Z, y1z = sa * Y + ca * Z;
    const double y2x = cb * X + sb * y1z, y2z = -sb * X + cb * y1z;
    r.y3[0] = cg * y2x - sg * y1y;
    r.y3[1] = sg * y2x + cg * y1y;
    r.y3[2] = y2z;
    if (JAC) {
        // d/da: Rz Ry (0, -y1z, y1y)
        const double ax = sb * y1y, az = cb * y1y;
        r.da[0] = cg * ax + sg * y1z;
        r.da[1] = sg * ax - cg * y1z;
        r.da[2] = az;
        // d/db: Rz (y2z, 0, -y2x)
        r.db[0] = cg * y2z;
        r.db[1] = sg * y2z;
        r.db[2] = -y2x;
        // d/dg: (-y3y, y3x, 0)
        r.dg[0] = -r.y3[1];
        r.dg[1] = r.y3[0];
        r.dg[2] = 0.0;
    }
}

// (lat_deg, lon_deg, alt) of an ECEF point by the closed form of ref:bundle_adjust/geo_utils.py:236-255 and,
// if JAC, G[i][j] = d out_i / d X_j obtained by differentiating exactly that formula.
template <bool JAC>
__device__ inline void geodetic(double x, double y, double z, double out[3], double G[3][3]) {
    const double a = WGS84_A, esq = WGS84_E * WGS84_E;
    const double b = sqrt(a * a * (1.0 - esq));
    const double ep2 = (a * a - b * b) / (b * b);
    const double rad2deg = 180.0 / M_PI;
    const double p2 = x * x + y * y;
    const double p = sqrt(p2);
    const double u = a * z, w = b * p;
    // th = atan2(u, w) and lat = atan2(num, den) only enter through their sines and cosines (plus lat itself):
    // take those from the triangle sides instead of atan2 + sincos (2 atan2 instead of 3, no sincos)
    const double ihw = 1.0 / sqrt(u * u + w * w);
    const double s = u * ihw, c = w * ihw;
    const double num = z + ep2 * b * s * s * s;
    const double den = p - esq * a * c * c * c;
    const double lat = atan2(num, den);
    const double lon = atan2(y, x);
    const double ihl = 1.0 / sqrt(num * num + den * den);
    const double sl = num * ihl, cl = den * ihl;
    const double t = 1.0 - esq * sl * sl;
    const double rt = sqrt(t);
    const double Nv = a / rt;
    out[0] = lat * rad2deg;
    out[1] = lon * rad2deg;
    out[2] = p / cl - Nv;
    if (JAC) {
        const double ip = 1.0 / p;
        const double dp[3] = {x * ip, y * ip, 0.0};
        const double iuw = 1.0 / (u * u + w * w);
        const double kn = ep2 * b * 3.0 * s * s * c, kd = esq * a * 3.0 * c * c * s;
        const double ind = 1.0 / (num * num + den * den);
        const double kN = a * esq * sl * cl / (t * rt);
        const double kalt = p * sl / (cl * cl);
        for (int j = 0; j < 3; ++j) {
            const double du = (j == 2) ? a : 0.0;
            const double dth = (w * du - u * b * dp[j]) * iuw;
            const double dnum = ((j == 2) ? 1.0 : 0.0) + kn * dth;
            const double dden = dp[j] + kd * dth;
            const double dlat = (den * dnum - num * dden) * ind;
            G[0][j] = dlat * rad2deg;
            G[2][j] = dp[j] / cl + kalt * dlat - kN * dlat;
        }
        const double ip2 = 1.0 / p2;
        G[1][0] = -y * ip2 * rad2deg;
        G[1][1] = x * ip2 * rad2deg;
        G[1][2] = 0.0;
    }
}

// 20-term cubic in RPC00B order (L = lon, P = lat, H = alt, all normalised)
//   1, L, P, H, LP, LH, PH, L2, P2, H2, PLH, L3, LP2, LH2, L2P, P3, PH2, L2H, P2H, H3
// and its gradient, as straight-line code on ten shared products: no monomial arrays (the array form cost 160
// VGPRs and pushed every RPC kernel into scratch).
template <bool JAC>
__device__ inline void rpc_poly(const double* __restrict__ c, double L, double P, double H, double LL, double PP, double HH,
                                double LP, double LH, double PH, double& val, double& dL, double& dP, double& dH) {
    val = c[0] + c[1] * L + c[2] * P + c[3] * H + c[4] * LP + c[5] * LH + c[6] * PH + c[7] * LL + c[8] * PP + c[9] * HH +
          c[10] * LP * H + c[11] * LL * L + c[12] * L * PP + c[13] * L * HH + c[14] * LL * P + c[15] * PP * P +
          c[16] * P * HH + c[17] * LL * H + c[18] * PP * H + c[19] * HH * H;
    if (JAC) {
        dL = c[1] + c[4] * P + c[5] * H + 2.0 * c[7] * L + c[10] * PH + 3.0 * c[11] * LL + c[12] * PP + c[13] * HH +
             2.0 * c[14] * LP + 2.0 * c[17] * LH;
        dP = c[2] + c[4] * L + c[6] * H + 2.0 * c[8] * P + c[10] * LH + 2.0 * c[12] * LP + c[14] * LL + 3.0 * c[15] * PP +
             c[16] * HH + 2.0 * c[18] * PH;
        dH = c[3] + c[5] * L + c[6] * P + 2.0 * c[9] * H + c[10] * LP + 2.0 * c[13] * LH + 2.0 * c[16] * PH + c[17] * LL +
             c[18] * PP + 3.0 * c[19] * HH;
    }
}

// projection of an already-adjusted ECEF point through one RPC record; D = d(col,row)/dX' if JAC
template <bool JAC>
__device__ inline void rpc_project(const double* __restrict__ tab, double x, double y, double z, double& col,
                                   double& row, double D[2][3]) {
    double geo[3], G[3][3];
    geodetic<JAC>(x, y, z, geo, G);
    const double ilon = 1.0 / tab[81], ilat = 1.0 / tab[83], ialt = 1.0 / tab[85];
    const double L = (geo[1] - tab[80]) * ilon;
    const double P = (geo[0] - tab[82]) * ilat;
    const double H = (geo[2] - tab[84]) * ialt;
    const double LL = L * L, PP = P * P, HH = H * H, LP = L * P, LH = L * H, PH = P * H;
    // col = tab[0..19] / tab[20..39]; row = tab[40..59] / tab[60..79]
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        double n, nL, nP, nH, d, dL, dP, dH;
#ifdef SATBA_RPC_FENCES
        asm volatile("" ::: "memory");
#endif
        rpc_poly<JAC>(tab + 40 * k, L, P, H, LL, PP, HH, LP, LH, PH, n, nL, nP, nH);
#ifdef SATBA_RPC_FENCES
        asm volatile("" ::: "memory");
#endif
        rpc_poly<JAC>(tab + 40 * k + 20, L, P, H, LL, PP, HH, LP, LH, PH, d, dL, dP, dH);
#ifdef SATBA_RPC_FENCES
        asm volatile("" ::: "memory");
#endif
        const double id = 1.0 / d;
        const double q = n * id;
        const double scale = tab[87 + 2 * k], off = tab[86 + 2 * k];
        (k == 0 ? col : row) = q * scale + off;
        if (JAC) {
            // derivative w.r.t. (lat_deg, lon_deg, alt), then chain through G
            const double gP = scale * (nP - q * dP) * id * ilat;
            const double gL = scale * (nL - q * dL) * id * ilon;
            const double gH = scale * (nH - q * dH) * id * ialt;
            for (int j = 0; j < 3; ++j) D[k][j] = gP * G[0][j] + gL * G[1][j] + gH * G[2][j];
        }
    }
}

// RPC: Jc and Jp of one observation from D = d(col,row)/dX' (2 x 3), the camera's rotation and the point -- the tail of the chain
// (ref:bundle_adjust/ba_core.py:126-130 differentiated): Jc[:, i] = D dR/dtheta_i (X - T - C), Jp = D R, d/dT = -D R.  Every pass behind
// a linearisation recomputes the blocks from the stored D (round 5: 64 bytes per observation instead of the 128 / 192-byte rows Jc | Jp).
template <int NP>
__device__ inline void rpc_jac_from_d(const double* __restrict__ cc, double X, double Y, double Z, const double (&D)[2][3], double Jc[2][NP],
                                      double Jp[2][3]) {
    Rot r;
    rotate<true>(cc, X - cc[CAMX] - cc[CAMX + 3], Y - cc[CAMX + 1] - cc[CAMX + 4], Z - cc[CAMX + 2] - cc[CAMX + 5], r);
    const double* d[3] = {r.da, r.db, r.dg};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Jc[0][i] = D[0][0] * d[i][0] + D[0][1] * d[i][1] + D[0][2] * d[i][2];
        Jc[1][i] = D[1][0] * d[i][0] + D[1][1] * d[i][1] + D[1][2] * d[i][2];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        Jp[0][j] = D[0][0] * cc[6 + j] + D[0][1] * cc[9 + j] + D[0][2] * cc[12 + j];
        Jp[1][j] = D[1][0] * cc[6 + j] + D[1][1] * cc[9 + j] + D[1][2] * cc[12 + j];
    }
    if constexpr (NP == 6) {  // d/dT = -D R
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            Jc[0][3 + j] = -Jp[0][j];
            Jc[1][3 + j] = -Jp[1][j];
        }
    }
}

// RPC projection of one observation; if JAC also D = d(col,row)/dX' and (BLOCKS) the blocks Jc, Jp formed from it
template <int NP, bool JAC, bool BLOCKS = true>
__device__ inline void project_rpc_d(const double* __restrict__ cc, const double* __restrict__ rpc_tab, double X, double Y, double Z, bool f32,
                                     double& u, double& v, double Jc[2][NP], double Jp[2][3], double (&D)[2][3]) {
    // X' = R (X - T - C) + C   (ref:bundle_adjust/ba_core.py:126-130)
    const double C0 = cc[CAMX + 3], C1 = cc[CAMX + 4], C2 = cc[CAMX + 5];
    Rot r;
    rotate<JAC && BLOCKS>(cc, X - cc[CAMX] - C0, Y - cc[CAMX + 1] - C1, Z - cc[CAMX + 2] - C2, r);
    rpc_project<JAC>(rpc_tab, r.y3[0] + C0, r.y3[1] + C1, r.y3[2] + C2, u, v, D);
    if (f32) {  // ref:bundle_adjust/ba_core.py:150 stores the projections in a float32 array
        u = (double)(float)u;
        v = (double)(float)v;
    }
    if (JAC && BLOCKS) {
        const double* d[3] = {r.da, r.db, r.dg};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            Jc[0][i] = D[0][0] * d[i][0] + D[0][1] * d[i][1] + D[0][2] * d[i][2];
            Jc[1][i] = D[1][0] * d[i][0] + D[1][1] * d[i][1] + D[1][2] * d[i][2];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            Jp[0][j] = D[0][0] * cc[6 + j] + D[0][1] * cc[9 + j] + D[0][2] * cc[12 + j];
            Jp[1][j] = D[1][0] * cc[6 + j] + D[1][1] * cc[9 + j] + D[1][2] * cc[12 + j];
        }
        if constexpr (NP == 6) {  // d/dT = -D R
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                Jc[0][3 + j] = -Jp[0][j];
                Jc[1][3 + j] = -Jp[1][j];
            }
        }
    }
}

// One observation: projection (u, v) and, if JAC, Jc (2 x NP) w.r.t. [angles(3), T(NP-3)] and Jp (2 x 3).
template <int MODEL, int NP, bool JAC>
__device__ inline void project(const double* __restrict__ cc, const double* __restrict__ rpc_tab, double X, double Y,
                               double Z, bool f32, double& u, double& v, double Jc[2][NP], double Jp[2][3]) {
    Rot r;
    if constexpr (MODEL == AFFINE) {
        rotate<JAC>(cc, X, Y, Z, r);
        const double t0 = cc[CAMX], t1 = cc[CAMX + 1], fx = cc[CAMX + 2], fy = cc[CAMX + 3], sk = cc[CAMX + 4];
        const double q0 = r.y3[0] + t0, q1 = r.y3[1] + t1;
        u = fx * q0 + sk * q1;
        v = fy * q1;
        if (JAC) {
            const double* d[3] = {r.da, r.db, r.dg};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                Jc[0][i] = fx * d[i][0] + sk * d[i][1];
                Jc[1][i] = fy * d[i][1];
            }
            if constexpr (NP == 5) {  // d/dt = A
                Jc[0][3] = fx; Jc[0][4] = sk;
                Jc[1][3] = 0.0; Jc[1][4] = fy;
            }
            // rows 0 and 1 of R from the six trig values already in registers (same expressions as cam_constants):
            // six fewer gathers from the LDS camera table per observation; the kernels are bound by the LDS pipe
            const double ca = cc[0], sa = cc[1], cb = cc[2], sb = cc[3], cg = cc[4], sg = cc[5];
            const double R0[3] = {cg * cb, cg * sb * sa - sg * ca, cg * sb * ca + sg * sa};
            const double R1[3] = {sg * cb, sg * sb * sa + cg * ca, sg * sb * ca - cg * sa};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                Jp[0][j] = fx * R0[j] + sk * R1[j];
                Jp[1][j] = fy * R1[j];
            }
        }
    } else if constexpr (MODEL == PERSPECTIVE) {
        rotate<JAC>(cc, X, Y, Z, r);
        const double fx = cc[CAMX + 3], fy = cc[CAMX + 4], sk = cc[CAMX + 5], cx = cc[CAMX + 6], cy = cc[CAMX + 7];
        const double q0 = r.y3[0] + cc[CAMX], q1 = r.y3[1] + cc[CAMX + 1], q2 = r.y3[2] + cc[CAMX + 2];
        const double un = fx * q0 + sk * q1 + cx * q2;
        const double vn = fy * q1 + cy * q2;
        u = un / q2;
        v = vn / q2;
        if (JAC) {
            const double iz = 1.0 / q2;
            const double D[2][3] = {{fx * iz, sk * iz, (cx - u) * iz}, {0.0, fy * iz, (cy - v) * iz}};
            const double* d[3] = {r.da, r.db, r.dg};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                Jc[0][i] = D[0][0] * d[i][0] + D[0][1] * d[i][1] + D[0][2] * d[i][2];
                Jc[1][i] = D[1][1] * d[i][1] + D[1][2] * d[i][2];
            }
            if constexpr (NP == 6) {  // d/dt = D
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    Jc[0][3 + j] = D[0][j];
                    Jc[1][3 + j] = D[1][j];
                }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                Jp[0][j] = D[0][0] * cc[6 + j] + D[0][1] * cc[9 + j] + D[0][2] * cc[12 + j];
                Jp[1][j] = D[1][1] * cc[9 + j] + D[1][2] * cc[12 + j];
            }
        }
    } else {
        double D[2][3];
        project_rpc_d<NP, JAC>(cc, rpc_tab, X, Y, Z, f32, u, v, Jc, Jp, D);
    }
}

// Robust loss of one scalar residual, scipy semantics
// (scipy:optimize/_lsq/least_squares.py:172-227, scipy:optimize/_lsq/common.py:720-731):
//   rho0 = f_scale^2 rho(z), z = (f / f_scale)^2;  js = sqrt(max(rho' + 2 rho'' z, eps));  fs = f rho' / js
__device__ inline void robust(int loss, double f_scale, double f, double& rho0, double& fs, double& js, bool fast_soft = false) {
    // no contraction: the kernels that evaluate the loss at the same point (k_linearize, k_residual, k_cam_sums; their host loops
    // compare the costs) must round the same way whatever they inline this into
#pragma clang fp contract(off)
    if (loss == 0) {
        rho0 = f * f;
        fs = f;
        js = 1.0;
        return;
    }
    const double q = f / f_scale;
    const double z = q * q;
    double r0, r1, r2;
    if (loss == 1) {  // soft_l1
        // rho' = t^-1/2 and rho' + 2 rho'' z = t^-1/2 - z t^-3/2 = t^-3/2 with t = 1 + z, so that js = t^-3/4 and
        // fs = f rho' / js = f t^1/4: two reciprocal square roots (t^-1/2, then (t^1/2)^-1/2) instead of the two square roots and
        // three divisions of the generic form below -- 8 divisions and 4 square roots per observation were half of the
        // arithmetic of the soft_l1 linearize kernel.  The generic form takes over where its clamp would act (t^-3/2 < eps).
        // fast_soft: affine and perspective cameras (every kernel of a model takes the same form).  The RPC chain keeps the generic
        // form: its tight runs against the 3-point reference stop on 1e-15 tests along a flat direction and moved by 1e-6 with it.
        const double t = 1.0 + z;
        if (fast_soft && t < 1.0e10) {
            const double r = rsqrt(t), st = t * r;  // t^-1/2, t^1/2
            const double q4 = rsqrt(st);            // t^-1/4
            rho0 = f_scale * f_scale * (2.0 * (st - 1.0));
            js = r * q4;
            fs = f * (st * q4);
            return;
        }
        const double st = sqrt(t);
        r0 = 2.0 * (st - 1.0); r1 = 1.0 / st; r2 = -0.5 / (t * st);
    } else if (loss == 2) {  // huber
        if (z <= 1.0) { r0 = z; r1 = 1.0; r2 = 0.0; }
        else { const double sz = sqrt(z); r0 = 2.0 * sz - 1.0; r1 = 1.0 / sz; r2 = -0.5 / (z * sz); }
    } else if (loss == 3) {  // cauchy
        const double t = 1.0 + z;
        r0 = log1p(z); r1 = 1.0 / t; r2 = -1.0 / (t * t);
    } else {  // arctan
        const double t = 1.0 + z * z;
        r0 = atan(z); r1 = 1.0 / t; r2 = -2.0 * z / (t * t);
    }
    rho0 = f_scale * f_scale * r0;
    // rho[2] /= f_scale^2 and J_scale = rho1 + 2 rho2 f^2  ==  r1 + 2 r2 z
    double j2 = r1 + 2.0 * r2 * z;
    j2 = j2 < 2.220446049250313e-16 ? 2.220446049250313e-16 : j2;
    js = sqrt(j2);
    fs = f * r1 / js;
}

}  // namespace satba
