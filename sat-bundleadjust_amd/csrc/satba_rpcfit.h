// satba_rpcfit.h -- RPC re-fit after the solve, batched over the cameras on the device:
//   ref:bundle_adjust/ba_rpcfit.py:88-153  weighted_lsq: per camera and image axis a 39-unknown rational fit
//       (20 numerator + 19 denominator coefficients) of ~1000 normalised 3-D -> 2-D correspondences: one unweighted linear solve,
//       then <= 20 re-weighted solves (weights 1 / denominator^2, Tikhonov term h^2 I) until the RMSE moves by less than tol
//   ref:bundle_adjust/ba_rpcfit.py:156-198 scaling_params / initialize_rpc: offsets and scales from the extrema of the samples
//   rpcm.RPCModel.localization (third-party, absent): image point + altitude -> lon / lat by inverting the projection;
//       k_rpc_localize runs the Newton / secant inversion of satba_triangulate.h (same root, its own path to it)
//
// One workgroup per camera.  The normal equations of both axes are accumulated from tiles of 64 samples staged in LDS (every
// thread owns a few entries of the two 39 x 40 augmented matrices), solved in LDS by Gaussian elimination with partial pivoting
// (the reference calls numpy.linalg.inv: LU with partial pivoting), and the RMSE of the updated model decides about another pass.
// The matrices are badly conditioned (cond 1e13 - 1e17; for an affine camera the true denominators are constant and the
// unregularised first solve is rank deficient in exact arithmetic): two correct solvers agree on the fitted projection to
// ~1e-3 px and not on the coefficients, and may need a different number of passes.  The work is tiny (3 MFLOP per pass and
// camera); it is on the device because its inputs (the localisation grid, the corrected projection) and its consumers are.
#pragma once
#include <hip/hip_runtime.h>
#include "satba_triangulate.h"

namespace satba {

constexpr int FIT_N = 39;        // unknowns per axis
constexpr int FIT_LD = 41;       // row stride of the augmented matrices in LDS
constexpr int FIT_TILE = 64;     // samples staged at a time
constexpr int FIT_THREADS = 256;

__device__ inline void fit_monomials(double L, double P, double H, double (&m)[20]) {  // RPC00B order, L = lon, P = lat, H = alt
    m[0] = 1.0; m[1] = L; m[2] = P; m[3] = H; m[4] = L * P; m[5] = L * H; m[6] = P * H; m[7] = L * L; m[8] = P * P; m[9] = H * H;
    m[10] = P * L * H; m[11] = L * L * L; m[12] = L * P * P; m[13] = L * H * H; m[14] = L * L * P; m[15] = P * P * P; m[16] = P * H * H;
    m[17] = L * L * H; m[18] = P * P * H; m[19] = H * H * H;
}
__device__ inline double fit_block_reduce(double v, double* s_red, bool is_max) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v, o); v = is_max ? fmax(v, t) : v + t; }
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    double r = s_red[0];
    for (int w = 1; w < FIT_THREADS / 64; ++w) r = is_max ? fmax(r, s_red[w]) : r + s_red[w];
    return r;
}

// target: n_cam x n x 2 (col, row); locs: n_cam x n x 3 (lon, lat, alt); tables: n_cam x 90 (include/satba.h record)
__global__ __launch_bounds__(FIT_THREADS) void k_rpc_fit(int n, const double* __restrict__ target, const double* __restrict__ locs, double h,
                                                         double tol, int max_iter, double* __restrict__ tables, double* __restrict__ rmse_out,
                                                         int* __restrict__ iters_out) {
    __shared__ double s_A[2][FIT_N][FIT_LD];      // augmented normal equations of the col (0) and row (1) axis
    __shared__ double s_m[FIT_TILE][20];          // monomials of the staged samples
    __shared__ double s_x[FIT_TILE][2], s_w[FIT_TILE][2];  // normalised targets and weights
    __shared__ double s_coef[2][40];              // [axis][num(20) den(20)]
    __shared__ double s_red[FIT_THREADS / 64];
    __shared__ int s_piv;
    const int cam = blockIdx.x, tid = threadIdx.x;
    const double* T = target + (size_t)cam * n * 2;
    const double* X = locs + (size_t)cam * n * 3;
    // offsets and scales: (max - min) / 2 and min + scale (ba_rpcfit.py:156-164), order col row lon lat alt
    double off[5], scl[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        double lo = INFINITY, hi = -INFINITY;
        for (int i = tid; i < n; i += FIT_THREADS) {
            const double v = q < 2 ? T[2 * i + q] : X[3 * i + (q - 2)];
            lo = fmin(lo, v); hi = fmax(hi, v);
        }
        hi = fit_block_reduce(hi, s_red, true);
        lo = -fit_block_reduce(-lo, s_red, true);
        scl[q] = (hi - lo) / 2; off[q] = lo + scl[q];
    }
    auto normalised = [&](int i, double& L, double& P, double& H, double& C, double& R) {
        C = (T[2 * i] - off[0]) / scl[0]; R = (T[2 * i + 1] - off[1]) / scl[1];
        L = (X[3 * i] - off[2]) / scl[2]; P = (X[3 * i + 1] - off[3]) / scl[3]; H = (X[3 * i + 2] - off[4]) / scl[4];
    };
    double rmse = 0.0;
    int iters = 0;
    for (int pass = 0; pass <= max_iter; ++pass) {
        constexpr int NE = FIT_N * (FIT_N + 1), PER = (NE + FIT_THREADS - 1) / FIT_THREADS;
        double acc[2][PER];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < PER; ++e) acc[a][e] = 0.0;
        for (int base = 0; base < n; base += FIT_TILE) {
            __syncthreads();
            if (tid < FIT_TILE) {
                const int i = base + tid;
                double m[20] = {0};
                double C = 0.0, R = 0.0, wc = 0.0, wr = 0.0;
                if (i < n) {
                    double L, P, H;
                    normalised(i, L, P, H, C, R);
                    fit_monomials(L, P, H, m);
                    wc = wr = 1.0;
                    if (pass > 0) {  // 1 / denominator^2 of the current model
                        double dc = 0.0, dr = 0.0;
#pragma unroll
                        for (int k = 0; k < 20; ++k) { dc += s_coef[0][20 + k] * m[k]; dr += s_coef[1][20 + k] * m[k]; }
                        wc = 1.0 / (dc * dc); wr = 1.0 / (dr * dr);
                    }
                }
#pragma unroll
                for (int k = 0; k < 20; ++k) s_m[tid][k] = m[k];
                s_x[tid][0] = C; s_x[tid][1] = R; s_w[tid][0] = wc; s_w[tid][1] = wr;
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                const int idx = tid + e * FIT_THREADS;
                if (idx >= NE) break;
                const int r = idx / (FIT_N + 1), c = idx % (FIT_N + 1);
                for (int t = 0; t < FIT_TILE; ++t) {
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const double x = s_x[t][a];
                        const double mr = r < 20 ? s_m[t][r] : -x * s_m[t][r - 19];
                        const double mc = c == FIT_N ? x : (c < 20 ? s_m[t][c] : -x * s_m[t][c - 19]);
                        acc[a][e] += s_w[t][a] * mr * mc;
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int idx = tid + e * FIT_THREADS;
            if (idx >= NE) break;
            const int r = idx / (FIT_N + 1), c = idx % (FIT_N + 1);
            s_A[0][r][c] = acc[0][e] + ((pass > 0 && r == c) ? h * h : 0.0);
            s_A[1][r][c] = acc[1][e] + ((pass > 0 && r == c) ? h * h : 0.0);
        }
        __syncthreads();
        // Gaussian elimination with partial pivoting, both axes one after the other
        for (int a = 0; a < 2; ++a) {
            for (int k = 0; k < FIT_N; ++k) {
                if (tid < 64) {
                    const int r = k + tid;
                    double v = r < FIT_N ? fabs(s_A[a][r][k]) : -1.0;
                    int best = r;
                    for (int o = 32; o > 0; o >>= 1) {
                        const double v2 = __shfl_xor(v, o);
                        const int b2 = __shfl_xor(best, o);
                        if (v2 > v || (v2 == v && b2 < best)) { v = v2; best = b2; }
                    }
                    if (tid == 0) s_piv = best;
                }
                __syncthreads();
                const int p = s_piv;
                if (p != k && tid <= FIT_N) { const double t = s_A[a][k][tid]; s_A[a][k][tid] = s_A[a][p][tid]; s_A[a][p][tid] = t; }
                __syncthreads();
                const double piv = s_A[a][k][k];
                const int rows = FIT_N - 1 - k, cols = FIT_N - k;  // columns k + 1 .. FIT_N
                double f_keep[8], up[8];
                int cnt = 0;
                for (int idx = tid; idx < rows * cols; idx += FIT_THREADS) {
                    const int r = k + 1 + idx / cols, c = k + 1 + idx % cols;
                    f_keep[cnt] = s_A[a][r][k] / piv; up[cnt] = s_A[a][k][c]; ++cnt;
                }
                __syncthreads();
                cnt = 0;
                for (int idx = tid; idx < rows * cols; idx += FIT_THREADS) {
                    const int r = k + 1 + idx / cols, c = k + 1 + idx % cols;
                    s_A[a][r][c] -= f_keep[cnt] * up[cnt]; ++cnt;
                }
                __syncthreads();
            }
            if (tid == 0) {  // back substitution -> [num(20) | 1, den(19)]
                double sol[FIT_N];
                for (int k = FIT_N - 1; k >= 0; --k) {
                    double v = s_A[a][k][FIT_N];
                    for (int j = k + 1; j < FIT_N; ++j) v -= s_A[a][k][j] * sol[j];
                    sol[k] = v / s_A[a][k][k];
                }
                for (int k = 0; k < 20; ++k) s_coef[a][k] = sol[k];
                s_coef[a][20] = 1.0;
                for (int k = 0; k < 19; ++k) s_coef[a][21 + k] = sol[20 + k];
            }
            __syncthreads();
        }
        // RMSE of the updated model in pixels (ba_rpcfit.py:76-85): sqrt(mean(MSE_col, MSE_row))
        double se_c = 0.0, se_r = 0.0;
        for (int i = tid; i < n; i += FIT_THREADS) {
            double L, P, H, C, R, m[20];
            normalised(i, L, P, H, C, R);
            fit_monomials(L, P, H, m);
            double nc = 0.0, dc = 0.0, nr = 0.0, dr = 0.0;
#pragma unroll
            for (int k = 0; k < 20; ++k) { nc += s_coef[0][k] * m[k]; dc += s_coef[0][20 + k] * m[k]; nr += s_coef[1][k] * m[k]; dr += s_coef[1][20 + k] * m[k]; }
            const double ec = (nc / dc - C) * scl[0], er = (nr / dr - R) * scl[1];
            se_c += ec * ec; se_r += er * er;
        }
        se_c = fit_block_reduce(se_c, s_red, false); se_r = fit_block_reduce(se_r, s_red, false);
        const double prev = rmse;
        rmse = sqrt(0.5 * (se_c / n + se_r / n));
        iters = pass;
        if (pass > 0 && fabs(prev - rmse) < tol) break;
    }
    if (tid < 80) {  // record: col_num col_den row_num row_den | lon lat alt col row (offset, scale)
        tables[(size_t)cam * 90 + tid] = s_coef[tid / 40][tid % 40];
    }
    if (tid == 0) {
        double* t = tables + (size_t)cam * 90 + 80;
        t[0] = off[2]; t[1] = scl[2]; t[2] = off[3]; t[3] = scl[3]; t[4] = off[4]; t[5] = scl[4];
        t[6] = off[0]; t[7] = scl[0]; t[8] = off[1]; t[9] = scl[1];
        if (rmse_out) rmse_out[cam] = rmse;
        if (iters_out) iters_out[cam] = iters;
    }
}

// rpcm.RPCModel.localization: (col, row, alt) -> (lon, lat) through one camera's table
__global__ __launch_bounds__(256) void k_rpc_localize(long long n, const double* __restrict__ table, const double* __restrict__ col,
                                                      const double* __restrict__ row, const double* __restrict__ alt, double* __restrict__ lon,
                                                      double* __restrict__ lat) {
    __shared__ double s_tab[TRI_RPC_STRIDE];
    if (threadIdx.x < 90) s_tab[threadIdx.x] = table[threadIdx.x];
    __syncthreads();
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double lo, la;
    tri_localize(TabLds{s_tab}, col[i], row[i], alt[i], lo, la);
    lon[i] = lo; lat[i] = la;
}

// ------------------------------------------------------------------------------------------------ re-fit, device resident (round 3)
// fit_Rt_corrected_rpc for a batch of cameras without the host in the data path (ref:bundle_adjust/ba_rpcfit.py:270-345; round 2
// built the grids with numpy and ran one localisation launch per camera and margin round: 1 s for 50 cameras, 6 ms of it on the
// device).  Per margin round, for the cameras still to do:
//   k_refit_grid   thread = grid node: (col, row, alt) of the n x n x n mesh over the crop + margin (numpy.linspace's arithmetic,
//                  columns fastest), localisation through the ORIGINAL RPC, geodetic -> ECEF (grid), + global transform, the
//                  corrected projection x = P(R (X - T - C) + C) (satba_models.h: project<RPC, 6>) -> target;
//   k_rpc_fit      the fit (above), one workgroup per camera;
//   k_refit_check  one workgroup per camera: reprojection error of the fitted model per sample (check_errors, :357-370) and the
//                  coverage test (:347-355): the grid re-projected through the FITTED model, its convex hull (gift wrapping from the
//                  lowest point, cross products in the reference's float64), the four image corners against every hull edge with
//                  the tolerance of the host version (1e-9 on unit normals).
// The host reads one flag per camera and round (covered / not), doubles the margin of the others and goes again.
constexpr int REFIT_MAX_N = 16;  // grid nodes per axis (the reference uses 10)

struct RefitCam {          // per camera, device memory
    double table[90];      // original RPC
    double rt[9];          // Euler angles, T, C of the correction
    double crop[4];        // col0, row0, width, height
    double alt[2];         // altitude range of the mesh
    double margin;
    int slot, pad;         // position in the output arrays
};

__device__ inline double refit_linspace(double lo, double hi, int n, int i) {  // numpy.linspace(lo, hi, n)[i]: start + i * step, the end point exact
#pragma clang fp contract(off)  // numpy multiplies, then adds: two roundings
    if (n == 1) return lo;
    if (i == n - 1) return hi;
    const double step = (hi - lo) / (double)(n - 1);
    return lo + (double)i * step;
}

__device__ inline void refit_to_ecef(double lat, double lon, double alt, double& x, double& y, double& z) {  // ref:bundle_adjust/geo_utils.py:218-233
    const double rl = lat * (M_PI / 180.0), ro = lon * (M_PI / 180.0);
    const double f = 1.0 / 298.257223563, e2 = 1.0 - (1.0 - f) * (1.0 - f);
    double sl, cl, so, co;
    sincos(rl, &sl, &cl); sincos(ro, &so, &co);
    const double v = WGS84_A / sqrt(1.0 - e2 * sl * sl);
    x = (v + alt) * cl * co; y = (v + alt) * cl * so; z = (v * (1.0 - e2) + alt) * sl;
}

// grid: n_cam x n^3 x 3 ECEF nodes (without the global transform), locs: lon, lat, alt, target: corrected projection
__global__ __launch_bounds__(256) void k_refit_grid(int n, const RefitCam* __restrict__ cams, const double* __restrict__ gt, double* __restrict__ grid,
                                                    double* __restrict__ locs, double* __restrict__ target) {
    __shared__ double s_tab[TRI_RPC_STRIDE];
    __shared__ double s_cc[CAMC];
    const int cam = blockIdx.y, n3 = n * n * n;
    const RefitCam& c = cams[cam];
    if (threadIdx.x < 90) s_tab[threadIdx.x] = c.table[threadIdx.x];
    if (threadIdx.x == 0) cam_constants(RPC, c.rt, s_cc);
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n3) return;
    const int ic = i % n, ir = (i / n) % n, ia = i / (n * n);
    const double col = refit_linspace(c.crop[0] - c.margin, c.crop[0] + c.crop[2] + c.margin, n, ic);
    const double row = refit_linspace(c.crop[1] - c.margin, c.crop[1] + c.crop[3] + c.margin, n, ir);
    const double alt = refit_linspace(c.alt[0], c.alt[1], n, ia);
    double lon, lat;
    tri_localize(TabLds{s_tab}, col, row, alt, lon, lat);
    double x, y, z;
    refit_to_ecef(lat, lon, alt, x, y, z);
    const size_t o = (size_t)c.slot * n3 + i;
    grid[3 * o] = x; grid[3 * o + 1] = y; grid[3 * o + 2] = z;
    locs[3 * o] = lon; locs[3 * o + 1] = lat; locs[3 * o + 2] = alt;
    double u, v, Jc[2][6], Jp[2][3];
    project<RPC, 6, false>(s_cc, s_tab, x + gt[0], y + gt[1], z + gt[2], false, u, v, Jc, Jp);
    target[2 * o] = u; target[2 * o + 1] = v;
}

// err: n_cam x n^3 reprojection errors of the fitted model; covered[slot] = 1 if the hull of the re-projected grid holds the crop
__global__ __launch_bounds__(256) void k_refit_check(int n, const RefitCam* __restrict__ cams, const double* __restrict__ tables,
                                                     const double* __restrict__ grid, const double* __restrict__ locs, const double* __restrict__ target,
                                                     double* __restrict__ err, int* __restrict__ covered) {
    __shared__ double s_tab[TRI_RPC_STRIDE];
    extern __shared__ double2 s_p[];  // n^3 re-projected nodes (dynamic: 16 KB at the reference's n = 10, 64 KB at REFIT_MAX_N)
    __shared__ int s_idx[256];
    __shared__ int s_bad;
    const int cam = blockIdx.x, tid = threadIdx.x, n3 = n * n * n;
    const RefitCam& c = cams[cam];
    const size_t base = (size_t)c.slot * n3;
    if (tid < 90) s_tab[tid] = tables[(size_t)c.slot * 90 + tid];
    if (tid == 0) s_bad = 0;
    __syncthreads();
    for (int i = tid; i < n3; i += 256) {
        const size_t o = base + i;
        double u, v;
        tri_project(TabLds{s_tab}, locs[3 * o], locs[3 * o + 1], locs[3 * o + 2], u, v);
        err[o] = hypot(u - target[2 * o], v - target[2 * o + 1]);
        // the coverage test projects the ECEF grid: geodetic coordinates by the reference's closed form, not the localised ones
        double geo[3], G[3][3];
        geodetic<false>(grid[3 * o], grid[3 * o + 1], grid[3 * o + 2], geo, G);
        tri_project(TabLds{s_tab}, geo[1], geo[0], geo[2], u, v);
        s_p[i] = make_double2(u, v);
    }
    __syncthreads();
    // gift wrapping: start at the lowest (then leftmost) point; the next vertex is the one all others lie to the left of
    auto arg_best = [&](auto better) {  // index of the best point under `better(a, b)` = a beats b; block-wide
        int bi = -1;
        for (int i = tid; i < n3; i += 256)
            if (bi < 0 || better(i, bi)) bi = i;
        s_idx[tid] = bi;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) {
                const int a = s_idx[tid], b = s_idx[tid + st];
                if (a < 0 || (b >= 0 && better(b, a))) s_idx[tid] = b;
            }
            __syncthreads();
        }
        const int r = s_idx[0];
        __syncthreads();
        return r;
    };
    const int start = arg_best([&](int a, int b) { return s_p[a].y < s_p[b].y || (s_p[a].y == s_p[b].y && s_p[a].x < s_p[b].x); });
    const double cx[4] = {c.crop[0], c.crop[0], c.crop[0] + c.crop[2], c.crop[0] + c.crop[2]};
    const double cy[4] = {c.crop[1], c.crop[1] + c.crop[3], c.crop[1] + c.crop[3], c.crop[1]};
    int cur = start;
    for (int guard = 0; guard <= n3; ++guard) {
        const double2 pc = s_p[cur];
        // next hull vertex (counter-clockwise): q such that no point is to the right of pc -> q; ties: the farthest
        const int nxt = arg_best([&](int a, int b) {
            if (a == cur) return false;
            if (b == cur) return true;
            const double ax = s_p[a].x - pc.x, ay = s_p[a].y - pc.y, bx = s_p[b].x - pc.x, by = s_p[b].y - pc.y;
            const double cr = bx * ay - by * ax;  // > 0: a is to the left of pc -> b, i.e. b is the more clockwise one
            if (cr < 0) return true;              // a is to the right of pc -> b: a wraps tighter
            if (cr > 0) return false;
            return ax * ax + ay * ay > bx * bx + by * by;
        });
        if (nxt < 0) break;
        if (tid == 0) {  // the image corners against the edge pc -> pn: outward normal (dy, -dx) / |d|, tolerance 1e-9 like the host version
            const double dx = s_p[nxt].x - pc.x, dy = s_p[nxt].y - pc.y, len = sqrt(dx * dx + dy * dy);
            if (len > 0.0)
                for (int k = 0; k < 4; ++k)
                    if ((dy * (cx[k] - pc.x) - dx * (cy[k] - pc.y)) / len > 1e-9) s_bad = 1;
        }
        cur = nxt;
        if (cur == start) break;
    }
    __syncthreads();
    if (tid == 0) covered[c.slot] = s_bad ? 0 : 1;
}

}  // namespace satba
