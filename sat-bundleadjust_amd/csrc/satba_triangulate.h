// satba_triangulate.h -- initial 3-D points from the feature tracks, on the device:
//   ref:bundle_adjust/feature_tracks/ft_triangulate.py:57-127  init_pts3d: for every triangulation pair (in list order) the tracks
//       seen by both cameras are triangulated and folded into a float32 running mean per track
//   ref:...ft_triangulate.py:18-34  linear triangulation (cv2.triangulatePoints: null vector of the 4 x 4 DLT matrix)
//   ref:...ft_triangulate.py:37-54 -> ref:bundle_adjust/s2p/triangulation.py:82-135 -> ref:c/disp_to_h.c:40-64 -> ref:c/rpc.c:480-514
//       RPC triangulation: height iteration along the epipolar curve, each step two localisations (Newton inversion of the
//       projection, ref:c/rpc.c:372-408) and two projections
//
// The reference walks the pairs and touches every track they share; the mean is float32 and therefore depends on the order, so a
// track replays exactly that order: it enumerates the ordered camera pairs of its own observations, looks their list indices up
// in an M x M table, and lists them in ascending index (a bounded sorted buffer in LDS, refilled until the track's pairs are
// exhausted).  The triangulations of all tracks then form one flat work list (one thread each: the tracks are few and of very
// different lengths, the triangulations many), and a last pass folds each track's slice into the mean with the reference's
// sequence of float32 operations (no fused multiply-add).  The triangulations themselves are float64 and use fused
// multiply-adds: they agree with the reference's to ~1e-8 m, which the float32 store (ulp 0.125 - 0.5 m at ECEF magnitudes)
// almost always hides.
//
// RPC: VALU-bound (about 30 k float64 instructions per triangulation, two thirds of them in the cubic polynomials); the tables of
// all cameras sit in LDS with an odd stride (91 doubles) when they fit, because the lanes of a wave work on different cameras.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

namespace satba {

constexpr int TRI_THREADS = 128;   // threads per workgroup of k_tri_list
constexpr int TRI_BUF = 24;        // pair indices a track holds sorted at a time
constexpr int TRI_RPC_STRIDE = 91; // doubles per camera of the LDS copy of the RPC tables
constexpr int TRI_LOC_MAXIT = 100; // the reference's localisation loops without a bound (ref:c/rpc.c:394)

// ------------------------------------------------------------------------------------------------ linear (DLT) triangulation
// Right singular vector of the smallest singular value of the 4 x 4 matrix a (rows = equations) by one-sided Jacobi rotations of
// its columns: accurate for the badly scaled columns of this problem (the homogeneous column is ~1e6 times the others).
__device__ inline void tri_null_vector4(double (&a)[4][4], double (&x)[4]) {
    double v[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) v[r][c] = r == c ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) { al += a[r][p] * a[r][p]; be += a[r][q] * a[r][q]; ga += a[r][p] * a[r][q]; }
                if (fabs(ga) > 1e-16 * sqrt(al * be) && ga != 0.0) {
                    rotated = true;
                    // t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)), zeta = (be - al) / (2 ga), with one division and one square root
                    const double d = be - al, g2 = 2.0 * ga;
                    const double t = copysign(1.0, d) * g2 / (fabs(d) + sqrt(d * d + g2 * g2));
                    const double c = rsqrt(1.0 + t * t), s = c * t;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double ap = a[r][p], aq = a[r][q], vp = v[r][p], vq = v[r][q];
                        a[r][p] = c * ap - s * aq; a[r][q] = s * ap + c * aq;
                        v[r][p] = c * vp - s * vq; v[r][q] = s * vp + c * vq;
                    }
                }
            }
        if (!rotated) break;
    }
    double best = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double nn = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) nn += a[r][c] * a[r][c];
        if (c == 0 || nn < best) {
            best = nn;
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = v[r][c];
        }
    }
}
// P1, P2: 3 x 4 row-major; (x1, y1), (x2, y2): the observations (col, row)
__device__ inline void tri_linear(const double* __restrict__ P1, const double* __restrict__ P2, double x1, double y1, double x2, double y2,
                                  double (&X)[3]) {
    double a[4][4], h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[0][k] = x1 * P1[8 + k] - P1[k];
        a[1][k] = y1 * P1[8 + k] - P1[4 + k];
        a[2][k] = x2 * P2[8 + k] - P2[k];
        a[3][k] = y2 * P2[8 + k] - P2[4 + k];
    }
    tri_null_vector4(a, h);
    X[0] = h[0] / h[3]; X[1] = h[1] / h[3]; X[2] = h[2] / h[3];
}

// ------------------------------------------------------------------------------------------------ RPC triangulation
// T: callable k -> entry k of the camera's SATBA_RPC_TABLE_LEN record
template <class T>
__device__ inline void tri_nrpci(const T& c, double L, double P, double H, double& xo, double& yo) {  // ref:c/rpc.c:337-348
    const double m[20] = {1.0, L, P, H, L * P, L * H, P * H, L * L, P * P, H * H, P * L * H, L * L * L, L * P * P, L * H * H, L * L * P,
                          P * P * P, P * H * H, L * L * H, P * P * H, H * H * H};
    double cn = 0.0, cd = 0.0, rn = 0.0, rd = 0.0;
#pragma unroll
    for (int i = 0; i < 20; ++i) {
        cn += c(i) * m[i]; cd += c(20 + i) * m[i]; rn += c(40 + i) * m[i]; rd += c(60 + i) * m[i];
    }
    xo = cn / cd; yo = rn / rd;
}
// The localisation evaluates the four cubics ~20 times at one height: they are first reduced to bivariate cubics in (L, P)
// (10 coefficients each, held in registers), so that the Newton loop reads no table and does 40 instead of 80 multiply-adds per
// evaluation.  Same polynomial, different association of the sum: the results differ from the reference's in the last digits.
struct TriCubic2 {
    double b[4][10];  // [col_num, col_den, row_num, row_den][1, L, P, LP, L2, P2, L3, LP2, L2P, P3]
    template <class T>
    __device__ inline void reduce(const T& c, double H) {
        const double H2 = H * H, H3 = H2 * H;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int o = 20 * k;
            b[k][0] = c(o) + c(o + 3) * H + c(o + 9) * H2 + c(o + 19) * H3;
            b[k][1] = c(o + 1) + c(o + 5) * H + c(o + 13) * H2;
            b[k][2] = c(o + 2) + c(o + 6) * H + c(o + 16) * H2;
            b[k][3] = c(o + 4) + c(o + 10) * H;
            b[k][4] = c(o + 7) + c(o + 17) * H;
            b[k][5] = c(o + 8) + c(o + 18) * H;
            b[k][6] = c(o + 11); b[k][7] = c(o + 12); b[k][8] = c(o + 14); b[k][9] = c(o + 15);
        }
    }
    __device__ inline void eval(double L, double P, double& xo, double& yo) const {
        const double LP = L * P;
        double r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            r[k] = b[k][0] + L * (b[k][1] + L * (b[k][4] + L * b[k][6])) + P * (b[k][2] + P * (b[k][5] + P * b[k][9])) +
                   LP * (b[k][3] + b[k][7] * P + b[k][8] * L);
        xo = r[0] / r[1]; yo = r[2] / r[3];
    }
};
// localisation: image (x, y) at height z -> (lon, lat), Newton inversion of the projection (ref:c/rpc.c:372-408, 428-438), delta = 0.1
template <class T>
__device__ inline void tri_localize(const T& c, double x, double y, double z, double& lon_o, double& lat_o) {
    const double xf = (x - c(86)) / c(87), yf = (y - c(88)) / c(89);
    TriCubic2 f;
    f.reduce(c, (z - c(84)) / c(85));
    const double delta = 0.1;
    double lon = -delta, lat = -delta, eps = 2.0 * delta;
    for (int it = 0; it <= TRI_LOC_MAXIT; ++it) {
        double x00, x01, x10, x11, x20, x21;
        f.eval(lon, lat, x00, x01);
        const double u0 = xf - x00, u1 = yf - x01;
        if (!(u0 * u0 + u1 * u1 > 1e-18)) break;
        f.eval(lon + eps, lat, x10, x11); f.eval(lon, lat + eps, x20, x21);
        const double e10 = x10 - x00, e11 = x11 - x01, e20 = x20 - x00, e21 = x21 - x01;
        const double det = e10 * e21 - e11 * e20;
        lon += (e21 * u0 - e20 * u1) / det * eps;
        lat += (-e11 * u0 + e10 * u1) / det * eps;
        eps = 0.1;
    }
    lon_o = lon * c(81) + c(80); lat_o = lat * c(83) + c(82);
}
template <class T>
__device__ inline void tri_project(const T& c, double lon, double lat, double z, double& x, double& y) {  // ref:c/rpc.c:441-451
    double a, b;
    tri_nrpci(c, (lon - c(80)) / c(81), (lat - c(82)) / c(83), (z - c(84)) / c(85), a, b);
    x = a * c(87) + c(86); y = b * c(89) + c(88);
}
// ref:c/rpc.c:480-514 (rpc_height) + ref:c/disp_to_h.c:52-61 + ref:bundle_adjust/geo_utils.py:218-233: ECEF point and the distance
// to the epipolar curve of the last iteration.  The keypoints go through float32 (ref:bundle_adjust/s2p/triangulation.py:115).
// One loop whose passes alternate between the heights h and h + 1 (one instance of the localisation in the code); the pass at h
// that follows the last update is the final localisation of disp_to_h.c:55.
template <class TA, class TB>
__device__ inline float tri_rpc(const TA& ca, const TB& cb, double xa_, double ya_, double xb_, double yb_, double (&X)[3]) {
    const double xa = (double)(float)xa_, ya = (double)(float)ya_, xb = (double)(float)xb_, yb = (double)(float)yb_;
    double h = 0.0, err = 0.0, p0 = 0.0, p1 = 0.0, lon, lat;
    int t = 0, s = 0;
    bool done = false;
#pragma nounroll
    while (true) {
        const double hs = h + (double)s;
        tri_localize(ca, xa, ya, hs, lon, lat);
        if (s == 0 && done) break;
        double q0, q1;
        tri_project(cb, lon, lat, hs, q0, q1);
        if (s == 0) {
            p0 = q0; p1 = q1;
        } else {
            const double a0 = q0 - p0, a1 = q1 - p1, b0 = xb - p0, b1 = yb - p1;
            const double lam = (a0 * b0 + a1 * b1) / (a0 * a0 + a1 * a1);
            err = hypot(p0 + lam * a0 - xb, p1 + lam * a1 - yb);
            h += lam;
            done = fabs(lam) < 0.00001 || ++t >= 100;
        }
        s ^= 1;
    }
    const double rl = lat * (M_PI / 180.0), ro = lon * (M_PI / 180.0);
    const double f = 1.0 / 298.257223563, e2 = 1.0 - (1.0 - f) * (1.0 - f);
    double sl, cl, so, co;
    sincos(rl, &sl, &cl); sincos(ro, &so, &co);
    const double v = 6378137.0 / sqrt(1.0 - e2 * sl * sl);
    X[0] = (v + h) * cl * co; X[1] = (v + h) * cl * so; X[2] = (v * (1.0 - e2) + h) * sl;
    return (float)err;
}

// ((count - 1) * avg + new) / count with every operation rounded to float32 on its own: the __f*_rn intrinsics of this toolchain
// are plain operators, which the compiler would contract into a fused multiply-add
__device__ inline float tri_mean_update(float avg, float cm1, float x, float cnt) {
#pragma clang fp contract(off)
    const float prod = cm1 * avg;
    const float sum = prod + x;
    return __fdiv_rn(sum, cnt);
}

struct TabGlobal {
    const double* t;
    __device__ inline double operator()(int k) const { return t[k]; }
};
struct TabLds {
    const double* t;  // points into __shared__ memory; the loads below are generic -- the inlined kernel sees the address space
    __device__ inline double operator()(int k) const { return t[k]; }
};

// one pair of cameras, n correspondences: the reference's linear_triangulation_multiple_pts / rpc_triangulation
template <int MODEL>
__global__ __launch_bounds__(256) void k_tri_pairwise(long long n, const double* __restrict__ cam_i, const double* __restrict__ cam_j,
                                                      const double* __restrict__ pts_i, const double* __restrict__ pts_j,
                                                      double* __restrict__ out, float* __restrict__ err) {
    __shared__ double s_cam[2 * TRI_RPC_STRIDE];
    if constexpr (MODEL == 2) {
        for (int i = threadIdx.x; i < 180; i += 256) s_cam[(i / 90) * TRI_RPC_STRIDE + i % 90] = i < 90 ? cam_i[i] : cam_j[i - 90];
        __syncthreads();
    }
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double X[3];
    float e = 0.f;
    if constexpr (MODEL == 2) {
        e = tri_rpc(TabLds{s_cam}, TabLds{s_cam + TRI_RPC_STRIDE}, pts_i[2 * i], pts_i[2 * i + 1], pts_j[2 * i], pts_j[2 * i + 1], X);
    } else {
        tri_linear(cam_i, cam_j, pts_i[2 * i], pts_i[2 * i + 1], pts_j[2 * i], pts_j[2 * i + 1], X);
    }
    out[3 * i] = X[0]; out[3 * i + 1] = X[1]; out[3 * i + 2] = X[2];
    if (err) err[i] = e;
}

struct TriArgs {
    int M, N, n_pairs, cam_len;
    int ordered;            // the pair list ascends lexicographically with c_i < c_j and the cameras of every track ascend: the nested
                            // enumeration of a track's cameras already meets its pairs in list order
    const int* pt_ofs;      // N + 1
    const int* cam_ind;     // K, the cameras of a track (any order)
    const double* obs;      // K x 2
    const double* cams;     // M x cam_len: 12 (3 x 4 projection matrices) or SATBA_RPC_TABLE_LEN
    const int* pair_first;  // M x M: first list index of the ordered pair (c_i, c_j), -1 if none
    const int* pair_next;   // n_pairs: next list index with the same ordered pair, -1 at the end
    const int* pairs;       // n_pairs x 2
    int* n_tri;             // N + 1: triangulations per track (k_tri_count; the last entry stays 0)
    const int* tri_ofs;     // N + 1: exclusive prefix sum of n_tri
    int2* entries;          // per triangulation, track-major and in list order within a track: the two observations
    float* res;             // per triangulation: the point, rounded to float32
    float* out;             // N x 3
    // a problem handle's resident layout instead of track-major lists (satba_init_pts3d_resident): observation k of track q sits at
    // slice_base[q / 64] + 64 k + q % 64 of cam_ind / obs (sliced ELL, satba_layout.h), pt_ofs only gives the track lengths, tracks
    // are the handle's internal points and results go to the caller's point order through perm
    const int* slice_base = nullptr;
    const unsigned char* removed = nullptr;  // per observation position: 1 = treat as absent (outlier mask), or null
    const int* perm = nullptr;               // output row of track q (null: q)
    int* n_tri_out = nullptr;                // triangulations per track in the caller's order (null: not wanted)
};
__device__ inline int tri_pos(const TriArgs& a, int q, int o0, int k) { return a.slice_base ? a.slice_base[q >> 6] + 64 * k + (q & 63) : o0 + k; }
__device__ inline int tri_cam(const TriArgs& a, int pos) { return (a.removed && a.removed[pos]) ? -1 : a.cam_ind[pos]; }

// Four passes.  (1) k_tri_count: triangulations per track = listed ordered pairs among its cameras; a prefix sum gives every track
// its slice of the work list.  (2) k_tri_list: the track writes its triangulations into the slice in the order the reference's pair
// loop reaches them.  (3) k_tri_points: one thread per triangulation -- the arithmetic, spread over the whole chip whatever the
// track lengths are.  (4) k_tri_mean: the float32 running mean of a track over its slice.
__global__ __launch_bounds__(256) void k_tri_count(const TriArgs a) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= a.N) return;
    const int o0 = a.pt_ofs[q], k = a.pt_ofs[q + 1] - o0;
    int total = 0;
    for (int ia = 0; ia < k; ++ia) {
        const int ca = tri_cam(a, tri_pos(a, q, o0, ia));
        if (ca < 0) continue;
        for (int ib = a.ordered ? ia + 1 : 0; ib < k; ++ib) {
            if (ib == ia) continue;
            const int cb = tri_cam(a, tri_pos(a, q, o0, ib));
            if (cb < 0) continue;
            for (int id = a.pair_first[(size_t)ca * a.M + cb]; id >= 0; id = a.pair_next[id]) ++total;
        }
    }
    a.n_tri[q] = total;
}
// ordered lists (what a pipeline's pair selection produces): no sorting
__global__ __launch_bounds__(256) void k_tri_list_ordered(const TriArgs a) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= a.N) return;
    const int o0 = a.pt_ofs[q], k = a.pt_ofs[q + 1] - o0;
    int w = a.tri_ofs[q];
    if (a.tri_ofs[q + 1] == w) return;
    for (int ia = 0; ia < k; ++ia) {
        const int pa = tri_pos(a, q, o0, ia), ca = tri_cam(a, pa);
        if (ca < 0) continue;
        for (int ib = ia + 1; ib < k; ++ib) {
            const int pb = tri_pos(a, q, o0, ib), cb = tri_cam(a, pb);
            if (cb >= 0 && a.pair_first[(size_t)ca * a.M + cb] >= 0) a.entries[w++] = make_int2(pa, pb);
        }
    }
}

__global__ __launch_bounds__(TRI_THREADS) void k_tri_list(const TriArgs a) {
    __shared__ int s_buf[TRI_BUF][TRI_THREADS];
    const int q = blockIdx.x * TRI_THREADS + threadIdx.x;
    if (q >= a.N) return;
    const int tid = threadIdx.x;
    const int o0 = a.pt_ofs[q], k = a.pt_ofs[q + 1] - o0;
    int last = -1, w = a.tri_ofs[q];
    if (a.tri_ofs[q + 1] == w) return;
    while (true) {
        // the TRI_BUF smallest list indices above `last` among the ordered camera pairs of the track, ascending
        int nb = 0;
        for (int ia = 0; ia < k; ++ia) {
            const int ca = tri_cam(a, tri_pos(a, q, o0, ia));
            if (ca < 0) continue;
            for (int ib = 0; ib < k; ++ib) {
                if (ib == ia) continue;
                const int cb = tri_cam(a, tri_pos(a, q, o0, ib));
                if (cb < 0) continue;
                for (int id = a.pair_first[(size_t)ca * a.M + cb]; id >= 0; id = a.pair_next[id]) {
                    if (id <= last) continue;
                    if (nb == TRI_BUF && id >= s_buf[TRI_BUF - 1][tid]) continue;
                    int pos = nb < TRI_BUF ? nb++ : TRI_BUF - 1;
                    while (pos > 0 && s_buf[pos - 1][tid] > id) { s_buf[pos][tid] = s_buf[pos - 1][tid]; --pos; }
                    s_buf[pos][tid] = id;
                }
            }
        }
        for (int e = 0; e < nb; ++e) {
            const int id = s_buf[e][tid];
            const int ci = a.pairs[2 * id], cj = a.pairs[2 * id + 1];
            int ia = 0, ib = 0;
            for (int t = 0; t < k; ++t) {
                const int c = tri_cam(a, tri_pos(a, q, o0, t));
                if (c == ci) ia = t;
                if (c == cj) ib = t;
            }
            a.entries[w++] = make_int2(tri_pos(a, q, o0, ia), tri_pos(a, q, o0, ib));
        }
        if (nb < TRI_BUF) break;
        last = s_buf[nb - 1][tid];
    }
}

// MODEL 2 = rpc, else linear; TAB_LDS: RPC tables staged in (dynamic) LDS
template <int MODEL, bool TAB_LDS>
__global__ __launch_bounds__(256) void k_tri_points(const TriArgs a, int n_entries) {
    extern __shared__ double s_tab[];
    if constexpr (MODEL == 2 && TAB_LDS) {
        for (int i = threadIdx.x; i < a.M * 90; i += 256) s_tab[(i / 90) * TRI_RPC_STRIDE + i % 90] = a.cams[i];
        __syncthreads();
    }
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n_entries; e += gridDim.x * 256) {
        const int2 ob = a.entries[e];
        const int ci = a.cam_ind[ob.x], cj = a.cam_ind[ob.y];
        const double2 pi = reinterpret_cast<const double2*>(a.obs)[ob.x], pj = reinterpret_cast<const double2*>(a.obs)[ob.y];
        double X[3];
        if constexpr (MODEL == 2) {
            if constexpr (TAB_LDS) tri_rpc(TabLds{s_tab + ci * TRI_RPC_STRIDE}, TabLds{s_tab + cj * TRI_RPC_STRIDE}, pi.x, pi.y, pj.x, pj.y, X);
            else tri_rpc(TabGlobal{a.cams + (size_t)ci * 90}, TabGlobal{a.cams + (size_t)cj * 90}, pi.x, pi.y, pj.x, pj.y, X);
        } else {
            tri_linear(a.cams + (size_t)ci * 12, a.cams + (size_t)cj * 12, pi.x, pi.y, pj.x, pj.y, X);
        }
        a.res[3 * (size_t)e] = (float)X[0]; a.res[3 * (size_t)e + 1] = (float)X[1]; a.res[3 * (size_t)e + 2] = (float)X[2];
    }
}

__global__ __launch_bounds__(256) void k_tri_mean(const TriArgs a) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= a.N) return;
    float avg[3] = {0.f, 0.f, 0.f}, cnt = 0.f;
    for (int e = a.tri_ofs[q]; e < a.tri_ofs[q + 1]; ++e) {
        // count += 1; avg = ((count - 1) * avg + new) / count, every operation rounded to float32 (ft_triangulate.py:77-81)
        cnt += 1.f;
        const float cm1 = cnt - 1.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) avg[d] = tri_mean_update(avg[d], cm1, a.res[3 * (size_t)e + d], cnt);
    }
    const size_t row = a.perm ? (size_t)a.perm[q] : (size_t)q;
    a.out[3 * row] = avg[0]; a.out[3 * row + 1] = avg[1]; a.out[3 * row + 2] = avg[2];
    if (a.n_tri_out) a.n_tri_out[row] = a.tri_ofs[q + 1] - a.tri_ofs[q];
}
// outlier mask in the caller's observation order -> the handle's observation positions
__global__ void k_tri_mask_ell(long long K, const int* __restrict__ obs_pos, const unsigned char* __restrict__ rm, unsigned char* __restrict__ rm_ell) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o < K) rm_ell[obs_pos[o]] = rm[o];
}

}  // namespace satba
