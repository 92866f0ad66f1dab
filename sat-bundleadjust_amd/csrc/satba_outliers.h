// satba_outliers.h -- outlier rejection between the soft-L1 and the L2 solve of the pipeline, on the device:
// per-observation reprojection error, per-camera elbow threshold, observations above it
// (ref:bundle_adjust/ba_outliers.py:14-58 `get_elbow_value`, :112-155 `compute_obs_to_remove`).
//
// The elbow of a camera is the sorted error value farthest from the chord between the smallest and the largest one.  The
// removed set must come out index-exact against the reference, so the distance is evaluated with the reference's own
// sequence of IEEE operations (numpy evaluates every step separately: no fused multiply-adds here) and ties go to the
// first index like np.argmax.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

namespace satba {

// err[pos] = | |f / w| |_2 of the residual pair (ref:bundle_adjust/ba_core.py:335-349), ELL order
__global__ void k_out_err_ell(int P, const int* __restrict__ e_cam, const double2* __restrict__ f, const double* __restrict__ w,
                              double* __restrict__ err) {
    // no contraction: HIP's __dmul_rn / __dadd_rn are plain operators, and a * a + b * b fused into fma(a, a, b * b) differs from
    // numpy's two products and one sum in the last bit (round 5: the errors are returned to the caller in place of the host formula)
#pragma clang fp contract(off)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
        if (e_cam[i] < 0) continue;
        const double a = fabs(__ddiv_rn(f[i].x, w[i])), b = fabs(__ddiv_rn(f[i].y, w[i]));
        const double aa = a * a, bb = b * b;
        err[i] = __dsqrt_rn(aa + bb);
    }
}
// camera-major copy of the errors: from ELL order (src_is_ell) or from the caller's observation order through obs_pos^-1
__global__ void k_out_gather_cm(long long K, const int* __restrict__ cm_pos, const double* __restrict__ err_ell, double* __restrict__ cm_err) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < K; i += (long long)gridDim.x * blockDim.x) cm_err[i] = err_ell[cm_pos[i]];
}
__global__ void k_out_scatter_ell(long long K, const int* __restrict__ obs_pos, const double* __restrict__ err_obs, double* __restrict__ err_ell) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < K; o += (long long)gridDim.x * blockDim.x) err_ell[obs_pos[o]] = err_obs[o];
}

// one workgroup per camera over its ascending error values v[0 .. n): threshold of ba_outliers.py:131-138
//   elbow = v[argmax_i dist((i, v_i), chord)], success = elbow >= percentile(v, 100 - 20)
//   thr = success ? max(elbow, min_thr) : v[n - 1];  cam_thr = round(thr, 2)
__global__ __launch_bounds__(256) void k_out_elbow(const int* __restrict__ cam_ofs, const double* __restrict__ sorted, double predef_thr,
                                                   double min_thr, double* __restrict__ cam_thr) {
    const int cam = blockIdx.x;
    if (predef_thr >= 0.0) {
        if (threadIdx.x == 0) cam_thr[cam] = __ddiv_rn(rint(__dmul_rn(predef_thr, 100.0)), 100.0);
        return;
    }
    const int b = cam_ofs[cam], n = cam_ofs[cam + 1] - b;
    const double* v = sorted + b;
    if (n <= 0) {
        if (threadIdx.x == 0) cam_thr[cam] = 0.0;
        return;
    }
    // chord from (0, v0) to (n - 1, v_last), normalised (numpy: line_vec / sqrt(sum(line_vec ** 2)))
    const double lx = (double)(n - 1), ly = __dsub_rn(v[n - 1], v[0]);
    const double nrm = __dsqrt_rn(__dadd_rn(__dmul_rn(lx, lx), __dmul_rn(ly, ly)));
    const double ux = __ddiv_rn(lx, nrm), uy = __ddiv_rn(ly, nrm);
    double best = -1.0;
    int best_i = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double px = (double)i, py = __dsub_rn(v[i], v[0]);
        const double sp = __dadd_rn(__dmul_rn(px, ux), __dmul_rn(py, uy));
        const double qx = __dsub_rn(px, __dmul_rn(sp, ux)), qy = __dsub_rn(py, __dmul_rn(sp, uy));
        const double d = __dsqrt_rn(__dadd_rn(__dmul_rn(qx, qx), __dmul_rn(qy, qy)));
        if (d > best) { best = d; best_i = i; }  // ascending i per thread: the first maximum of the thread's subset
    }
    __shared__ double s_d[256];
    __shared__ int s_i[256];
    s_d[threadIdx.x] = best; s_i[threadIdx.x] = best_i;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const double d2 = s_d[threadIdx.x + s];
            const int i2 = s_i[threadIdx.x + s];
            if (d2 > s_d[threadIdx.x] || (d2 == s_d[threadIdx.x] && i2 < s_i[threadIdx.x])) { s_d[threadIdx.x] = d2; s_i[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int arg = s_i[0];
        if (!(s_d[0] >= 0.0) || arg >= n) arg = 0;  // n == 1: the chord is a point and numpy's distances are NaN -> argmax 0
        const double elbow = v[arg];
        // np.percentile(err, 80), linear interpolation with numpy's two-sided lerp.  Virtual index: method "linear" of numpy 2.2
        // (numpy/lib/_function_base_impl.py, _QuantileMethods["linear"]: get_virtual_index = (n - 1) * quantiles, NOT the generic
        // n q + (alpha + q (1 - alpha - beta)) - 1 of the other methods); checked bit for bit against np.percentile for n < 3000
        // in tests/test_host_logic.py
        const double vi = __dmul_rn((double)(n - 1), 0.8);
        const int lo = (int)floor(vi), hi = min(lo + 1, n - 1);
        const double t = __dsub_rn(vi, (double)lo), a = v[lo], bb = v[hi], diff = __dsub_rn(bb, a);
        double pct = __dadd_rn(a, __dmul_rn(diff, t));
        if (t >= 0.5) pct = __dsub_rn(bb, __dmul_rn(diff, __dsub_rn(1.0, t)));
        const bool success = !(elbow < pct);
        const double thr = success ? fmax(elbow, min_thr) : v[n - 1];
        cam_thr[cam] = __ddiv_rn(rint(__dmul_rn(thr, 100.0)), 100.0);  // np.round(thr, 2)
    }
}

// remove[o] = err > cam_thr[cam] in the caller's observation order; *count += removed
__global__ void k_out_mask(long long K, const int* __restrict__ obs_pos, const int* __restrict__ e_cam, const double* __restrict__ err_ell,
                           const double* __restrict__ cam_thr, unsigned char* __restrict__ remove, unsigned long long* __restrict__ count) {
    unsigned long long c = 0;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < K; o += (long long)gridDim.x * blockDim.x) {
        const int pos = obs_pos[o];
        const unsigned char r = err_ell[pos] > cam_thr[e_cam[pos]] ? 1 : 0;
        remove[o] = r;
        c += r;
    }
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

}  // namespace satba
