// satba_linearize3.h -- K2 v3: residual + analytic Jacobian -> normal-equation blocks with NO cross-lane
// reductions and NO LDS atomics: two passes, each accumulating in registers.
//
// Measurements that led here (profiles/r1_pmc_linearize_schur.txt, tools/ubench/lds_atomics.hip): the fused
// wave-tile kernel (v1) is bound by LDS *instruction issue* -- 20-27 ds_add_f64 per observation for the camera
// table (~3 lanes/clk/CU) plus 108 ds_bpermute per tile for the per-point segmented sums -- and routing the
// camera products through LDS buckets (v2) needs as many LDS instructions.  v3 removes the LDS from the data
// path:
//
//   k_lin_points   one THREAD per point walks the point's observations (they are contiguous): residual pair f,
//                  V_p (6) and g_p (3) accumulate in registers and are stored once -- no shuffles, no atomics, any
//                  number of observations per point.  Neighbouring lanes read addresses one point apart; every
//                  fetched line is consumed over the next iterations of the same wave (L1/TA hits).
//   k_lin_cameras  a second pass over a CAMERA-MAJOR copy of the observation data (coalesced 28 B/obs stream +
//                  a 24 B gather of the point, served by L2 / Infinity Cache): each thread accumulates the
//                  NP(NP+1)/2 + NP products of its camera in registers over a strided slice of that camera's list,
//                  one LDS tree reduction per workgroup at the end, one partial per (camera, chunk).
//
// Both passes evaluate the projection and Jacobian (2x the arithmetic of v1; the kernels stay memory bound).
// The camera constants are staged in LDS once per workgroup when they fit.
#pragma once
#include "satba_kernels.h"

namespace satba {

struct Lin3Args {
    const int* __restrict__ pt_ofs;   // N + 1
    double2* __restrict__ f;          // K
    double* __restrict__ V;           // N x 6
    double* __restrict__ gp;          // N x 3
    double* __restrict__ hdr_cost;
    double* __restrict__ hdr_gpmax;
};

template <int MODEL, int NP, bool ROBUST, bool CL>
__global__ __launch_bounds__(256) void k_lin_points(ObsArgs a, Lin3Args s) {
    extern __shared__ double s_camc[];
    const double* cbase = cam_table<CL>(a, s_camc, 256);
    double cost = 0.0, gmax = 0.0;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < a.N; p += gridDim.x * 256) {
        const int o0 = s.pt_ofs[p], o1 = s.pt_ofs[p + 1];
        double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int o = o0; o < o1; ++o) {
            const int cam = a.cam[o];
            ObsEval<MODEL, NP, true, ROBUST> e;
            e.eval(a, o, cam, p, cbase + (size_t)cam * CAMC);
            if constexpr (MODEL == RPC) { if (a.Jpm) e.store_jac(a, o); }
            if (a.sc) a.sc[o] = make_double2(e.sw[0], e.sw[1]);
            s.f[o] = make_double2(e.ftrue[0], e.ftrue[1]);
            cost += e.rho;
            v[0] += e.Jp[0][0] * e.Jp[0][0] + e.Jp[1][0] * e.Jp[1][0];
            v[1] += e.Jp[0][0] * e.Jp[0][1] + e.Jp[1][0] * e.Jp[1][1];
            v[2] += e.Jp[0][0] * e.Jp[0][2] + e.Jp[1][0] * e.Jp[1][2];
            v[3] += e.Jp[0][1] * e.Jp[0][1] + e.Jp[1][1] * e.Jp[1][1];
            v[4] += e.Jp[0][1] * e.Jp[0][2] + e.Jp[1][1] * e.Jp[1][2];
            v[5] += e.Jp[0][2] * e.Jp[0][2] + e.Jp[1][2] * e.Jp[1][2];
            v[6] += e.Jp[0][0] * e.fs[0] + e.Jp[1][0] * e.fs[1];
            v[7] += e.Jp[0][1] * e.fs[0] + e.Jp[1][1] * e.fs[1];
            v[8] += e.Jp[0][2] * e.fs[0] + e.Jp[1][2] * e.fs[1];
        }
        double* Vp = s.V + 6 * (size_t)p;
        double* gq = s.gp + 3 * (size_t)p;
#pragma unroll
        for (int k = 0; k < 6; ++k) Vp[k] = v[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) gq[k] = v[6 + k];
        gmax = fmax(gmax, fmax(fabs(v[6]), fmax(fabs(v[7]), fabs(v[8]))));
    }
    cost = wave_sum(cost);
    gmax = wave_max(gmax);
    __shared__ double s_red[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_red[0][wave] = cost; s_red[1][wave] = gmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(s.hdr_cost, 0.5 * (s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3]));
        atomic_max_pos(s.hdr_gpmax, fmax(fmax(s_red[1][0], s_red[1][1]), fmax(s_red[1][2], s_red[1][3])));
    }
}

// camera-major copy of the observation data
struct CamMajor {
    const int* __restrict__ cam_ofs;     // M + 1
    const double2* __restrict__ obs;     // K, camera-major
    const double* __restrict__ w;        // K
    const int* __restrict__ pt;          // K
    const int* __restrict__ oidx;        // K: observation index of each camera-major position
};

constexpr int LINC_THREADS = 256;

// grid: (chunks_per_cam, M); part: [M][chunks][CU]
template <int MODEL, int NP, bool ROBUST>
__global__ __launch_bounds__(LINC_THREADS) void k_lin_cameras(ObsArgs a, CamMajor c, double* __restrict__ part) {
    constexpr int CU = cam_acc_len(NP);
    const int cam = blockIdx.y, chunk = blockIdx.x, n_chunks = gridDim.x;
    const int b = c.cam_ofs[cam], e = c.cam_ofs[cam + 1];
    const long long len = e - b;
    const int lo = b + (int)(len * chunk / n_chunks), hi = b + (int)(len * (chunk + 1) / n_chunks);
    const double* cc = a.camc + (size_t)cam * CAMC;  // one camera per workgroup: wave-uniform (scalar) loads
    const double* tab = (MODEL == RPC) ? a.rpc + (size_t)cam * 90 : nullptr;
    const double mc = (cam >= a.n_cam_fix) ? 1.0 : 0.0;
    double acc[CU];
#pragma unroll
    for (int k = 0; k < CU; ++k) acc[k] = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += LINC_THREADS) {
        const double2 ob = c.obs[i];
        const double w = c.w[i];
        const int pt = c.pt[i];
        const double* px = a.x + a.n_c + 3 * (size_t)pt;
        double u, v, Jc[2][NP], Jp[2][3];
        project<MODEL, NP, true>(cc, tab, px[0], px[1], px[2], a.f32 != 0, u, v, Jc, Jp);
        const double f0 = w * (u - ob.x), f1 = w * (v - ob.y);
        double fs0 = f0, fs1 = f1, js0 = 1.0, js1 = 1.0, r0, r1;
        if constexpr (ROBUST) {
            robust(a.loss, a.f_scale, f0, r0, fs0, js0);
            robust(a.loss, a.f_scale, f1, r1, fs1, js1);
        }
        const double s0 = w * js0 * mc, s1 = w * js1 * mc;
#pragma unroll
        for (int k = 0; k < NP; ++k) { Jc[0][k] *= s0; Jc[1][k] *= s1; }
        int k = 0;
#pragma unroll
        for (int r = 0; r < NP; ++r)
#pragma unroll
            for (int q = r; q < NP; ++q) acc[k++] += Jc[0][r] * Jc[0][q] + Jc[1][r] * Jc[1][q];
#pragma unroll
        for (int r = 0; r < NP; ++r) acc[k++] += Jc[0][r] * fs0 + Jc[1][r] * fs1;
    }
    __shared__ double s_red[LINC_THREADS / 64][CU];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < CU; ++k) {
        const double t = wave_sum(acc[k]);
        if (lane == 0) s_red[wave][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < CU) {
        double t = 0.0;
        for (int wv = 0; wv < LINC_THREADS / 64; ++wv) t += s_red[wv][threadIdx.x];
        part[((size_t)cam * n_chunks + chunk) * CU + threadIdx.x] = t;
    }
}

// sum the (camera, chunk) partials and expand to the exchange payload U (M x NP x NP, full) | g_c (M x NP)
__global__ void k_lin3_finish(int M, int NP, int n_chunks, const double* __restrict__ part, double* __restrict__ U,
                              double* __restrict__ gc) {
    const int CU = cam_acc_len(NP);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * CU) return;
    const int cam = idx / CU, k = idx % CU;
    double s = 0.0;
    for (int ch = 0; ch < n_chunks; ++ch) s += part[((size_t)cam * n_chunks + ch) * CU + k];
    const int ntri = NP * (NP + 1) / 2;
    if (k >= ntri) {
        gc[cam * NP + (k - ntri)] = s;
        return;
    }
    int i = 0, rem = k;
    while (rem >= NP - i) { rem -= NP - i; ++i; }
    const int j = i + rem;
    U[(size_t)cam * NP * NP + i * NP + j] = s;
    U[(size_t)cam * NP * NP + j * NP + i] = s;
}

}  // namespace satba
