// satba_kernels.h -- HIP kernels of the bundle-adjustment hot path (gfx950, wave64).
//
// Work decomposition.  Observations are point-major (ref:bundle_adjust/ba_params.py:142-147): all
// observations of a point are contiguous.  The host cuts the observation stream into WAVE TILES of whole
// points with at most 64 observations; one wavefront processes one tile, lane = observation.  Every
// global read of the observation arrays is then a coalesced 64-lane access, per-point sums (V_p, g_p,
// W^T dc) are segmented wave reductions with no atomics, and per-camera sums (U_c, g_c) are accumulated
// with LDS atomics in a per-workgroup table that is flushed once per workgroup.
// A point with more than 64 observations is split over several tiles flagged `split`: those use
// global atomics for the per-point sums (rare slow path).
//
// Nothing here is GEMM shaped (2x3, 2x6, 3x3, 6x6 blocks): MFMA is not used; the kernels are bound by HBM
// traffic (residual / linearize / back-substitution / Jv products) or fp64 VALU + atomics (Schur).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "satba_models.h"

namespace satba {


__host__ __device__ constexpr int cam_acc_len(int np) { return np * (np + 1) / 2 + np; }
// row stride of the LDS camera table: odd, so that (stride * cam + k) visits all 32 bank pairs (an even stride such
// as 20 folds the cameras onto 8 of them: 8-way conflicts of the ds_add_f64)
__host__ __device__ constexpr int cam_acc_stride(int np) { return cam_acc_len(np) | 1; }

struct ObsArgs {
    const double2* __restrict__ obs;   // K observed (col, row)
    const double* __restrict__ w;      // K weights
    const int* __restrict__ cam;       // K camera index
    const int* __restrict__ pt;        // K local point index
    const int* __restrict__ tile_start;       // n_tiles + 1
    const unsigned char* __restrict__ tile_split;  // n_tiles
    const double* __restrict__ x;      // variable vector whose POINT part is used
    const double* __restrict__ camc;   // M x CAMC camera constants built from the same vector
    const double* __restrict__ rpc;    // M x 90 or null
    double* __restrict__ Jpm;          // RPC only (else null): K x (2 NP + 6) Jacobian blocks Jc | Jp of the current
                                       // linearisation, observation order; written by the linearize kernels and read by
                                       // every later pass (the RPC chain costs 2-3 kflop per evaluation)
    double2* __restrict__ sc;          // weighted / robust runs (else null): K Jacobian row scales (w js0, w js1) of the current
                                       // linearisation, observation order; written by the linearize kernels, read by the
                                       // Schur pair kernel
    long long K;
    int n_tiles, M, N, n_c, n_cam_fix, n_pts_fix, loss, f32;
    int unit;                          // every weight is 1 and the loss is linear
    double f_scale;
};

// weighted, robust-scaled residual and (optionally) Jacobian blocks of observation o
// ROBUST: the loss is not linear; SOFT: it is soft_l1 (folds the runtime loss switch away: registers, no log / atan code)
// UNITW (with !ROBUST): every weight is 1 and the loss is linear -- the weight is not applied and the Jacobian blocks
// are returned WITHOUT the fixed-camera / fixed-point masks (the caller masks its sums instead: ~20 multiplications
// per observation fewer)
template <int MODEL, int NP, bool JAC, bool ROBUST = true, bool SOFT = false, bool UNITW = false>
struct ObsEval {
    double ftrue[2];  // w * (proj - obs)
    double fs[2];     // robust-scaled residual
    double rho;       // contribution to 2 * cost
    double Jc[2][NP];
    double Jp[2][3];
    double sw[2];     // Jacobian row scales w * js (before the fixed-camera / fixed-point masks)

    __device__ inline void eval(const ObsArgs& a, long long o, int cam, int pt) {
        eval(a, o, cam, pt, a.camc + (size_t)cam * CAMC);
    }

    // cc: the camera's constant record (global memory, or a copy staged in LDS)
    __device__ inline void eval(const ObsArgs& a, long long o, int cam, int pt, const double* cc) {
        const double2 ob = a.obs[o];
        const double w = a.w[o];
        const double* px = a.x + a.n_c + 3 * (size_t)pt;
        eval_loaded(a, cam, pt, cc, ob, w, px[0], px[1], px[2]);
    }

    // the same with the observation, its weight and the point already in registers (software-pipelined callers)
    __device__ inline void eval_loaded(const ObsArgs& a, int cam, int pt, const double* cc, const double2 ob, const double w,
                                       const double X, const double Y, const double Z) {
        const double* tab = (MODEL == RPC) ? a.rpc + (size_t)cam * 90 : nullptr;
        double u, v;
        project<MODEL, NP, JAC>(cc, tab, X, Y, Z, a.f32 != 0, u, v, Jc, Jp);
        if constexpr (UNITW && !ROBUST) {
            ftrue[0] = u - ob.x; ftrue[1] = v - ob.y;
            fs[0] = ftrue[0]; fs[1] = ftrue[1];
            rho = ftrue[0] * ftrue[0] + ftrue[1] * ftrue[1];
            sw[0] = 1.0; sw[1] = 1.0;
            return;
        }
        ftrue[0] = w * (u - ob.x);
        ftrue[1] = w * (v - ob.y);
        double r0, r1, js0 = 1.0, js1 = 1.0;
        if constexpr (ROBUST) {
            const int loss = SOFT ? 1 : a.loss;
            robust(loss, a.f_scale, ftrue[0], r0, fs[0], js0);
            robust(loss, a.f_scale, ftrue[1], r1, fs[1], js1);
        } else {  // linear loss, specialised at compile time (no transcendental code, far fewer registers)
            fs[0] = ftrue[0]; fs[1] = ftrue[1];
            r0 = ftrue[0] * ftrue[0]; r1 = ftrue[1] * ftrue[1];
        }
        rho = r0 + r1;
        if (JAC) {
            const double mc = (cam >= a.n_cam_fix) ? 1.0 : 0.0;
            const double mp = (pt >= a.n_pts_fix) ? 1.0 : 0.0;
            const double s0 = w * js0, s1 = w * js1;
            sw[0] = s0; sw[1] = s1;
#pragma unroll
            for (int i = 0; i < NP; ++i) { Jc[0][i] *= s0 * mc; Jc[1][i] *= s1 * mc; }
#pragma unroll
            for (int j = 0; j < 3; ++j) { Jp[0][j] *= s0 * mp; Jp[1][j] *= s1 * mp; }
        }
    }

    // the blocks of the current linearisation, stored / reloaded (RPC): 2 NP + 6 doubles = NP + 3 16-byte words
    __device__ inline void store_jac(const ObsArgs& a, long long o) const {
        double t[2 * NP + 6];
#pragma unroll
        for (int k = 0; k < NP; ++k) { t[k] = Jc[0][k]; t[NP + k] = Jc[1][k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { t[2 * NP + k] = Jp[0][k]; t[2 * NP + 3 + k] = Jp[1][k]; }
        double2* q = reinterpret_cast<double2*>(a.Jpm + (size_t)o * (2 * NP + 6));
#pragma unroll
        for (int k = 0; k < NP + 3; ++k) q[k] = make_double2(t[2 * k], t[2 * k + 1]);
    }
    __device__ inline void load_jac(const ObsArgs& a, long long o) {
        const double2* q = reinterpret_cast<const double2*>(a.Jpm + (size_t)o * (2 * NP + 6));
        double t[2 * NP + 6];
#pragma unroll
        for (int k = 0; k < NP + 3; ++k) { const double2 v = q[k]; t[2 * k] = v.x; t[2 * k + 1] = v.y; }
#pragma unroll
        for (int k = 0; k < NP; ++k) { Jc[0][k] = t[k]; Jc[1][k] = t[NP + k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { Jp[0][k] = t[2 * NP + k]; Jp[1][k] = t[2 * NP + 3 + k]; }
    }
    // Jacobian blocks only, for the passes that follow a linearisation at the same x: from the store when there is one
    // (RPC); otherwise the unit-weight, linear-loss Jacobian times the row scales the linearize kernel stored (a.sc;
    // null when every weight is 1 and the loss is linear).  Neither the observation nor its weight is read and the
    // loss function is not evaluated: 8 bytes per observation are streamed instead of 32.
    __device__ inline void jac(const ObsArgs& a, long long o, int cam, int pt, const double* cc) {
        if constexpr (MODEL == RPC) {
            if (a.Jpm) { load_jac(a, o); return; }
        }
        const double* px = a.x + a.n_c + 3 * (size_t)pt;
        const double* tab = (MODEL == RPC) ? a.rpc + (size_t)cam * 90 : nullptr;
        double u, v;
        project<MODEL, NP, true>(cc, tab, px[0], px[1], px[2], a.f32 != 0, u, v, Jc, Jp);
        double s0 = 1.0, s1 = 1.0;
        if (a.sc) { const double2 t = a.sc[o]; s0 = t.x; s1 = t.y; }
        const double mc = (cam >= a.n_cam_fix) ? 1.0 : 0.0;
        const double mp = (pt >= a.n_pts_fix) ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < NP; ++i) { Jc[0][i] *= s0 * mc; Jc[1][i] *= s1 * mc; }
#pragma unroll
        for (int j = 0; j < 3; ++j) { Jp[0][j] *= s0 * mp; Jp[1][j] *= s1 * mp; }
    }
};

// Camera-constant table access.  CL = true: the workgroup stages the whole table in dynamic LDS once and every
// lookup is a ds_read (the pointer never merges with a global one, so no FLAT instructions are generated);
// CL = false (table too large for LDS): gathers from global memory through L1.
template <bool CL>
__device__ inline const double* cam_table(const ObsArgs& a, double* s_camc, int nthreads) {
    if constexpr (CL) {
        for (int i = threadIdx.x; i < a.M * CAMC; i += nthreads) s_camc[i] = a.camc[i];
        __syncthreads();
        return s_camc;
    } else {
        return a.camc;
    }
}

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    return v;
}
__device__ inline double wave_max(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fmax(v, __shfl_down(v, d));
    return v;
}

// sum v[] over the lanes that share `pt` (contiguous runs); the total lands in the first lane of each run
template <int NV>
__device__ inline void seg_reduce(double (&v)[NV], int pt, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int opt = __shfl_down(pt, d);
        const bool ok = (lane + d < 64) && (opt == pt);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const double o = __shfl_down(v[k], d);
            if (ok) v[k] += o;
        }
    }
}

// Sum NV per-thread values over the workgroup and add each total to *dst[k] with ONE atomic per workgroup
// (same-address atomics serialise at ~12 ns each on gfx950: one per wave was costing 150-300 us per kernel).
template <int NV>
__device__ inline void block_sum_atomic(double (&v)[NV], double* const (&dst)[NV]) {
    __shared__ double s_part[NV][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double t = wave_sum(v[k]);
        if (lane == 0) s_part[k][wave] = t;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double t = 0.0;
        for (int w = 0; w < nw; ++w) t += s_part[threadIdx.x][w];
        atomicAdd(dst[threadIdx.x], t);
    }
}

__device__ inline void atomic_max_pos(double* addr, double v) {  // v >= 0
    atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

// ------------------------------------------------------------------------------------------------ camera constants
__global__ void k_cam_consts(int model, int M, int n_p, int c_p, const double* __restrict__ x,
                             const double* __restrict__ cam_static, double* __restrict__ camc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= M) return;
    double full[11];
    for (int i = 0; i < c_p; ++i) full[i] = cam_static[(size_t)c * c_p + i];
    for (int i = 0; i < n_p; ++i) full[i] = x[(size_t)c * n_p + i];
    cam_constants(model, full, camc + (size_t)c * CAMC);
}

// ------------------------------------------------------------------------------------------------ K1 residuals
// ba_core.fun (ref:bundle_adjust/ba_core.py:157-183): one thread per observation, grid-stride.
// hdr_cost += 0.5 * sum rho.  f may be null (cost only).
// UNITW: every weight is 1 and the loss is linear (the weight array is not read: 8 of 32 streamed bytes per observation)
template <int MODEL, int NP, bool CL, bool UNITW = false>
__global__ __launch_bounds__(512) void k_residual(ObsArgs a, double2* __restrict__ f, double* __restrict__ hdr_cost) {
    extern __shared__ double s_camc_res[];
    const double* cbase = cam_table<CL>(a, s_camc_res, 512);
    double acc = 0.0;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < a.K; o += (long long)gridDim.x * blockDim.x) {
        const int cam = a.cam[o];
        if constexpr (UNITW) {
            ObsEval<MODEL, NP, false, false, false, true> e;
            const double* px = a.x + a.n_c + 3 * (size_t)a.pt[o];
            e.eval_loaded(a, cam, 0, cbase + (size_t)cam * CAMC, a.obs[o], 1.0, px[0], px[1], px[2]);
            if (f) f[o] = make_double2(e.ftrue[0], e.ftrue[1]);
            acc += e.rho;
        } else {
            ObsEval<MODEL, NP, false> e;
            e.eval(a, o, cam, a.pt[o], cbase + (size_t)cam * CAMC);
            if (f) f[o] = make_double2(e.ftrue[0], e.ftrue[1]);
            acc += e.rho;
        }
    }
    double v[1] = {0.5 * acc};
    double* const dst[1] = {hdr_cost};
    block_sum_atomic<1>(v, dst);
}

// ------------------------------------------------------------------------------------------------ K2 linearize
// residual + analytic Jacobian -> normal-equation blocks (replaces scipy's finite differences,
// scipy:optimize/_numdiff.py:628-705, and compute_grad, scipy:optimize/_lsq/common.py:590-595):
//   f[o]            true residual pair                        (16 B / obs written)
//   V[pt] (6), g_p  per-point blocks: the 9 products of a tile are staged in the wave's LDS rows and lane (run, value)
//                   sums its run -- 9 ds_write + ~10 ds_read per tile instead of 108 ds_bpermute (72 B / point written)
//   part[block][M][cam_acc_len]   per-workgroup camera partials (upper triangle of U_c, then g_c), accumulated with
//                   ds_add_f64 in an LDS table (measured ~3 lanes/clk/CU, tools/ubench/lds_atomics.hip)
//   hdr[0] += cost;  hdr[slot] = max |g_p|
// Camera constants come from an LDS copy of the table (CL); linear loss is specialised at compile time (ROBUST).
// row stride of the per-wave staging area: 65, so that the nine value rows of one observation sit in nine different
// bank pairs when the run sums read them (a stride of 64 puts them all in the same one)
constexpr int LIN_STAGE = 65;

// Affine cameras with R+T corrected, unit weights, linear loss: d(col,row)/dT = [[fx, skew], [0, fy]] for every
// observation, so the two translation entries of diag(U_c) are n_obs(c) * fx^2 and n_obs(c) * (skew^2 + fy^2).  The
// kernel skips those two LDS atomics (8 instead of 10 per observation) and k_lin_finish fills the entries in.
__host__ __device__ constexpr bool lin_const_t(int model, int np, bool robust, bool unit) {
    return model == AFFINE && np == 5 && !robust && unit;
}

template <bool ROBUST>
struct LinCfg {
    static constexpr int THREADS = ROBUST ? 512 : 1024;  // the generic robust variants need > 128 VGPRs
    static constexpr int WAVES = THREADS / 64;
};

// LDS layout (dynamic): camera accumulators [M][CU] | camera constants [M][CAMC] (if CL) | per-wave staging [WAVES][9][64]
// FULLU = false: only diag(U_c) and g_c are accumulated (2 NP atomics per observation instead of NP(NP+3)/2): that
// is all the solver needs before the Schur phase, whose camera-major pass (k_schur_diag) forms the full J_c^T J_c
// blocks in registers anyway.
// SOFT (with ROBUST): soft_l1 specialised at compile time; it fits the 1024-thread configuration of the linear loss
// UNITW (linear loss, every weight 1): the weight array is not read, the Jacobians carry no masks -- the fixed-camera mask
// is applied when the workgroup's camera table is flushed, the fixed-point mask when a point's sums are stored
template <int MODEL, int NP, bool ROBUST, bool CL, bool FULLU, bool SOFT = false, bool UNITW = false>
__global__ __launch_bounds__(LinCfg<ROBUST && !SOFT>::THREADS) void k_linearize(ObsArgs a, double2* __restrict__ f, double* __restrict__ V,
                                                                      double* __restrict__ gp, double* __restrict__ part,
                                                                      double* __restrict__ hdr_cost, double* __restrict__ hdr_gpmax) {
    constexpr int CU = cam_acc_len(NP), CUS = cam_acc_stride(NP);
    constexpr int THREADS = LinCfg<ROBUST && !SOFT>::THREADS, WAVES = LinCfg<ROBUST && !SOFT>::WAVES;
    extern __shared__ double s_lin[];
    double* s_acc = s_lin;                                          // M * CUS
    double* s_camc = s_acc + (size_t)a.M * CUS;                     // M * CAMC
    double* s_stage = s_camc + (CL ? (size_t)a.M * CAMC : 0);       // WAVES * 9 * LIN_STAGE
    __shared__ unsigned char s_seg[WAVES][66];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < a.M * CUS; i += THREADS) s_acc[i] = 0.0;
    if constexpr (CL)
        for (int i = threadIdx.x; i < a.M * CAMC; i += THREADS) s_camc[i] = a.camc[i];
    __syncthreads();
    const double* cbase;
    if constexpr (CL) cbase = s_camc; else cbase = a.camc;
    double* stage = s_stage + (size_t)wave * 9 * LIN_STAGE;

    double cost = 0.0, gmax = 0.0;
    // Software pipeline (PIPE): a tile costs three dependent global loads (tile range -> observation record -> point)
    // before the first flop and ~300 clocks of LDS traffic after it; with 4 waves per SIMD those latencies were
    // exposed.  The record of the NEXT tile is requested before the arithmetic of the current one, its point
    // gather before the run sums, the range of the tile after next before that.  RPC keeps the plain loop
    // (register budget).
    constexpr bool PIPE = MODEL != RPC;
    const bool const_t = !FULLU && lin_const_t(MODEL, NP, ROBUST, a.unit != 0);
    const int stride = gridDim.x * WAVES;
    int tile = blockIdx.x * WAVES + wave;
    // ranges: wave-uniform -> scalar loads
    auto range = [&](int t, int& r0, int& r1, int& rs) {
        const int tu = __builtin_amdgcn_readfirstlane(t);
        if (tu < a.n_tiles) { r0 = a.tile_start[tu]; r1 = a.tile_start[tu + 1]; rs = a.tile_split[tu]; } else { r0 = 0; r1 = 0; rs = 0; }
    };
    int o0, o1, osplit, n0 = 0, n1 = 0, nsplit = 0;
    range(tile, o0, o1, osplit);
    int cam = 0, pt = -1 - lane, ncam = 0, npt = -1 - lane;
    double2 ob = make_double2(0.0, 0.0), nob = make_double2(0.0, 0.0);
    double w = 0.0, nw = 0.0, X = 0.0, Y = 0.0, Z = 0.0, nX = 0.0, nY = 0.0, nZ = 0.0;
    if constexpr (PIPE) {
        if (o0 + lane < o1) {
            const long long o = (long long)o0 + lane;
            pt = a.pt[o]; cam = a.cam[o]; ob = a.obs[o];
            if constexpr (!UNITW) w = a.w[o];
            const double* px = a.x + a.n_c + 3 * (size_t)pt;
            X = px[0]; Y = px[1]; Z = px[2];
        }
        range(tile + stride, n0, n1, nsplit);
    }
    for (; tile < a.n_tiles; tile += stride) {
        const long long o = (long long)o0 + lane;
        const bool active = o < o1;
        int nn0 = 0, nn1 = 0, nnsplit = 0;
        if constexpr (PIPE) {
            // request the next tile's record and the range after it
            npt = -1 - lane; ncam = 0;
            if (n0 + lane < n1) {
                const long long on = (long long)n0 + lane;
                npt = a.pt[on]; ncam = a.cam[on]; nob = a.obs[on];
                if constexpr (!UNITW) nw = a.w[on];
            }
            range(tile + 2 * stride, nn0, nn1, nnsplit);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            pt = -1 - lane; cam = 0;
        }
        if (active) {
            ObsEval<MODEL, NP, true, ROBUST, SOFT, UNITW> e;
            if constexpr (PIPE) {
                e.eval_loaded(a, cam, pt, cbase + (size_t)cam * CAMC, ob, w, X, Y, Z);
            } else {
                cam = a.cam[o];
                pt = a.pt[o];
                e.eval(a, o, cam, pt, cbase + (size_t)cam * CAMC);
            }
            if constexpr (MODEL == RPC) { if (a.Jpm) e.store_jac(a, o); }
            if (a.sc) a.sc[o] = make_double2(e.sw[0], e.sw[1]);
            f[o] = make_double2(e.ftrue[0], e.ftrue[1]);
            cost += e.rho;
            // per-point products into the wave's staging rows (conflict-free 8-byte stores)
            stage[0 * LIN_STAGE + lane] = e.Jp[0][0] * e.Jp[0][0] + e.Jp[1][0] * e.Jp[1][0];
            stage[1 * LIN_STAGE + lane] = e.Jp[0][0] * e.Jp[0][1] + e.Jp[1][0] * e.Jp[1][1];
            stage[2 * LIN_STAGE + lane] = e.Jp[0][0] * e.Jp[0][2] + e.Jp[1][0] * e.Jp[1][2];
            stage[3 * LIN_STAGE + lane] = e.Jp[0][1] * e.Jp[0][1] + e.Jp[1][1] * e.Jp[1][1];
            stage[4 * LIN_STAGE + lane] = e.Jp[0][1] * e.Jp[0][2] + e.Jp[1][1] * e.Jp[1][2];
            stage[5 * LIN_STAGE + lane] = e.Jp[0][2] * e.Jp[0][2] + e.Jp[1][2] * e.Jp[1][2];
            stage[6 * LIN_STAGE + lane] = e.Jp[0][0] * e.fs[0] + e.Jp[1][0] * e.fs[1];
            stage[7 * LIN_STAGE + lane] = e.Jp[0][1] * e.fs[0] + e.Jp[1][1] * e.fs[1];
            stage[8 * LIN_STAGE + lane] = e.Jp[0][2] * e.fs[0] + e.Jp[1][2] * e.fs[1];
            // camera block: LDS atomics (ds_add_f64) into this workgroup's table
            double* acc = s_acc + (size_t)cam * CUS;
            int k = 0;
#ifndef SATBA_ABLATE_CAM_ATOMICS
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int j = i; j < NP; ++j) {
                    if (FULLU || (i == j && !(const_t && i >= 3)))
                        atomicAdd(acc + k, e.Jc[0][i] * e.Jc[0][j] + e.Jc[1][i] * e.Jc[1][j]);
                    ++k;
                }
#pragma unroll
            for (int i = 0; i < NP; ++i) atomicAdd(acc + (k++), e.Jc[0][i] * e.fs[0] + e.Jc[1][i] * e.fs[1]);
#else  // ablation build only (tools): keep the products alive with ONE atomic
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int j = i; j < NP; ++j) t += e.Jc[0][i] * e.Jc[0][j] + e.Jc[1][i] * e.Jc[1][j];
#pragma unroll
            for (int i = 0; i < NP; ++i) t += e.Jc[0][i] * e.fs[0] + e.Jc[1][i] * e.fs[1];
            atomicAdd(acc + k, t);
#endif
        }
        if constexpr (PIPE) {
            // the next tile's points: in flight during the run sums
            __builtin_amdgcn_sched_barrier(0);
            if (npt >= 0) {
                const double* px = a.x + a.n_c + 3 * (size_t)npt;
                nX = px[0]; nY = px[1]; nZ = px[2];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#ifndef SATBA_ABLATE_POINT_SUMS
        // runs of equal point index -> lane (q, v) sums value v over run q (7 runs per pass); no shuffles
        const int prev = __shfl_up(pt, 1);
        const bool head = active && (lane == 0 || prev != pt);
        const unsigned long long heads = __ballot(head);
        const int n_runs = __popcll(heads);
        if (head) s_seg[wave][__popcll(heads & ((1ull << lane) - 1ull))] = (unsigned char)lane;
        if (lane == 0) s_seg[wave][n_runs] = (unsigned char)(o1 - o0);
        for (int q0 = 0; q0 < n_runs; q0 += 7) {
            const int q = q0 + lane / 9, v = lane % 9;
            const bool owner = lane < 63 && q < n_runs;
            const int b = owner ? s_seg[wave][q] : 0, en = owner ? s_seg[wave][q + 1] : 0;
            const int ptq = __shfl(pt, b);  // the run's point index sits in the register of the run's first lane
            if (owner) {
                const double* col = stage + v * LIN_STAGE;
                // four independent partial sums: the reads of a run are in flight together instead of one LDS
                // latency per element
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                for (int l = b; l < en; l += 4) {
                    const double a0 = col[l];
                    const double a1 = (l + 1 < en) ? col[l + 1] : 0.0;
                    const double a2 = (l + 2 < en) ? col[l + 2] : 0.0;
                    const double a3 = (l + 3 < en) ? col[l + 3] : 0.0;
                    s0 += a0; s1 += a1; s2 += a2; s3 += a3;
                }
                double sum = (s0 + s1) + (s2 + s3);
                if constexpr (UNITW) sum = (ptq >= a.n_pts_fix) ? sum : 0.0;  // fixed points: their blocks are masked here
                double* dst = (v < 6) ? V + 6 * (size_t)ptq + v : gp + 3 * (size_t)ptq + (v - 6);
                if (osplit) atomicAdd(dst, sum);
                else *dst = sum;
                if (v >= 6) gmax = fmax(gmax, fabs(sum));
            }
        }
#endif
        if constexpr (PIPE) {
            o0 = n0; o1 = n1; osplit = nsplit; n0 = nn0; n1 = nn1; nsplit = nnsplit;
            cam = ncam; pt = npt; ob = nob; w = nw; X = nX; Y = nY; Z = nZ;
        } else {
            range(tile + stride, o0, o1, osplit);
        }
    }
    // per-workgroup epilogue
    __shared__ double s_red[2][WAVES];
    cost = wave_sum(cost);
    gmax = wave_max(gmax);
    if (lane == 0) { s_red[0][wave] = cost; s_red[1][wave] = gmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = 0.0, g = 0.0;
        for (int i = 0; i < WAVES; ++i) { c += s_red[0][i]; g = fmax(g, s_red[1][i]); }
        atomicAdd(hdr_cost, 0.5 * c);
        atomic_max_pos(hdr_gpmax, g);
    }
    double* out = part + (size_t)blockIdx.x * a.M * CU;
    for (int i = threadIdx.x; i < a.M * CU; i += THREADS) {
        const double t = s_acc[(i / CU) * CUS + i % CU];
        out[i] = (UNITW && i / CU < a.n_cam_fix) ? 0.0 : t;  // fixed cameras: masked here on the unit-weight path
    }
}

// sum the per-workgroup camera partials and expand to the exchange payload: U (M x NP x NP, full), g_c (M x NP).
// 64 outputs per workgroup, 16 waves each summing a strided subset of the workgroups, combined through LDS.
__global__ __launch_bounds__(1024) void k_lin_finish(int M, int NP, int nblocks, const double* __restrict__ part,
                                                     const double* __restrict__ overflow, double* __restrict__ U,
                                                     double* __restrict__ gc, const int* __restrict__ cam_ofs,
                                                     const double* __restrict__ camc, int n_cam_fix) {
    const int CU = cam_acc_len(NP);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + lane;
    __shared__ double s_sum[16][64];
    double s = 0.0;
    if (idx < M * CU)
        for (int b = wave; b < nblocks; b += 16) s += part[(size_t)b * M * CU + idx];
    s_sum[wave][lane] = s;
    __syncthreads();
    if (wave != 0 || idx >= M * CU) return;
    s = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += s_sum[w][lane];
    if (overflow) s += overflow[idx];
    const int cam = idx / CU, k = idx % CU;
    const int ntri = NP * (NP + 1) / 2;
    if (k >= ntri) {
        gc[cam * NP + (k - ntri)] = s;
        return;
    }
    int i = 0, rem = k;
    while (rem >= NP - i) { rem -= NP - i; ++i; }
    const int j = i + rem;
    if (cam_ofs && i == j && i >= 3) {  // lin_const_t: translation entries of diag(U_c) in closed form
        const double* cc = camc + (size_t)cam * CAMC;
        const double fx = cc[17], fy = cc[18], sk = cc[19];
        const double cnt = cam >= n_cam_fix ? (double)(cam_ofs[cam + 1] - cam_ofs[cam]) : 0.0;
        s = cnt * (i == 3 ? fx * fx : sk * sk + fy * fy);
    }
    U[(size_t)cam * NP * NP + i * NP + j] = s;
    U[(size_t)cam * NP * NP + j * NP + i] = s;
}

// per-point sums of a split point have been accumulated with atomics: its |g_p| still has to reach the header
__global__ void k_gpmax(int N, const double* __restrict__ gp, double* __restrict__ hdr_gpmax) {
    double m = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 3 * N; i += gridDim.x * blockDim.x) m = fmax(m, fabs(gp[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomic_max_pos(hdr_gpmax, m);
}

// ------------------------------------------------------------------------------------------------ prepare
// x_scale="jac" (scipy:optimize/_lsq/common.py:598-610): scale_inv = column norms of J = sqrt(diag(J^T J)),
// zeros -> 1 on the first evaluation, running maximum afterwards; g_h = g / scale_inv.
// hdr[1] += |g_h|^2, hdr[3] += |x * scale_inv|^2 (camera part only if lead), hdr[4] = lead * |g_c|_inf
__global__ __launch_bounds__(256) void k_prepare_vec(int n, int n_c, int NP, int first, double lead,
                                                     const double* __restrict__ U, const double* __restrict__ gc_red,
                                                     const double* __restrict__ V, const double* __restrict__ x,
                                                     double* __restrict__ g, double* __restrict__ scale_inv,
                                                     double* __restrict__ gh, double* __restrict__ ghs, double* __restrict__ hdr) {
    // ghs = g_h / scale_inv: the unscaled direction of g_h, input of the Jacobian-vector product that follows
    double s_gh = 0.0, s_xs = 0.0, m_gc = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double diag, gi, wgt;
        if (i < n_c) {
            const int cam = i / NP, k = i % NP;
            diag = U[(size_t)cam * NP * NP + k * NP + k];
            gi = gc_red[i];
            g[i] = gi;
            wgt = lead;
            m_gc = fmax(m_gc, fabs(gi));
        } else {
            const int j = i - n_c, p = j / 3, k = j % 3;
            diag = V[6 * (size_t)p + (k == 0 ? 0 : (k == 1 ? 3 : 5))];
            gi = g[i];
            wgt = 1.0;
        }
        double si = sqrt(diag);
        if (first) si = (si == 0.0) ? 1.0 : si;
        else si = fmax(si, scale_inv[i]);
        scale_inv[i] = si;
        const double h = gi / si;
        gh[i] = h;
        ghs[i] = h / si;
        s_gh += wgt * h * h;
        const double xs = x[i] * si;
        s_xs += wgt * xs * xs;
    }
    double v[2] = {s_gh, s_xs};
    double* const dst[2] = {hdr + 1, hdr + 3};
    block_sum_atomic<2>(v, dst);
    m_gc = wave_max(m_gc);
    if ((threadIdx.x & 63) == 0 && m_gc > 0.0) atomic_max_pos(hdr + 4, lead * m_gc);
}

// ------------------------------------------------------------------------------------------------ Jacobian-vector products
constexpr int JVP_ROW = 15;  // per-camera constants of the affine form of k_jvp: B (6) | b (2) | A (6), odd stride
// For NV vectors given in scaled variables (v = q / scale_inv): sums of (J v_a) . (J v_b) over the observations.
// NV = 1: out[0] += |J v1|^2.   NV = 2: out[0] += |J v1|^2, out[1] += (J v1).(J v2), out[2] += |J v2|^2.
template <int MODEL, int NP, int NV, bool CL, bool PRE = false>
__global__ __launch_bounds__(512) void k_jvp(ObsArgs a, const double* __restrict__ q1, const double* __restrict__ q2,
                                             const double* __restrict__ scale_inv, double* __restrict__ out) {
    extern __shared__ double s_camc_jvp[];
    const double* cbase = cam_table<CL>(a, s_camc_jvp, 512);
    // PRE (NV = 1): q1 is already divided by scale_inv (k_prepare_vec); its camera part is staged in LDS behind the
    // camera constants -- no divisions and no global gathers per observation
    double* s_v = s_camc_jvp + (CL ? (size_t)a.M * CAMC : 0);
    if constexpr (PRE) {
        for (int i = threadIdx.x; i < a.n_c; i += 512) s_v[i] = q1[i];
        __syncthreads();
    }
    double s11 = 0.0, s12 = 0.0, s22 = 0.0;
    if constexpr (MODEL == AFFINE && PRE && NV == 1 && CL) {
        // affine cameras: J_c v_c = B_c X + b_c with B_c = sum_i v_ci D_ci, b_c = K-columns . v_cT, and J_p = A_c -- the
        // per-camera constants of k_backsub (JVP_ROW doubles per camera, odd stride), built here from the staged vector;
        // an observation costs 14 LDS reads and 18 multiply-adds instead of the Jacobian evaluation and 16 reads
        __syncthreads();  // every thread is done with the staging loops above
        double* tab = s_camc_jvp;  // overwrites the camera constants: M x JVP_ROW <= M x CAMC
        double row_[JVP_ROW];
        for (int c0 = 0; c0 < a.M; c0 += 512) {
            const int c = c0 + threadIdx.x;
            if (c < a.M) {
                const double* cc = a.camc + (size_t)c * CAMC;
                double u, v, Jc[2][NP], Jp[2][3], b[2] = {0.0, 0.0};
                const double mc = (c >= a.n_cam_fix) ? 1.0 : 0.0;
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    project<AFFINE, NP, true>(cc, nullptr, m == 0 ? 1.0 : 0.0, m == 1 ? 1.0 : 0.0, m == 2 ? 1.0 : 0.0, false, u, v, Jc, Jp);
                    double b0 = 0.0, b1 = 0.0;
#pragma unroll
                    for (int i = 0; i < 3; ++i) { b0 += Jc[0][i] * s_v[c * NP + i]; b1 += Jc[1][i] * s_v[c * NP + i]; }
                    row_[m] = mc * b0; row_[3 + m] = mc * b1;
                }
#pragma unroll
                for (int i = 3; i < NP; ++i) { b[0] += Jc[0][i] * s_v[c * NP + i]; b[1] += Jc[1][i] * s_v[c * NP + i]; }
                row_[6] = mc * b[0]; row_[7] = mc * b[1];
#pragma unroll
                for (int m = 0; m < 3; ++m) { row_[8 + m] = Jp[0][m]; row_[11 + m] = Jp[1][m]; }
            }
            __syncthreads();  // (first trip) all reads of the camera constants through a.camc are global: nothing to wait for
            if (c < a.M) {
#pragma unroll
                for (int k = 0; k < 14; ++k) tab[(size_t)c * JVP_ROW + k] = row_[k];
            }
        }
        __syncthreads();
        // four observations per thread and trip: the loop is a chain of two dependent loads (index -> point), so the
        // number of independent chains in flight is what sets its speed
        constexpr int UN = 4;
        const long long stride = (long long)gridDim.x * blockDim.x;
        for (long long o0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; o0 < a.K; o0 += UN * stride) {
            int cam[UN], pt[UN];
            double2 sc[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const long long o = o0 + u * stride;
                const bool ok = o < a.K;
                cam[u] = ok ? a.cam[o] : 0;
                pt[u] = ok ? a.pt[o] : -1;
                sc[u] = (ok && a.sc) ? a.sc[o] : make_double2(ok ? 1.0 : 0.0, ok ? 1.0 : 0.0);
            }
            double X[UN], Y[UN], Z[UN], v0[UN], v1[UN], v2[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const size_t q = (size_t)a.n_c + 3 * (size_t)(pt[u] < 0 ? 0 : pt[u]);
                X[u] = a.x[q]; Y[u] = a.x[q + 1]; Z[u] = a.x[q + 2];
                const double mp = (pt[u] >= a.n_pts_fix) ? 1.0 : 0.0;
                v0[u] = mp * q1[q]; v1[u] = mp * q1[q + 1]; v2[u] = mp * q1[q + 2];
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const double* row = tab + (size_t)cam[u] * JVP_ROW;
                const double j0 = sc[u].x * (row[0] * X[u] + row[1] * Y[u] + row[2] * Z[u] + row[6] + row[8] * v0[u] + row[9] * v1[u] + row[10] * v2[u]);
                const double j1 = sc[u].y * (row[3] * X[u] + row[4] * Y[u] + row[5] * Z[u] + row[7] + row[11] * v0[u] + row[12] * v1[u] + row[13] * v2[u]);
                s11 += j0 * j0 + j1 * j1;
            }
        }
        double v[1] = {s11};
        double* const dst[1] = {out};
        block_sum_atomic<1>(v, dst);
        return;
    }
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < a.K; o += (long long)gridDim.x * blockDim.x) {
        const int cam = a.cam[o], pt = a.pt[o];
        ObsEval<MODEL, NP, true> e;
        e.jac(a, o, cam, pt, cbase + (size_t)cam * CAMC);
        const size_t ic = (size_t)cam * NP, ip = (size_t)a.n_c + 3 * (size_t)pt;
        double j1[2] = {0, 0}, j2[2] = {0, 0};
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            double v1;
            if constexpr (PRE) v1 = s_v[cam * NP + i];
            else {
                const double si = 1.0 / scale_inv[ic + i];
                v1 = q1[ic + i] * si;
                if (NV == 2) { const double v2 = q2[ic + i] * si; j2[0] += e.Jc[0][i] * v2; j2[1] += e.Jc[1][i] * v2; }
            }
            j1[0] += e.Jc[0][i] * v1; j1[1] += e.Jc[1][i] * v1;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double v1;
            if constexpr (PRE) v1 = q1[ip + j];
            else {
                const double si = 1.0 / scale_inv[ip + j];
                v1 = q1[ip + j] * si;
                if (NV == 2) { const double v2 = q2[ip + j] * si; j2[0] += e.Jp[0][j] * v2; j2[1] += e.Jp[1][j] * v2; }
            }
            j1[0] += e.Jp[0][j] * v1; j1[1] += e.Jp[1][j] * v1;
        }
        s11 += j1[0] * j1[0] + j1[1] * j1[1];
        if (NV == 2) {
            s12 += j1[0] * j2[0] + j1[1] * j2[1];
            s22 += j2[0] * j2[0] + j2[1] * j2[1];
        }
    }
    if (NV == 2) {
        double v[3] = {s11, s12, s22};
        double* const dst[3] = {out, out + 1, out + 2};
        block_sum_atomic<3>(v, dst);
    } else {
        double v[1] = {s11};
        double* const dst[1] = {out};
        block_sum_atomic<1>(v, dst);
    }
}

// ------------------------------------------------------------------------------------------------ K3 Schur complement
// (V_p + lam Dp^2)^-1 per point, symmetric 3x3 stored as xx xy xz yy yz zz
#ifndef SATBA_PV_STRIDE
#define SATBA_PV_STRIDE 16
#endif
constexpr int PV_STRIDE = SATBA_PV_STRIDE;  // doubles per packed point record: 12 used, padded to 16 so that a record is exactly one
                                            // 128-byte line (96-byte records straddle lines: 1.5 lines per gather; Schur 0.945 -> 0.845 ms)
// PV (optional): packed per-point record X(3) | Vinv(6) | g_p(3) for the gather-heavy Schur v3 kernels
// lam_dev (optional): the damping is read from device memory (satba_schur_auto) instead of the argument
__global__ void k_vinv(int N, double lam, const double* __restrict__ lam_dev, const double* __restrict__ V,
                       const double* __restrict__ scale_inv_p, double* __restrict__ Vinv, const double* __restrict__ xp,
                       const double* __restrict__ gp, double* __restrict__ PV) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    if (lam_dev) lam = *lam_dev;
    // 16-byte accesses: V and Vinv rows are 48 bytes apart, records 128 (hipMalloc aligns the arrays to 256 bytes)
    const double2* v2 = reinterpret_cast<const double2*>(V + 6 * (size_t)p);
    const double2 va = v2[0], vb = v2[1], vc = v2[2];
    const double* s = scale_inv_p + 3 * (size_t)p;
    const double a = va.x + lam * s[0] * s[0], b = va.y, c = vb.x;
    const double d = vb.y + lam * s[1] * s[1], e = vc.x, f = vc.y + lam * s[2] * s[2];
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double idet = 1.0 / (a * c00 + b * c01 + c * c02);
    const double o0 = c00 * idet, o1 = c01 * idet, o2 = c02 * idet;
    const double o3 = (a * f - c * c) * idet, o4 = (b * c - a * e) * idet, o5 = (a * d - b * b) * idet;
    double2* o = reinterpret_cast<double2*>(Vinv + 6 * (size_t)p);
    o[0] = make_double2(o0, o1); o[1] = make_double2(o2, o3); o[2] = make_double2(o4, o5);
    if (PV) {
        static_assert(PV_STRIDE % 2 == 0, "records are written as 16-byte words");
        double2* q = reinterpret_cast<double2*>(PV + PV_STRIDE * (size_t)p);
        const double x0 = xp[3 * (size_t)p], x1 = xp[3 * (size_t)p + 1], x2 = xp[3 * (size_t)p + 2];
        const double g0 = gp[3 * (size_t)p], g1 = gp[3 * (size_t)p + 1], g2 = gp[3 * (size_t)p + 2];
        q[0] = make_double2(x0, x1); q[1] = make_double2(x2, o0); q[2] = make_double2(o1, o2);
        q[3] = make_double2(o3, o4); q[4] = make_double2(o5, g0); q[5] = make_double2(g1, g2);
    }
}

// S <- (lead) * blockdiag(U_c + lam Dc^2), rhs <- (lead) * g_c ; S column-major n_c x n_c (already zeroed)
// use_U = 0: the J_c^T J_c blocks are added by k_schur_diag instead (only the damping goes on the diagonal here)
__global__ void k_schur_init(int M, int NP, double lam, const double* __restrict__ lam_dev, double lead, int use_U,
                             const double* __restrict__ U, const double* __restrict__ gc, const double* __restrict__ scale_inv,
                             double* __restrict__ S, double* __restrict__ rhs) {
    if (lam_dev) lam = *lam_dev;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_c = M * NP;
    if (idx < M * NP * NP) {
        const int cam = idx / (NP * NP), r = (idx / NP) % NP, c = idx % NP;
        double v = use_U ? U[idx] : 0.0;
        if (r == c) { const double s = scale_inv[cam * NP + r]; v += lam * s * s; }
        S[(size_t)(cam * NP + r) + (size_t)(cam * NP + c) * n_c] = lead * v;
    }
    if (idx < n_c) rhs[idx] = lead * gc[idx];
}

// Local Schur contributions of every point:  S -= W_a Vinv W_b^T for the observation pairs (a, b), a <= b, of
// the point (cameras ascend inside a point, so cam_a <= cam_b: block (cam_b, cam_a) of the column-major lower
// triangle);  rhs -= W_a Vinv g_p.   W = Jc^T Jp (NP x 3).
// v1: W of the tile staged in LDS, S accumulated with global float64 atomics.
template <int MODEL, int NP>
__global__ __launch_bounds__(256) void k_schur(ObsArgs a, const double* __restrict__ Vinv, const double* __restrict__ gp,
                                               double* __restrict__ S, double* __restrict__ rhs) {
    constexpr int WL = NP * 3;
    extern __shared__ double s_mem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* sW = s_mem + (size_t)wave * 64 * WL;  // [64][WL]
    double* s_rhs = s_mem + (size_t)4 * 64 * WL;  // [n_c]
    for (int i = threadIdx.x; i < a.n_c; i += 256) s_rhs[i] = 0.0;
    __syncthreads();
    for (int tile = blockIdx.x * 4 + wave; tile < a.n_tiles; tile += gridDim.x * 4) {
        if (a.tile_split[tile]) continue;  // handled by k_schur_split
        const int o0 = a.tile_start[tile], o1 = a.tile_start[tile + 1];
        const long long o = (long long)o0 + lane;
        const bool active = o < o1;
        int pt = -1 - lane, cam = 0;
        double T[NP][3];
        if (active) {
            cam = a.cam[o];
            pt = a.pt[o];
            ObsEval<MODEL, NP, true> e;
            e.eval(a, o, cam, pt);
            double W[NP][3];
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    W[i][j] = e.Jc[0][i] * e.Jp[0][j] + e.Jc[1][i] * e.Jp[1][j];
                    sW[lane * WL + i * 3 + j] = W[i][j];
                }
            const double* vi = Vinv + 6 * (size_t)pt;
            const double i00 = vi[0], i01 = vi[1], i02 = vi[2], i11 = vi[3], i12 = vi[4], i22 = vi[5];
            const double* g = gp + 3 * (size_t)pt;
            const double g0 = g[0], g1 = g[1], g2 = g[2];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                T[i][0] = W[i][0] * i00 + W[i][1] * i01 + W[i][2] * i02;
                T[i][1] = W[i][0] * i01 + W[i][1] * i11 + W[i][2] * i12;
                T[i][2] = W[i][0] * i02 + W[i][1] * i12 + W[i][2] * i22;
                atomicAdd(s_rhs + cam * NP + i, -(T[i][0] * g0 + T[i][1] * g1 + T[i][2] * g2));
            }
        }
        // run boundaries: this lane pairs with lanes lane .. end-1 of its run
        const int prev = __shfl_up(pt, 1);
        const unsigned long long heads = __ballot(lane == 0 || prev != pt);
        const unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1));
        const int end = above ? lane + 1 + __ffsll((long long)above) - 1 : 64;
        int span = active ? end - lane : 0;
        int maxspan = span;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) maxspan = max(maxspan, __shfl_xor(maxspan, d));
        // the shuffle of cam must be executed by every lane, so it sits outside the divergent branch
        for (int off = 0; off < maxspan; ++off) {
            const int b = min(lane + off, 63);
            const int camb = __shfl(cam, b);
            if (off < span) {
                const double* Wb = sW + b * WL;
                double* Sblk = S + (size_t)(camb * NP) + (size_t)(cam * NP) * a.n_c;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const double w0 = Wb[j * 3 + 0], w1 = Wb[j * 3 + 1], w2 = Wb[j * 3 + 2];
#pragma unroll
                    for (int i = 0; i < NP; ++i)
                        atomicAdd(Sblk + j + (size_t)i * a.n_c, -(T[i][0] * w0 + T[i][1] * w1 + T[i][2] * w2));
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.n_c; i += 256)
        if (s_rhs[i] != 0.0) atomicAdd(rhs + i, s_rhs[i]);
}

// slow path: one wave per point with more than 64 observations; pairs (a, b) over ALL its observations
template <int MODEL, int NP>
__global__ __launch_bounds__(64) void k_schur_split(ObsArgs a, int n_split, const int* __restrict__ split_pts,
                                                    const int* __restrict__ split_obs0, const int* __restrict__ split_obs1,
                                                    const double* __restrict__ Vinv, const double* __restrict__ gp,
                                                    double* __restrict__ S, double* __restrict__ rhs) {
    const int lane = threadIdx.x;
    for (int sidx = blockIdx.x; sidx < n_split; sidx += gridDim.x) {
        const int pt = split_pts[sidx];
        const int o0 = split_obs0[sidx], o1 = split_obs1[sidx];
        const double* vi = Vinv + 6 * (size_t)pt;
        const double i00 = vi[0], i01 = vi[1], i02 = vi[2], i11 = vi[3], i12 = vi[4], i22 = vi[5];
        const double g0 = gp[3 * (size_t)pt], g1 = gp[3 * (size_t)pt + 1], g2 = gp[3 * (size_t)pt + 2];
        for (int base = o0; base < o1; base += 64) {
            const int oa = base + lane;
            const bool active = oa < o1;
            int cam = 0;
            double T[NP][3];
            if (active) {
                cam = a.cam[oa];
                ObsEval<MODEL, NP, true> e;
                e.eval(a, oa, cam, pt);
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    double W[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) W[j] = e.Jc[0][i] * e.Jp[0][j] + e.Jc[1][i] * e.Jp[1][j];
                    T[i][0] = W[0] * i00 + W[1] * i01 + W[2] * i02;
                    T[i][1] = W[0] * i01 + W[1] * i11 + W[2] * i12;
                    T[i][2] = W[0] * i02 + W[1] * i12 + W[2] * i22;
                    atomicAdd(rhs + cam * NP + i, -(T[i][0] * g0 + T[i][1] * g1 + T[i][2] * g2));
                }
            }
            for (int ob = base; ob < o1; ++ob) {  // uniform over the wave
                const int camb = a.cam[ob];
                ObsEval<MODEL, NP, true> eb;
                eb.eval(a, ob, camb, pt);
                if (active && ob >= oa) {
                    double* Sblk = S + (size_t)(camb * NP) + (size_t)(cam * NP) * a.n_c;
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        const double w0 = eb.Jc[0][j] * eb.Jp[0][0] + eb.Jc[1][j] * eb.Jp[1][0];
                        const double w1 = eb.Jc[0][j] * eb.Jp[0][1] + eb.Jc[1][j] * eb.Jp[1][1];
                        const double w2 = eb.Jc[0][j] * eb.Jp[0][2] + eb.Jc[1][j] * eb.Jp[1][2];
#pragma unroll
                        for (int i = 0; i < NP; ++i)
                            atomicAdd(Sblk + j + (size_t)i * a.n_c, -(T[i][0] * w0 + T[i][1] * w1 + T[i][2] * w2));
                    }
                }
            }
        }
    }
}


// ---- Schur complement v2: camera-tile column panels of S privatised in LDS, no global atomics ----------------
// A workgroup owns the columns of S that belong to T consecutive cameras (NP * T columns x n_c rows of the
// column-major lower triangle = up to ~120 KB of LDS) and a chunk of the observation stream.  It walks the
// camera-major observation lists of its cameras inside the chunk; for an observation (i, p) it visits the
// observations (j, p), j >= i, of the same point (they follow it in the point-major stream), re-evaluates their
// Jacobian blocks and accumulates   -Jc_i^T (Jp_i Vinv_p Jp_j^T) Jc_j   into the panel with LDS atomics
// (ds_add_f64).  The panel is then stored, coalesced, into this chunk's private copy of S; k_schur_reduce sums
// the copies.  Points with any number of observations are handled uniformly (no wave tiles involved).
struct SchurArgs {
    const int* __restrict__ cam_ofs;   // M + 1: camera-major lists
    const int* __restrict__ cam_obs;   // K observation ids, ascending inside a camera
    const int* __restrict__ pt_ofs;    // N + 1: observations of point p are [pt_ofs[p], pt_ofs[p+1])
    const double* __restrict__ Vinv;   // N x 6
    const double* __restrict__ gp;     // N x 3
    double* __restrict__ S_part;       // n_chunks x n_c x n_c
    double* __restrict__ rhs_part;     // n_chunks x n_c
    int T, n_ctiles, n_chunks, camc_in_lds;
};

constexpr int SCHUR_THREADS = 1024;
constexpr int SCHUR_MAX_T = 16;

template <int MODEL, int NP, bool CL>
__global__ __launch_bounds__(SCHUR_THREADS) void k_schur_panel(ObsArgs a, SchurArgs s) {
    extern __shared__ double s_lds[];
    const int n_c = a.n_c;
    const int tile = blockIdx.x % s.n_ctiles, chunk = blockIdx.x / s.n_ctiles;
    const int i0 = tile * s.T, nt = min(s.T, a.M - i0);
    double* panel = s_lds;                               // [nt * NP][n_c]
    double* s_rhs = panel + (size_t)s.T * NP * n_c;      // [T * NP]
    double* s_camc = s_rhs + s.T * NP;                   // [M][CAMC] if camc_in_lds
    __shared__ int s_lo[SCHUR_MAX_T], s_cnt[SCHUR_MAX_T + 1];

    for (int i = threadIdx.x; i < s.T * NP * n_c + s.T * NP; i += SCHUR_THREADS) panel[i] = 0.0;
    if constexpr (CL)
        for (int i = threadIdx.x; i < a.M * CAMC; i += SCHUR_THREADS) s_camc[i] = a.camc[i];
    const long long obs_lo = a.K * chunk / s.n_chunks, obs_hi = a.K * (chunk + 1) / s.n_chunks;
    if (threadIdx.x < nt) {  // sub-range of camera i0 + t's list that falls into this chunk (binary searches)
        const int b = s.cam_ofs[i0 + threadIdx.x], e = s.cam_ofs[i0 + threadIdx.x + 1];
        int lo = b, hi = e;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (s.cam_obs[m] < obs_lo) lo = m + 1; else hi = m; }
        const int first = lo;
        hi = e;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (s.cam_obs[m] < obs_hi) lo = m + 1; else hi = m; }
        s_lo[threadIdx.x] = first;
        s_cnt[threadIdx.x + 1] = lo - first;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s_cnt[0] = 0;
        for (int t = 0; t < nt; ++t) s_cnt[t + 1] += s_cnt[t];
    }
    __syncthreads();
    const int total = s_cnt[nt];
    const double* cbase;
    if constexpr (CL) cbase = s_camc; else cbase = a.camc;

    for (int idx = threadIdx.x; idx < total; idx += SCHUR_THREADS) {
        int t = 0;
        while (idx >= s_cnt[t + 1]) ++t;
        const int o = s.cam_obs[s_lo[t] + idx - s_cnt[t]];
        const int cam_i = i0 + t;
        const int p = a.pt[o];
        ObsEval<MODEL, NP, true> ei;
        ei.eval(a, o, cam_i, p, cbase + (size_t)cam_i * CAMC);
        const double* vi = s.Vinv + 6 * (size_t)p;
        const double v00 = vi[0], v01 = vi[1], v02 = vi[2], v11 = vi[3], v12 = vi[4], v22 = vi[5];
        double A[2][3];  // Jp_i Vinv
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            A[r][0] = ei.Jp[r][0] * v00 + ei.Jp[r][1] * v01 + ei.Jp[r][2] * v02;
            A[r][1] = ei.Jp[r][0] * v01 + ei.Jp[r][1] * v11 + ei.Jp[r][2] * v12;
            A[r][2] = ei.Jp[r][0] * v02 + ei.Jp[r][1] * v12 + ei.Jp[r][2] * v22;
        }
        {
            const double* g = s.gp + 3 * (size_t)p;
            const double ag0 = A[0][0] * g[0] + A[0][1] * g[1] + A[0][2] * g[2];
            const double ag1 = A[1][0] * g[0] + A[1][1] * g[1] + A[1][2] * g[2];
#pragma unroll
            for (int r = 0; r < NP; ++r) atomicAdd(s_rhs + t * NP + r, -(ei.Jc[0][r] * ag0 + ei.Jc[1][r] * ag1));
        }
        const int o_end = s.pt_ofs[p + 1];
        double* prow = panel + (size_t)(t * NP) * n_c;
        for (int o2 = o; o2 < o_end; ++o2) {
            ObsEval<MODEL, NP, true> ej;
            int cam_j = cam_i;
            if (o2 == o) {
                ej = ei;
            } else {
                cam_j = a.cam[o2];
                ej.eval(a, o2, cam_j, p, cbase + (size_t)cam_j * CAMC);
            }
            // Mm = A Jp_j^T (2 x 2), Y = Mm Jc_j (2 x NP)
            const double m00 = A[0][0] * ej.Jp[0][0] + A[0][1] * ej.Jp[0][1] + A[0][2] * ej.Jp[0][2];
            const double m01 = A[0][0] * ej.Jp[1][0] + A[0][1] * ej.Jp[1][1] + A[0][2] * ej.Jp[1][2];
            const double m10 = A[1][0] * ej.Jp[0][0] + A[1][1] * ej.Jp[0][1] + A[1][2] * ej.Jp[0][2];
            const double m11 = A[1][0] * ej.Jp[1][0] + A[1][1] * ej.Jp[1][1] + A[1][2] * ej.Jp[1][2];
            double* dst = prow + cam_j * NP;
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                const double y0 = m00 * ej.Jc[0][c] + m01 * ej.Jc[1][c];
                const double y1 = m10 * ej.Jc[0][c] + m11 * ej.Jc[1][c];
#pragma unroll
                for (int r = 0; r < NP; ++r) atomicAdd(dst + (size_t)r * n_c + c, -(ei.Jc[0][r] * y0 + ei.Jc[1][r] * y1));
            }
        }
    }
    __syncthreads();
    double* out = s.S_part + (size_t)chunk * n_c * n_c + (size_t)(i0 * NP) * n_c;
    for (int i = threadIdx.x; i < nt * NP * n_c; i += SCHUR_THREADS) out[i] = panel[i];
    for (int i = threadIdx.x; i < nt * NP; i += SCHUR_THREADS) s.rhs_part[(size_t)chunk * n_c + i0 * NP + i] = s_rhs[i];
}

// S += sum over chunks of S_part, rhs += sum of rhs_part
__global__ __launch_bounds__(256) void k_schur_reduce(int n_c, int n_chunks, const double* __restrict__ S_part,
                                                      const double* __restrict__ rhs_part, double* __restrict__ S,
                                                      double* __restrict__ rhs) {
    const size_t nn = (size_t)n_c * n_c;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nn + n_c; i += (size_t)gridDim.x * blockDim.x) {
        if (i < nn) {
            double t = 0.0;
            for (int c = 0; c < n_chunks; ++c) t += S_part[(size_t)c * nn + i];
            S[i] += t;
        } else {
            const size_t k = i - nn;
            double t = 0.0;
            for (int c = 0; c < n_chunks; ++c) t += rhs_part[(size_t)c * n_c + k];
            rhs[k] += t;
        }
    }
}

// Reduced system in scaled variables: S <- diag(scale) S diag(scale), rhs <- scale * rhs (scale = 1 / scale_inv).
// x_scale="jac" makes the scaled matrix unit-diagonal up to the damping, which keeps the dense factorisation
// well conditioned although the raw camera blocks span ~12 orders of magnitude (angles vs translations).
__global__ __launch_bounds__(256) void k_scale_system(int n_c, const double* __restrict__ scale_inv, double* __restrict__ S,
                                                      const double* __restrict__ rhs, double* __restrict__ rhs_scaled) {
    const size_t nn = (size_t)n_c * n_c;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nn + n_c; i += (size_t)gridDim.x * blockDim.x) {
        if (i < nn) {
            const int r = (int)(i % n_c), c = (int)(i / n_c);
            if (r >= c) S[i] /= scale_inv[r] * scale_inv[c];
        } else {
            rhs_scaled[i - nn] = rhs[i - nn] / scale_inv[i - nn];  // the solve works on a copy: the payload is all-reduced data
        }
    }
}

// dc = dc_h / scale_inv (unscaled camera step for the back-substitution).  The same launch prepares the header of the
// solve phase (nothing touches it between here and k_backsub_finish): zero, Cholesky status in slot 4, the scalars
// kept from the earlier phases of this iteration in slots SATBA_HDR_KEEP.. (rank 0 only: headers are summed over ranks)
__global__ void k_unscale(int n_c, const double* __restrict__ scale_inv, const double* __restrict__ dch, double* __restrict__ dc,
                          int hdr_len, double* __restrict__ hdr, const int* __restrict__ fail_flag, double lead,
                          const double* __restrict__ keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_c) dc[i] = dch[i] / scale_inv[i];
    if (i < hdr_len) {
        double v = 0.0;
        if (i == 4) v = (*fail_flag != 0) ? lead : 0.0;
        if (i >= SATBA_HDR_KEEP && i < SATBA_HDR_KEEP + SATBA_KEEP_LEN) v = lead * keep[i - SATBA_HDR_KEEP];
        hdr[i] = v;
    }
}

// start of the prepare phase, one launch: the (already all-reduced) linearize payload U | g_c is copied out of the
// exchange buffer, keep[0] = cost and keep[1] = max_rank |g_p|_inf are taken from its header, the header is zeroed
__global__ __launch_bounds__(1024) void k_prepare_stash(int nU, int n_c, int world, int hdr_len, double* __restrict__ xb,
                                                        double* __restrict__ U, double* __restrict__ gc, double* __restrict__ keep) {
    const double* payload = xb + hdr_len;
    for (int i = threadIdx.x; i < nU + n_c; i += blockDim.x) {
        if (i < nU) U[i] = payload[i];
        else gc[i - nU] = payload[i];
    }
    if (threadIdx.x == 0) {
        double m = 0.0;
        for (int r = 0; r < world; ++r) m = fmax(m, xb[SATBA_HDR_FIXED + r]);
        keep[0] = xb[0];
        keep[1] = m;
    }
    __syncthreads();
    if ((int)threadIdx.x < hdr_len) xb[threadIdx.x] = 0.0;
}

// ------------------------------------------------------------------------------------------------ K5 back-substitution
// t_p = sum_obs Jp^T (Jc dc[cam])  per point (segmented wave reduction; for three values the shuffle form beats the
// LDS-staged form of k_linearize: 0.25 vs 0.28 ms at C4)
// Per-point sums of three values per observation over a wave tile (runs of equal point index), through the wave's LDS
// staging rows as in k_linearize: 3 ds_write + ~12 ds_read per tile.  The shuffle form (seg_reduce<3>) issues 43
// ds_bpermute per tile and kept the back-substitution bound by the LDS pipe (0.132 -> 0.083 ms at the headline shape).
// stage: 3 * LIN_STAGE doubles, seg: 66 bytes, both private to the wave; n_obs = observations of the tile.
__device__ inline void point_sums3(double* stage, unsigned char* seg, const double (&v)[3], int pt, bool active, int lane,
                                   int n_obs, bool split, double* __restrict__ tbuf) {
    if (active) {
        stage[0 * LIN_STAGE + lane] = v[0];
        stage[1 * LIN_STAGE + lane] = v[1];
        stage[2 * LIN_STAGE + lane] = v[2];
    }
    const int prev = __shfl_up(pt, 1);
    const bool head = active && (lane == 0 || prev != pt);
    const unsigned long long heads = __ballot(head);
    const int n_runs = __popcll(heads);
    if (head) seg[__popcll(heads & ((1ull << lane) - 1ull))] = (unsigned char)lane;
    if (lane == 0) seg[n_runs] = (unsigned char)n_obs;
    for (int r0 = 0; r0 < n_runs; r0 += 21) {
        const int q = r0 + lane / 3, vi = lane % 3;
        const bool owner = lane < 63 && q < n_runs;
        const int rb = owner ? seg[q] : 0, re = owner ? seg[q + 1] : 0;
        const int ptq = __shfl(pt, rb);
        if (owner) {
            const double* col = stage + vi * LIN_STAGE;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            for (int l = rb; l < re; l += 4) {
                const double a0 = col[l];
                const double a1 = (l + 1 < re) ? col[l + 1] : 0.0;
                const double a2 = (l + 2 < re) ? col[l + 2] : 0.0;
                const double a3 = (l + 3 < re) ? col[l + 3] : 0.0;
                s0 += a0; s1 += a1; s2 += a2; s3 += a3;
            }
            const double sum = (s0 + s1) + (s2 + s3);
            double* t = tbuf + 3 * (size_t)ptq + vi;
            if (split) atomicAdd(t, sum);
            else *t = sum;
        }
    }
}

// Affine cameras: the projection is exactly affine in the point, so J_c dc = B_c X + b_c with per-camera constants
// B_c = sum_i dc_i D_ci (2 x 3), b_c = K-columns . dc_T, and J_p = A_c.  Every workgroup derives the 14 constants of each
// camera once (three evaluations of the projector's Jacobian at the unit vectors) into an LDS table with an odd row
// stride; an observation then costs 14 LDS reads and 14 multiply-adds instead of 11 + 5 gathers (the dc gathers went
// through the texture path) and the full Jacobian evaluation.
constexpr int BS_ROW = 15;
template <int MODEL, int NP, bool CL>
__global__ __launch_bounds__(256) void k_backsub(ObsArgs a, const double* __restrict__ dc, double* __restrict__ tbuf) {
    extern __shared__ double s_camc_bs[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ double s_stage[4][3 * LIN_STAGE];
    __shared__ unsigned char s_seg[4][66];
    double* stage = s_stage[wave];
    if constexpr (MODEL == AFFINE) {
        double* tab = s_camc_bs;  // M x BS_ROW: B (6) | b (2) | A (6)
        for (int c = threadIdx.x; c < a.M; c += 256) {
            const double* cc = a.camc + (size_t)c * CAMC;
            double u, v, Jc[2][NP], Jp[2][3], B[2][3], b[2] = {0.0, 0.0};
            const double mc = (c >= a.n_cam_fix) ? 1.0 : 0.0;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                project<AFFINE, NP, true>(cc, nullptr, m == 0 ? 1.0 : 0.0, m == 1 ? 1.0 : 0.0, m == 2 ? 1.0 : 0.0, false, u, v, Jc, Jp);
                B[0][m] = 0.0; B[1][m] = 0.0;
#pragma unroll
                for (int i = 0; i < 3; ++i) { B[0][m] += Jc[0][i] * dc[c * NP + i]; B[1][m] += Jc[1][i] * dc[c * NP + i]; }
            }
#pragma unroll
            for (int i = 3; i < NP; ++i) { b[0] += Jc[0][i] * dc[c * NP + i]; b[1] += Jc[1][i] * dc[c * NP + i]; }
            double* row = tab + (size_t)c * BS_ROW;
#pragma unroll
            for (int m = 0; m < 3; ++m) { row[m] = mc * B[0][m]; row[3 + m] = mc * B[1][m]; row[8 + m] = Jp[0][m]; row[11 + m] = Jp[1][m]; }
            row[6] = mc * b[0]; row[7] = mc * b[1];
        }
        __syncthreads();
        // software pipeline as in k_linearize: a wave walks ~20 tiles and each one was a chain of three dependent
        // loads (range -> record -> point); ranges run three tiles ahead, records two, point gathers one
        const int stride = gridDim.x * 4;
        int tile = blockIdx.x * 4 + wave;
        auto range = [&](int t, int& r0, int& r1, int& rs) {
            const int tu = __builtin_amdgcn_readfirstlane(t);
            if (tu < a.n_tiles) { r0 = a.tile_start[tu]; r1 = a.tile_start[tu + 1]; rs = a.tile_split[tu]; } else { r0 = 0; r1 = 0; rs = 0; }
        };
        int o0, o1, osplit, n0, n1, nsplit, m0, m1, msplit;
        range(tile, o0, o1, osplit);
        range(tile + stride, n0, n1, nsplit);
        range(tile + 2 * stride, m0, m1, msplit);
        int cam = 0, pt = -1 - lane, ncam = 0, npt = -1 - lane, mcam = 0, mpt = -1 - lane;
        double X = 0.0, Y = 0.0, Z = 0.0, nX = 0.0, nY = 0.0, nZ = 0.0;
        if (o0 + lane < o1) { pt = a.pt[o0 + lane]; cam = a.cam[o0 + lane]; }
        if (n0 + lane < n1) { npt = a.pt[n0 + lane]; ncam = a.cam[n0 + lane]; }
        if (pt >= 0) { const double* px = a.x + a.n_c + 3 * (size_t)pt; X = px[0]; Y = px[1]; Z = px[2]; }
        for (; tile < a.n_tiles; tile += stride) {
            const long long o = (long long)o0 + lane;
            const bool active = o < o1;
            int q0, q1, qsplit;
            if (npt >= 0) { const double* px = a.x + a.n_c + 3 * (size_t)npt; nX = px[0]; nY = px[1]; nZ = px[2]; }
            mpt = -1 - lane; mcam = 0;
            if (m0 + lane < m1) { mpt = a.pt[m0 + lane]; mcam = a.cam[m0 + lane]; }
            range(tile + 3 * stride, q0, q1, qsplit);
            __builtin_amdgcn_sched_barrier(0);
            double v[3] = {0, 0, 0};
            if (active) {
                const double* row = tab + (size_t)cam * BS_ROW;
                double u0 = row[0] * X + row[1] * Y + row[2] * Z + row[6];
                double u1 = row[3] * X + row[4] * Y + row[5] * Z + row[7];
                if (a.sc) { const double2 t = a.sc[o]; u0 *= t.x * t.x; u1 *= t.y * t.y; }  // both blocks carry the row scale
                const double mp = (pt >= a.n_pts_fix) ? 1.0 : 0.0;
                u0 *= mp; u1 *= mp;
#pragma unroll
                for (int j = 0; j < 3; ++j) v[j] = row[8 + j] * u0 + row[11 + j] * u1;
            }
            point_sums3(stage, s_seg[wave], v, pt, active, lane, o1 - o0, osplit != 0, tbuf);
            o0 = n0; o1 = n1; osplit = nsplit; n0 = m0; n1 = m1; nsplit = msplit; m0 = q0; m1 = q1; msplit = qsplit;
            cam = ncam; pt = npt; X = nX; Y = nY; Z = nZ; ncam = mcam; npt = mpt;
        }
        return;
    }
    const double* cbase = cam_table<CL>(a, s_camc_bs, 256);
    for (int tile = blockIdx.x * 4 + wave; tile < a.n_tiles; tile += gridDim.x * 4) {
        const int o0 = a.tile_start[tile], o1 = a.tile_start[tile + 1];
        const long long o = (long long)o0 + lane;
        const bool active = o < o1;
        int pt = -1 - lane;
        double v[3] = {0, 0, 0};
        if (active) {
            const int cam = a.cam[o];
            pt = a.pt[o];
            ObsEval<MODEL, NP, true> e;
            e.jac(a, o, cam, pt, cbase + (size_t)cam * CAMC);
            double u0 = 0.0, u1 = 0.0;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const double d = dc[cam * NP + i];  // global gather (L1 hits); an LDS copy of dc measured slower twice
                u0 += e.Jc[0][i] * d;
                u1 += e.Jc[1][i] * d;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) v[j] = e.Jp[0][j] * u0 + e.Jp[1][j] * u1;
        }
        point_sums3(stage, s_seg[wave], v, pt, active, lane, o1 - o0, a.tile_split[tile] != 0, tbuf);
    }
}

// gn_h = [dc_h ; scale_inv_p * Vinv (g_p - t)]  and the Gram matrix of (g_h, gn_h): hdr[1..3] += a, b, c
__global__ __launch_bounds__(256) void k_backsub_finish(int n_c, int N, double lead, const double* __restrict__ dch,
                                                        const double* __restrict__ Vinv, const double* __restrict__ g,
                                                        const double* __restrict__ tbuf, const double* __restrict__ scale_inv,
                                                        const double* __restrict__ gh, double* __restrict__ gn,
                                                        double* __restrict__ hdr) {
    double sa = 0.0, sb = 0.0, sc = 0.0;
    const int total = n_c + N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (i < n_c) {
            const double v = dch[i], h = gh[i];
            gn[i] = v;
            sa += lead * h * h; sb += lead * h * v; sc += lead * v * v;
        } else {
            const size_t p = i - n_c;
            const double* vi = Vinv + 6 * p;
            const size_t base = (size_t)n_c + 3 * p;
            const double r0 = g[base] - tbuf[3 * p], r1 = g[base + 1] - tbuf[3 * p + 1], r2 = g[base + 2] - tbuf[3 * p + 2];
            const double d[3] = {vi[0] * r0 + vi[1] * r1 + vi[2] * r2, vi[1] * r0 + vi[3] * r1 + vi[4] * r2,
                                 vi[2] * r0 + vi[4] * r1 + vi[5] * r2};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double v = d[k] * scale_inv[base + k], h = gh[base + k];
                gn[base + k] = v;
                sa += h * h; sb += h * v; sc += v * v;
            }
        }
    }
    double v[3] = {sa, sb, sc};
    double* const dst[3] = {hdr + 1, hdr + 2, hdr + 3};
    block_sum_atomic<3>(v, dst);
}

// ------------------------------------------------------------------------------------------------ subspace / trial vectors
// q1 = s g_h, w = gn_h - alpha g_h;  hdr[1] += w.w, hdr[2] += w.q1, hdr[6] += g_h.w
__global__ __launch_bounds__(256) void k_subspace_vec(int n, int n_c, double lead, double alpha, double s,
                                                      const double* __restrict__ gh, const double* __restrict__ gn,
                                                      double* __restrict__ q1, double* __restrict__ wv,
                                                      double* __restrict__ hdr) {
    double ww = 0.0, wq = 0.0, gw = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double h = gh[i], q = s * h, w = gn[i] - alpha * h;
        q1[i] = q;
        wv[i] = w;
        const double wgt = (i < n_c) ? lead : 1.0;
        ww += wgt * w * w; wq += wgt * w * q; gw += wgt * h * w;
    }
    double v[3] = {ww, wq, gw};
    double* const dst[3] = {hdr + 1, hdr + 2, hdr + 6};
    block_sum_atomic<3>(v, dst);
}

// x_new = x + (p0 q1 + p1 w) / scale_inv;  hdr[2] += |step|^2, hdr[3] += |x|^2
__global__ __launch_bounds__(256) void k_trial_vec(int n, int n_c, double lead, double p0, double p1,
                                                   const double* __restrict__ x, const double* __restrict__ q1,
                                                   const double* __restrict__ wv, const double* __restrict__ scale_inv,
                                                   double* __restrict__ x_new, double* __restrict__ hdr) {
    double ss = 0.0, xx = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double step = (p0 * q1[i] + p1 * wv[i]) / scale_inv[i];
        const double xi = x[i];
        x_new[i] = xi + step;
        const double wgt = (i < n_c) ? lead : 1.0;
        ss += wgt * step * step; xx += wgt * xi * xi;
    }
    double v[2] = {ss, xx};
    double* const dst[2] = {hdr + 2, hdr + 3};
    block_sum_atomic<2>(v, dst);
}

// ------------------------------------------------------------------------------------------------ inspection
template <int MODEL, int NP>
__global__ void k_jacobian(ObsArgs a, double* __restrict__ Jc, double* __restrict__ Jp) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < a.K; o += (long long)gridDim.x * blockDim.x) {
        ObsEval<MODEL, NP, true> e;
        e.eval(a, o, a.cam[o], a.pt[o]);
        for (int r = 0; r < 2; ++r) {
            for (int i = 0; i < NP; ++i) Jc[(o * 2 + r) * NP + i] = e.Jc[r][i];
            for (int j = 0; j < 3; ++j) Jp[(o * 2 + r) * 3 + j] = e.Jp[r][j];
        }
    }
}

}  // namespace satba
