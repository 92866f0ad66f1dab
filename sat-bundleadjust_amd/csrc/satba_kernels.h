// satba_kernels.h -- HIP kernels of the bundle-adjustment hot path (gfx950, wave64): residuals, linearisation, Jacobian-
// vector products, back-substitution and the vector kernels of the trust-region loop.
//
// Work decomposition.  Observations are stored in SLICED-ELL order (satba_layout.h): 64 consecutive points -- sorted by
// track length -- form a slice, slot k of the 64 points is contiguous.  One wavefront owns a slice, ONE LANE OWNS A POINT
// and walks its observations k = 0 .. count-1:
//   * every read of the observation arrays is a coalesced 64-lane access (cam 256 B, obs 1 KB per wave instruction);
//   * per-point sums (V_p, g_p, W^T dc) accumulate in the lane's registers -- no cross-lane reduction, no LDS staging,
//     no atomics, tracks of any length, and a fixed summation order (bitwise repeatable);
//   * per-camera sums (diag U_c, g_c) go through ds_add_f64 into a per-workgroup LDS table that is flushed once per
//     workgroup (or, on request, through a camera-major register pass with a fixed order: k_cam_sums).
// Round 1 used lane = observation with wave tiles of whole points: its per-point sums cost 9 LDS writes + ~12 LDS reads per
// tile and kept k_linearize bound by LDS instruction issue at 43 % of HBM peak.
//
// Nothing here is GEMM shaped (2x3, 2x6, 3x3 blocks): MFMA is not used; the kernels are bound by HBM traffic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "satba_models.h"

namespace satba {

__host__ __device__ constexpr int cam_acc_len(int np) { return np * (np + 1) / 2 + np; }
// row stride of the LDS camera table: odd, so that (stride * cam + k) visits all 32 bank pairs (an even stride such
// as 20 folds the cameras onto 8 of them: 8-way conflicts of the ds_add_f64)
__host__ __device__ constexpr int cam_acc_stride(int np) { return cam_acc_len(np) | 1; }
// k_linearize's LDS table holds diag(U_c) and g_c only, g_c in two 64-bit limbs: 3 np sums per camera, odd row stride (9, 15, 19);
// the per-workgroup partials have the same 3 np words per camera.  (A 32-bit low limb -- half the cost of a 64-bit atomic, 0.105
// instead of 0.112 ms at 200 x 1M x 10M -- carries 15 bits with 50 000 observations per camera: 61 bits below the term bound in
// all.  That is coarser than a float64 sum in a fixed order resolves near the solution, and the ratio test of scipy's ftol
// criterion, which lives on differences of 1e-10 of the cost there, tripped four evaluations later: nfev 9 instead of 5 for the
// headline solve.  With 40 more bits the sums are exact and the solve ends where round 2's did.)
__host__ __device__ constexpr int cam_sum_len(int np) { return 3 * np; }
__host__ __device__ constexpr int cam_sum_stride(int np) { return (3 * np) | 1; }
constexpr int FX_LO_SHIFT = 40;  // the low limb of a g_c term holds its remainder below the high limb's unit, scaled by 2^40
// The high limbs are scaled per problem for the camera with the most observations (k_lin_scales: Q = 62 - log2 n_max) and range-checked
// per term; the low limb is not: a term is at most 2^(FX_LO_SHIFT - 1) there, so its sum stays inside 63 bits for fewer than
// 2^(63 - 39) = 2^24 observations of one camera.  Problems above this bound (with a factor 2 to spare) take the camera-major sums.
constexpr long long FX_MAX_OBS_PER_CAM = 1ll << 23;
// (a multiple of 16 bytes: the camera-constant rows behind the table are read with ds_read_b128)
__host__ __device__ constexpr size_t cam_sum_bytes(int np, size_t rows) { return (rows * cam_sum_stride(np) * 8 + 15) & ~(size_t)15; }

// row stride of the LDS copy of the RPC tables: 90 doubles used, 47 sixteen-byte slots (odd: the rows of consecutive cameras start in all
// 16 slots of the bank row).  Round 6: with 91 doubles (8-byte alignment) the 90 reads of an evaluation were 45 ds_read2_b64 pairs -- half
// the rate of the 45 aligned ds_read_b128 they are now (satba_models.h: CAMC)
constexpr int RPCS = 94;
static_assert(RPCS >= 90 && RPCS % 2 == 0 && (RPCS / 2) % 2 == 1, "RPC rows in the LDS: an odd number of 16-byte slots");
// double -> 64-bit fixed point through one fma (k_linearize's camera sums): bits(t 2^e + 1.5 * 2^52) = FX_MAGIC_BITS + round(t 2^e)
constexpr int SATBA_HDR_PREP_GH = 6, SATBA_HDR_PREP_XS = 7;  // linearize header: point sums of |g_h|^2, |x_h|^2 (prepare fused into k_linearize)
constexpr int SATBA_HDR_FX = 5;  // linearize header: a term of the fixed-point camera sums exceeded its bound (summed over ranks)
constexpr int SATBA_K_FX = 7;    // ... and its place among the kept scalars (header slot SATBA_HDR_KEEP + 7 after satba_solve)
constexpr double FX_MAGIC = 6755399441055744.0;
constexpr unsigned long long FX_MAGIC_BITS = 0x4338000000000000ull;
// RPC: what a linearisation stores per observation for the passes behind it (io order): D' = diag(row scales) d(col,row)/dX', 2 x 3 =
// six doubles in a 64-byte row (half a line; never straddles one).  The passes rebuild Jc = mc D' dR(X - T - C), Jp = mp D' R from it
// (rpc_jac_from_d: ~45 multiply-adds).  Rounds 1 - 4 stored the blocks Jc | Jp themselves, 128 bytes (3 parameters) or 192 bytes (6) per
// observation: the pair kernel gathered two such rows per hit.
__host__ __device__ constexpr int jrow_stride(int) { return 8; }

// Device-resident LM loop (satba_lmdev.h): the host queues a fixed pattern of kernels per iteration without knowing whether the
// previous trial step was accepted; every kernel of the pattern starts by reading its gate -- a word of the loop's state in device
// memory -- and returns when it is 0.  Null: no gate (the phase entry points of the C ABI).
#ifndef SATBA_GATE
#define SATBA_GATE(g) do { if ((g) != nullptr && *(g) == 0) return; } while (0)
#endif

struct ObsArgs {
    const int* __restrict__ e_cam;       // P: camera of every ELL slot (-1: padding)
    const double2* __restrict__ e_obs;   // P: observed (col, row)
    const double* __restrict__ e_w;      // P: weights
    const int* __restrict__ slice_base;  // n_slices + 1: first ELL position of every slice
    const int* __restrict__ pt_cnt;      // N: track length of internal point q
    const int* __restrict__ perm;        // N: caller's local index of internal point q (fixed points: perm < n_pts_fix)
    const int* __restrict__ ipt_ofs;     // N + 1: io index (internal point-major) of every point's first observation
    const double* __restrict__ x;        // variable vector [cameras | internal points] whose POINT part is used
    const double* __restrict__ camc;     // M x CAMC camera constants built from the same vector
    const double* __restrict__ rpc;      // M x 90 or null
    double* __restrict__ Jpm;            // RPC only (else null): K x jrow_stride: D' = row-scaled d(col,row)/dX' of the current
                                         // linearisation, io order; written by the linearize kernel and read by every later
                                         // pass (the RPC chain costs 2-3 kflop per evaluation)
    double2* __restrict__ sc;            // weighted / robust runs of the affine and perspective models (else null): the Jacobian row scales
                                         // (w js0, w js1) of the current linearisation, written by the linearize kernel INTO the
                                         // merged point records (Layout::w_fix): observation k of internal point q at piece
                                         // ipt_ofs[q] + k, where `ipt_ofs` is then Layout::sc_ofs (rounds 1 - 4: an array of its own
                                         // in io order).  RPC: null -- the scales ride in the stored D'
    long long K;
    int P, n_slices, M, N, n_c, n_cam_fix, n_pts_fix, loss, f32;
    int unit;                            // every weight is 1 and the loss is linear
    int rep_shift;                       // log2 of the number of replicas of k_linearize's LDS camera table (few cameras)
    int sh;                              // log2 of the lanes per point (slice_unit): 0 for large problems
    int rev = 0;                         // for_each_slice: the slices in the opposite order in time
    double f_scale;
    const int* __restrict__ fxe;         // fixed-point scales of k_linearize's camera sums (k_lin_scales): exponents a_0 .. a_{NP-1}, b, then
                                         // the two constants of the range check
    int* __restrict__ fx_flag;           // set when a term leaves its range (the sums are then formed by k_cam_sums instead)
    const int* gate;                     // SATBA_GATE
    double* dir_tab;                     // affine cameras beyond the LDS (k_jvp / k_backsub with DG): their direction tables in global
                                         // memory (2 x M x JVP_ROW), filled by k_affine_dir_tab in front of the kernel
    // k_linearize, single-rank loops: the point part of the prepare phase (x_scale="jac" update, g_h, g_h / scale_inv, and the point
    // sums of |g_h|^2 and |x_h|^2) is done in the lane that has just formed the point's blocks -- the vector kernel of the prepare
    // phase then only visits the camera entries (it read V, g, x, scale_inv of 3 N entries again: 43 us at 1 M points).  Null: not fused.
    double* prep_scale;                  // scale_inv (whole vector, indexed like x)
    double* prep_gh;                     // g_h
    double* prep_ghs;                    // g_h / scale_inv
    const int* prep_first_dev;           // first linearisation of a solve (device-resident loop), else prep_first
    int prep_first;
};

// deterministic grid-wide sums: every workgroup writes its partial, the last one to arrive adds them up in index order
struct RedBuf {
    double* part;       // [NV][gridDim.x]
    unsigned* counter;  // zero between launches
};

// weighted, robust-scaled residual and (optionally) Jacobian blocks of one observation
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
    double Dr[2][3];  // RPC with JAC: d(col,row)/dX' (eval: unscaled; store_jac scales it)

    // cc: the camera's constant record, tab: its RPC table (global memory or LDS copies); mp: 0 for a fixed point
    __device__ inline void eval(const ObsArgs& a, int cam, double mp, const double* cc, const double* tab, const double2 ob,
                                const double w, const double X, const double Y, const double Z) {
        double u, v;
        // RPC: the chain leaves D; the blocks are formed from it BEHIND the loss function below (with them alive across the generic
        // loss code the robust variants spilled 880 bytes per thread)
        if constexpr (MODEL == RPC) project_rpc_d<NP, JAC, false>(cc, tab, X, Y, Z, a.f32 != 0, u, v, Jc, Jp, Dr);
        else project<MODEL, NP, JAC>(cc, tab, X, Y, Z, a.f32 != 0, u, v, Jc, Jp);
        if constexpr (UNITW && !ROBUST) {
            if constexpr (MODEL == RPC && JAC) rpc_jac_from_d<NP>(cc, X, Y, Z, Dr, Jc, Jp);
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
            robust(loss, a.f_scale, ftrue[0], r0, fs[0], js0, MODEL != RPC);
            robust(loss, a.f_scale, ftrue[1], r1, fs[1], js1, MODEL != RPC);
        } else {  // linear loss, specialised at compile time (no transcendental code, far fewer registers)
            fs[0] = ftrue[0]; fs[1] = ftrue[1];
            r0 = ftrue[0] * ftrue[0]; r1 = ftrue[1] * ftrue[1];
        }
        rho = r0 + r1;
        if (JAC) {
            if constexpr (MODEL == RPC) rpc_jac_from_d<NP>(cc, X, Y, Z, Dr, Jc, Jp);
            const double mc = (cam >= a.n_cam_fix) ? 1.0 : 0.0;
            const double s0 = w * js0, s1 = w * js1;
            sw[0] = s0; sw[1] = s1;
#pragma unroll
            for (int i = 0; i < NP; ++i) { Jc[0][i] *= s0 * mc; Jc[1][i] *= s1 * mc; }
#pragma unroll
            for (int j = 0; j < 3; ++j) { Jp[0][j] *= s0 * mp; Jp[1][j] *= s1 * mp; }
        }
    }

    // RPC: D' of the current linearisation, stored (after eval) / reloaded: three 16-byte words
    __device__ inline void store_jac(const ObsArgs& a, int io) const {
        double2* q = reinterpret_cast<double2*>(a.Jpm + (size_t)io * jrow_stride(NP));
        q[0] = make_double2(sw[0] * Dr[0][0], sw[0] * Dr[0][1]);
        q[1] = make_double2(sw[0] * Dr[0][2], sw[1] * Dr[1][0]);
        q[2] = make_double2(sw[1] * Dr[1][1], sw[1] * Dr[1][2]);
    }
    __device__ inline void load_d(const ObsArgs& a, int io) {
        const double2* q = reinterpret_cast<const double2*>(a.Jpm + (size_t)io * jrow_stride(NP));
        const double2 q0 = q[0], q1 = q[1], q2 = q[2];
        Dr[0][0] = q0.x; Dr[0][1] = q0.y; Dr[0][2] = q1.x; Dr[1][0] = q1.y; Dr[1][1] = q2.x; Dr[1][2] = q2.y;
    }
    // Jacobian blocks only, for the passes that follow a linearisation at the same x: from the store when there is one
    // (RPC); otherwise the unit-weight, linear-loss Jacobian times the row scales the linearize kernel stored (a.sc;
    // null when every weight is 1 and the loss is linear).  Neither the observation nor its weight is read and the
    // loss function is not evaluated.
    __device__ inline void jac(const ObsArgs& a, int io, int cam, double mp, const double* cc, const double* tab,
                               const double X, const double Y, const double Z) {
        if constexpr (MODEL == RPC) {  // always stored for RPC cameras: the blocks from D' (it carries the row scales), masks applied here
            load_d(a, io);
            rpc_jac_from_d<NP>(cc, X, Y, Z, Dr, Jc, Jp);
            const double mc = (cam >= a.n_cam_fix) ? 1.0 : 0.0;
#pragma unroll
            for (int i = 0; i < NP; ++i) { Jc[0][i] *= mc; Jc[1][i] *= mc; }
#pragma unroll
            for (int j = 0; j < 3; ++j) { Jp[0][j] *= mp; Jp[1][j] *= mp; }
            return;
        }
        double u, v;
        project<MODEL, NP, true>(cc, tab, X, Y, Z, false, u, v, Jc, Jp);
        double s0 = 1.0, s1 = 1.0;
        if (a.sc) { const double2 t = a.sc[io]; s0 = t.x; s1 = t.y; }
        const double mc = (cam >= a.n_cam_fix) ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < NP; ++i) { Jc[0][i] *= s0 * mc; Jc[1][i] *= s1 * mc; }
#pragma unroll
        for (int j = 0; j < 3; ++j) { Jp[0][j] *= s0 * mp; Jp[1][j] *= s1 * mp; }
    }
};

// one observation record of the ELL stream (camera, observed pixel, weight); `on` false: nothing is fetched
#ifndef SATBA_PF
#define SATBA_PF 2  // slots of the ELL stream in flight ahead of the one being evaluated
#endif
struct ObsRec {
    int cam = 0;
    double2 ob = {0.0, 0.0};
    double w = 1.0;
    template <bool UNITW>
    __device__ inline void load(const ObsArgs& a, int pos, bool on) {
        if (on) {
            cam = a.e_cam[pos];
            ob = a.e_obs[pos];
            if constexpr (!UNITW) w = a.e_w[pos];
        }
    }
};

// Per-camera tables in LDS.  CL / RL = true: the workgroup stages the whole camera-constant / RPC table in dynamic LDS once
// and every lookup is a ds_read (the pointers never merge with global ones, so no FLAT instructions are generated; odd row
// strides: no structural bank conflicts); false (table too large for LDS): gathers from global memory through L1.
template <bool CL, bool RL>
struct CamTables {
    const double* cbase;
    const double* rbase;
    int rstride;
    __device__ static size_t doubles(int M) { return (CL ? (size_t)M * CAMC : 0) + (RL ? (size_t)M * RPCS : 0); }
    // s: dynamic LDS area of doubles(M) doubles; ends with a barrier when anything was staged
    __device__ inline void stage(const ObsArgs& a, double* s, int nthreads) {
        if constexpr (CL) {
            for (int i = threadIdx.x; i < a.M * CAMC; i += nthreads) s[i] = a.camc[i];
            cbase = s;
        } else {
            cbase = a.camc;
        }
        if constexpr (RL) {
            double* r = s + (CL ? (size_t)a.M * CAMC : 0);
            for (int i = threadIdx.x; i < a.M * 90; i += nthreads) r[(i / 90) * RPCS + i % 90] = a.rpc[i];
            rbase = r;
            rstride = RPCS;
        } else {
            rbase = a.rpc;
            rstride = 90;
        }
        if constexpr (CL || RL) __syncthreads();
    }
    // (rows are 16-byte aligned in the LDS and in global memory: the pairs of a row are read as ds_read_b128 / global_load_dwordx4)
    __device__ inline const double* cc(int cam) const { return static_cast<const double*>(__builtin_assume_aligned(cbase + (size_t)cam * CAMC, 16)); }
    // (only dereferenced for RPC; 16-byte aligned rows in both places: 94 doubles in the LDS, 90 in global memory)
    __device__ inline const double* tab(int cam) const { return static_cast<const double*>(__builtin_assume_aligned(rbase + (size_t)cam * rstride, 16)); }
};

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

// Sum NV per-thread values over the whole grid and store each total in *dst[k]: wave shuffles, per-workgroup partials in
// LDS, one write-through store per workgroup and value, an arrival counter, and the LAST workgroup to arrive adds the
// partials in index order -- every step has a fixed order, so the totals are bitwise repeatable (global float atomics made
// the last bits run-dependent in round 1), and same-address atomics (~12 ns each on gfx950) are down to one per workgroup.
// Hand-off: sc1 (write-through) stores, vmcnt(0), agent-scope counter; the reader uses sc1 loads (MI355X_MICROARCH.md,
// inter-workgroup visibility).  rb.part holds NV x gridDim.x doubles.
// Returns true in the LAST workgroup (the one that formed the totals), false in every other one -- for a tail that may only run when the
// whole grid is through (k_residual<..., TRIAL>: the loop's second decision).
template <int NV>
__device__ inline bool grid_sum(double (&v)[NV], double* const (&dst)[NV], const RedBuf& rb) {
    __shared__ double s_part[NV][16];
    __shared__ int s_last;
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
        __hip_atomic_store(rb.part + (size_t)threadIdx.x * gridDim.x + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(rb.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (old == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return false;
    // the last workgroup: thread t adds the partials t, t + T, t + 2 T, ... (a fixed assignment, whatever the arrival order was),
    // then the same shuffle / LDS tree as above.  One thread adding all of them took ~0.12 us per workgroup of the grid.
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double t = 0.0;
        for (unsigned b = threadIdx.x; b < gridDim.x; b += blockDim.x)
            t += __hip_atomic_load(rb.part + (size_t)k * gridDim.x + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = wave_sum(t);
        __syncthreads();  // s_part of the first stage has been consumed
        if (lane == 0) s_part[k][wave] = t;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double t = 0.0;
        for (int w = 0; w < nw; ++w) t += s_part[threadIdx.x][w];
        *dst[threadIdx.x] = t;
    }
    if (threadIdx.x == 0) __hip_atomic_store(rb.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

__device__ inline void atomic_max_pos(double* addr, double v) {  // v >= 0; a maximum does not depend on the order
    atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}


// Which slices a wave processes.  Slices are sorted by track length (short first), so a plain grid-stride walk gives the
// waves with the highest indices the longest tracks in every round.  Here a wave alternates between the short end and
// the long end of the list: slice fi from the front, then slice n - 1 - fi from the back, with fi = wave, wave + G,
// wave + 2 G, ...  Every wave gets about the same number of observations, and neighbouring waves still read neighbouring
// slices.  body(g) is called with a wave-uniform slice index.
// rev: the same pairs in the opposite order in time (from the middle of the list outwards).  Consecutive lane = point kernels of an
// iteration alternate the direction, so that a kernel starts on the part of the ELL stream the one before it read last (what of its
// 200 MB is still in the 256 MB memory-side cache) instead of on the part that has been out of it longest.
template <class F>
__device__ inline void for_each_slice(int n_slices, int waves_per_block, F&& body, const int rev = 0) {
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * waves_per_block + (threadIdx.x >> 6)));
    const int G = gridDim.x * waves_per_block;
    const int half = (n_slices + 1) >> 1, rest = n_slices - half;
    for (int i = w; i < half; i += G) {
        const int fi = rev ? half - 1 - i : i;
        body(fi);
        if (fi < rest) body(n_slices - 1 - fi);
    }
}

// Lanes per point.  With lane = point a problem of 100 k points is 1 563 waves for 1 024 SIMDs, each with a serial chain of
// (track length) dependent evaluations: the small shapes ran at well under one wave per SIMD.  A slice (64 points) can therefore
// be processed by 2^sh "units" (waves): unit `sub` of slice gu takes the points 64 gu + sub (64 >> sh) .. and gives every point
// 2^sh neighbouring lanes; lane g of a point takes the slots g, g + 2^sh, ...  Per-point sums are combined with xor-shuffles
// (slice_point_sum) and per-point outputs written by the lane with g == 0.  sh = 0 is the plain lane = point walk.
struct SliceUnit {
    int gu, base, len, nt, q, g, pos, step, sh;
    __device__ inline SliceUnit(const ObsArgs& a, int u, int lane) {
        sh = a.sh;
        gu = u >> sh;
        const int sub = u & ((1 << sh) - 1);
        base = a.slice_base[gu];
        len = (a.slice_base[gu + 1] - base) >> 6;     // slots of the slice
        nt = (len + (1 << sh) - 1) >> sh;              // iterations of a lane
        g = lane & ((1 << sh) - 1);
        const int ql = sub * (64 >> sh) + (lane >> sh);  // point within the slice
        q = gu * 64 + ql;
        pos = base + 64 * g + ql;                      // ELL position of slot g
        step = 64 << sh;
    }
    __device__ inline int slot(int t) const { return (t << sh) + g; }
};
__device__ inline double slice_point_sum(double v, int sh) {
    for (int m = 1; m < (1 << sh); m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// ------------------------------------------------------------------------------------------------ camera constants
__global__ void k_cam_consts(int model, int M, int n_p, int c_p, const double* __restrict__ x,
                             const double* __restrict__ cam_static, double* __restrict__ camc, const int* gate) {
    SATBA_GATE(gate);
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= M) return;
    double full[11];
    for (int i = 0; i < c_p; ++i) full[i] = cam_static[(size_t)c * c_p + i];
    for (int i = 0; i < n_p; ++i) full[i] = x[(size_t)c * n_p + i];
    cam_constants(model, full, camc + (size_t)c * CAMC);
}

// trial point of the cameras, x_new = x + (c0 v0 + c1 v1) / scale_inv on their n_p entries, and the camera constants at x_new
// (the same launch clears the exchange header, hdr_len doubles at xb, which the residual kernel behind it writes into)
__global__ void k_trial_cams(int model, int M, int n_p, int c_p, const double* __restrict__ x, const double* __restrict__ v0,
                             const double* __restrict__ v1, const double* __restrict__ scale_inv, double c0, double c1,
                             const double* __restrict__ cam_static, double* __restrict__ x_new, double* __restrict__ camc_new,
                             double* __restrict__ xb, int hdr_len, const double* __restrict__ coef, const int* gate) {
    SATBA_GATE(gate);
    if (coef) { c0 = coef[0]; c1 = coef[1]; }
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < hdr_len) xb[c] = 0.0;
    if (c >= M) return;
    double full[11];
    for (int i = 0; i < c_p; ++i) full[i] = cam_static[(size_t)c * c_p + i];
    for (int i = 0; i < n_p; ++i) {
        const size_t k = (size_t)c * n_p + i;
        const double v = x[k] + (c0 * v0[k] + c1 * v1[k]) / scale_inv[k];
        x_new[k] = v;
        full[i] = v;
    }
    cam_constants(model, full, camc_new + (size_t)c * CAMC);
}

// ------------------------------------------------------------------------------------------------ point permutation
// caller order <-> internal order of the point part of a variable vector (cameras are copied): dir 0: out[internal] = in[caller]
__global__ void k_permute_vec(int n_c, int N, int dim, const int* __restrict__ perm, const double* __restrict__ in,
                              double* __restrict__ out, int dir) {
    const long long total = (long long)n_c + (long long)N * dim;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        if (i < n_c) { out[i] = in[i]; continue; }
        const long long j = i - n_c;
        const int q = (int)(j / dim), k = (int)(j % dim);
        const long long ext = (long long)n_c + (long long)perm[q] * dim + k;
        if (dir == 0) out[i] = in[ext];
        else out[ext] = in[i];
    }
}
// residual pairs from ELL positions to the caller's observation order
__global__ void k_gather_obs(long long K, const int* __restrict__ obs_pos, const double2* __restrict__ f, double2* __restrict__ out) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < K; o += (long long)gridDim.x * blockDim.x) out[o] = f[obs_pos[o]];
}

__global__ void k_gather_obs1(long long K, const int* __restrict__ obs_pos, const double* __restrict__ v, double* __restrict__ out) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < K; o += (long long)gridDim.x * blockDim.x) out[o] = v[obs_pos[o]];
}

// ------------------------------------------------------------------------------------------------ K1 residuals
// ba_core.fun (ref:bundle_adjust/ba_core.py:157-183).  *cost = 0.5 * sum rho.  f (ELL order) may be null (cost only).
// UNITW: every weight is 1 and the loss is linear (the weight array is not read: 8 of 28 streamed bytes per observation)
#ifndef SATBA_RES_THREADS
#define SATBA_RES_THREADS 512
#endif
constexpr int RES_THREADS = SATBA_RES_THREADS;
// TRIAL: the kernel forms the trial point itself -- x_new = x + (c0 v0 + c1 v1) / scale_inv for the points (the camera entries and
// the camera constants at x_new come from k_trial_cams, launched in front) -- and returns |step|^2 and |x|^2 beside the cost: a
// separate vector kernel moved 120 MB for this at 1 M points (27 us) and cost a launch.  a.x is not read then.
struct TrialArgs {
    const double* __restrict__ x;          // current variables
    const double* __restrict__ v0;         // two direction vectors in scaled variables ...
    const double* __restrict__ v1;
    const double* __restrict__ scale_inv;
    double* __restrict__ x_new;            // the trial point (camera entries already written)
    double c0, c1, lead;                   // ... and their coefficients; weight of the camera entries in the sums (rank 0 only)
    double* ss;                            // |step|^2 and |x|^2 (scaled variables are not involved: plain sums of squares)
    double* xx;
    double* cost2;                         // second copy of this shard's cost (bound of the fixed-point camera sums at x_new)
    const double* coef;                    // device-resident loop: c0, c1 are read from here (two doubles) instead of the arguments
    // device-resident loop on one rank (round 6): the loop's second decision (k_lm_decide2: radius update, accept / reject, termination,
    // report to the host) rides in this launch -- thread 0 of the LAST workgroup, on the totals it has just formed; a launch that is
    // gated off takes it in workgroup 0.  One dependent launch less per tick (4.7 - 5.9 us; 4 % of a 10 x 5 k x 30 k iteration).
    struct LmDev* lm_st = nullptr;
    struct LmSummary* lm_sum = nullptr;
};
__device__ void lm_decide2_body(struct LmDev* gst, double cost_new, double step_sq, double x_sq, struct LmSummary* sum);  // satba_lmdev.h
template <int MODEL, int NP, bool CL, bool RL, bool UNITW = false, bool TRIAL = false>
__global__ __launch_bounds__(RES_THREADS) void k_residual(ObsArgs a, double2* __restrict__ f, RedBuf rb, double* __restrict__ cost, TrialArgs t) {
    if (a.gate != nullptr && *a.gate == 0) {  // (SATBA_GATE)
        if constexpr (TRIAL) {
            if (t.lm_st && blockIdx.x == 0 && threadIdx.x == 0) lm_decide2_body(t.lm_st, 0.0, 0.0, 0.0, t.lm_sum);  // (no trial: the values are not read)
        }
        return;
    }
    if constexpr (TRIAL) { if (t.coef) { t.c0 = t.coef[0]; t.c1 = t.coef[1]; } }
    extern __shared__ __attribute__((aligned(16))) double s_dyn_res[];
    CamTables<CL, RL> T;
    T.stage(a, s_dyn_res, RES_THREADS);
    const int lane = threadIdx.x & 63;
    constexpr int WAVES = RES_THREADS / 64;
    double acc = 0.0, ss = 0.0, xx = 0.0;
    if (TRIAL && blockIdx.x == 0) {  // the camera entries of the two sums (x_new of the cameras is k_trial_cams' work)
        for (int i = threadIdx.x; i < a.n_c; i += RES_THREADS) {
            const double step = (t.c0 * t.v0[i] + t.c1 * t.v1[i]) / t.scale_inv[i], xi = t.x[i];
            ss += t.lead * step * step; xx += t.lead * xi * xi;
        }
    }
    for_each_slice(a.n_slices << a.sh, WAVES, [&](const int u) {
        const SliceUnit su(a, u, lane);
        const int q = su.q;
        const bool has = q < a.N;
        const int cnt = has ? a.pt_cnt[q] : 0;
        double X = 0.0, Y = 0.0, Z = 0.0;
        if (has) {
            const size_t ip = (size_t)a.n_c + 3 * (size_t)q;
            if constexpr (TRIAL) {
                double xn[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double step = (t.c0 * t.v0[ip + k] + t.c1 * t.v1[ip + k]) / t.scale_inv[ip + k], xi = t.x[ip + k];
                    xn[k] = xi + step;
                    if (su.g == 0) { t.x_new[ip + k] = xn[k]; ss += step * step; xx += xi * xi; }
                }
                X = xn[0]; Y = xn[1]; Z = xn[2];
            } else {
                const double* px = a.x + ip; X = px[0]; Y = px[1]; Z = px[2];
            }
        }
        int pos = su.pos;
        // software pipeline: the records of the next two slots are in flight during the arithmetic of the current one
        ObsRec r[SATBA_PF + 1];
#pragma unroll
        for (int j = 0; j < SATBA_PF; ++j) r[j].template load<UNITW>(a, pos + su.step * j, su.slot(j) < cnt);
        for (int tt = 0; tt < su.nt; ++tt, pos += su.step) {
            const int k = su.slot(tt);
            r[SATBA_PF].load<UNITW>(a, pos + su.step * SATBA_PF, su.slot(tt + SATBA_PF) < cnt);
            __builtin_amdgcn_sched_barrier(0);
            if (k < cnt) {
                ObsEval<MODEL, NP, false, !UNITW, false, UNITW> e;
                e.eval(a, r[0].cam, 1.0, T.cc(r[0].cam), T.tab(r[0].cam), r[0].ob, r[0].w, X, Y, Z);
                if (f) f[pos] = make_double2(e.ftrue[0], e.ftrue[1]);
                acc += e.rho;
            }
#pragma unroll
            for (int j = 0; j < SATBA_PF; ++j) r[j] = r[j + 1];
        }
    }, a.rev);
    if constexpr (TRIAL) {
        double v[4] = {0.5 * acc, ss, xx, 0.5 * acc};
        double* const dst[4] = {cost, t.ss, t.xx, t.cost2};
        const bool last = grid_sum<4>(v, dst, rb);
        if (t.lm_st && last) {
            __syncthreads();  // the totals are in place (threads 0 .. 3 of this workgroup stored them)
            if (threadIdx.x == 0)
                lm_decide2_body(t.lm_st, __hip_atomic_load(cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(t.ss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                __hip_atomic_load(t.xx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), t.lm_sum);
        }
    } else {
        double v[1] = {0.5 * acc};
        double* const dst[1] = {cost};
        grid_sum<1>(v, dst, rb);
    }
}

// ------------------------------------------------------------------------------------------------ K2 linearize
// residual + analytic Jacobian -> normal-equation blocks (replaces scipy's finite differences,
// scipy:optimize/_numdiff.py:628-705, and compute_grad, scipy:optimize/_lsq/common.py:590-595):
//   f[pos]          true residual pair (ELL order), if f != null: only k_cam_sums reads it, i.e. runs without the LDS camera
//                   table (SATBA_DETERMINISTIC, very many cameras); the default path does not write it     16 B / obs
//   V[q] (6), g_p   per-point blocks, summed in the lane's registers     72 B / point written
//   part[block][M][2 NP]   per-workgroup camera partials (diag U_c, then g_c), accumulated in 64-BIT FIXED POINT with
//                   ds_add_u64 in an LDS table (CAMSUMS): integer addition is associative, so the sums do not depend on the
//                   order in which the lanes, waves and workgroups arrive -- runs repeat bit for bit (float atomics did not:
//                   round 2 needed a separate fixed-order pass for that).  A term t becomes the integer r = round(t 2^e)
//                   through bits(ldexp(t, e) + 1.5 * 2^52) = bits(1.5 * 2^52) + r for |r| < 2^51; the n copies of the
//                   constant are taken off by k_lin_finish (n = observations of the camera), all arithmetic modulo 2^64 --
//                   only the FINAL sum has to fit.  The exponents come from k_lin_scales: a_i for the Jacobian column i, b for
//                   the residual, so that the diag(U_c) terms scale by 2 a_i and the g_c terms by a_i + b (NP + 1 scalars in
//                   SGPRs), chosen from a bound on the Jacobian at the corners of the points' bounding box and the cost at x
//                   such that |r| stays below 2^Q, Q = min(50, 62 - log2(most observations of a camera)).  Every r is checked
//                   against that range (two 32-bit operations on the high word) and a violation raises a flag: the caller then
//                   forms the sums with k_cam_sums.  The off-diagonal entries of U_c are only needed inside S and come out of
//                   k_schur_diag's registers
//   *hdr_cost = cost;  *hdr_gpmax = max |g_p|
// Affine cameras with R+T corrected, unit weights, linear loss: d(col,row)/dT = [[fx, skew], [0, fy]] for every
// observation, so the two translation entries of diag(U_c) are n_obs(c) * fx^2 and n_obs(c) * (skew^2 + fy^2).  The
// kernel skips those two LDS atomics (8 instead of 10 per observation) and k_lin_finish fills the entries in.
__host__ __device__ constexpr bool lin_const_t(int model, int np, bool robust, bool unit) {
    return model == AFFINE && np == 5 && !robust && unit;
}

// Experiment switch (make OUT=... EXTRA=-DSATBA_ABLATE_CAM_ATOMICS, profiles/r6_linearize_floor.txt): the floor of k_linearize without
// its LDS atomics.  The fixed-point conversion, the range check and the address arithmetic stay -- every term is folded into a
// register that reaches memory at the end --, only the ds_add_u64 go.  The camera sums are wrong in such a build: timing only.
#ifdef SATBA_ABLATE_CAM_ATOMICS
#define SATBA_CAM_ADD(ptr, val) (abl ^= (val) + (unsigned long long)(unsigned)row)
#else
#define SATBA_CAM_ADD(ptr, val) atomicAdd((ptr), (val))
#endif

template <bool BIG>
struct LinCfg {
#ifndef SATBA_LIN_THREADS
#define SATBA_LIN_THREADS 1024
#endif
    static constexpr int THREADS = BIG ? 512 : SATBA_LIN_THREADS;  // the generic robust variants and the RPC chain need > 128 VGPRs
    static constexpr int WAVES = THREADS / 64;
};

// LDS layout (dynamic): camera accumulators [M][CUS] (if CAMSUMS) | camera constants [M][CAMC] (if CL) | RPC tables [M][RPCS] (if RL)
// SOFT (with ROBUST): soft_l1 specialised at compile time
// UNITW (linear loss, every weight 1): the weight array is not read, the Jacobians carry no masks -- the fixed-camera mask
// is applied when the workgroup's camera table is flushed, the fixed-point mask when a point's sums are stored
// CAMSUMS = false: the camera sums are formed by k_cam_sums instead (deterministic runs, camera tables beyond the LDS)
template <int MODEL, int NP, bool ROBUST, bool CL, bool RL, bool SOFT, bool UNITW, bool CAMSUMS>
__global__ __launch_bounds__(LinCfg<(ROBUST && !SOFT) || MODEL == RPC>::THREADS) void k_linearize(
    ObsArgs a, double2* __restrict__ f, double* __restrict__ V, double* __restrict__ gp, double* __restrict__ part, RedBuf rb,
    double* __restrict__ hdr_cost, double* __restrict__ hdr_gpmax) {
    SATBA_GATE(a.gate);
    constexpr int CUS = cam_sum_stride(NP);
    using Cfg = LinCfg<(ROBUST && !SOFT) || MODEL == RPC>;
    constexpr int THREADS = Cfg::THREADS, WAVES = Cfg::WAVES;
    extern __shared__ __attribute__((aligned(16))) double s_lin[];
    unsigned long long* s_acc = reinterpret_cast<unsigned long long*>(s_lin);
    // With few cameras the 64 lanes of an atomic hit the same few addresses and serialise (10 cameras: 58 of the kernel's 75 us
    // at 10 x 5 k x 30 k): the table is replicated 2^rep_shift times, a lane adds to replica (lane mod replicas), the flush
    // adds the replicas up.
    const int n_rows = a.M << a.rep_shift;
    unsigned long long bad = 0ull;  // wave mask of the lanes that saw a term outside the fixed-point range (SALU: s_or_b64)
    [[maybe_unused]] unsigned long long abl = 0ull;  // (SATBA_ABLATE_CAM_ATOMICS)
    int fea[NP], feb = 0, fc1 = 0, fc2 = 0;
    if constexpr (CAMSUMS) {
        for (int i = threadIdx.x; i < n_rows * CUS; i += THREADS) s_acc[i] = 0ull;
        // wave-uniform scalars, read once (left to the compiler they were re-fetched with vector loads in every iteration)
#pragma unroll
        for (int i = 0; i < NP; ++i) fea[i] = __builtin_amdgcn_readfirstlane(a.fxe[i]);
        feb = __builtin_amdgcn_readfirstlane(a.fxe[NP]);
        fc1 = __builtin_amdgcn_readfirstlane(a.fxe[NP + 1]);
        fc2 = __builtin_amdgcn_readfirstlane(a.fxe[NP + 2]);
    }
    CamTables<CL, RL> T;
    T.stage(a, s_lin + (CAMSUMS ? cam_sum_bytes(NP, n_rows) / 8 : 0), THREADS);
    if constexpr (CAMSUMS && !CL && !RL) __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool const_t = lin_const_t(MODEL, NP, ROBUST, a.unit != 0);

    double cost = 0.0, gmax = 0.0, s_gh = 0.0, s_xs = 0.0;
    const int prep_first = a.prep_first_dev ? *a.prep_first_dev : a.prep_first;
    for_each_slice(a.n_slices << a.sh, WAVES, [&](const int u) {
        const SliceUnit su(a, u, lane);
        const int q = su.q;
        const bool has = q < a.N;
        const int cnt = has ? a.pt_cnt[q] : 0;
        double X = 0.0, Y = 0.0, Z = 0.0, mp = 0.0;
        if (has) {
            const double* px = a.x + a.n_c + 3 * (size_t)q;
            X = px[0]; Y = px[1]; Z = px[2];
            mp = (a.perm[q] >= a.n_pts_fix) ? 1.0 : 0.0;
        }
        double v[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int pos = su.pos;
        const int io0 = has ? a.ipt_ofs[q] : 0;
        // software pipeline: the records of the next two slots are in flight during the arithmetic and the LDS atomics of this one
        // (one slot for the robust variants: with two the soft_l1 kernel spilled 13 registers at its 128-VGPR limit)
        constexpr int PF = ROBUST ? 1 : SATBA_PF;
        ObsRec r[PF + 1];
#pragma unroll
        for (int j = 0; j < PF; ++j) r[j].template load<UNITW>(a, pos + su.step * j, su.slot(j) < cnt);
        for (int tt = 0; tt < su.nt; ++tt, pos += su.step) {
            const int k = su.slot(tt);
            r[PF].template load<UNITW>(a, pos + su.step * PF, su.slot(tt + PF) < cnt);
            __builtin_amdgcn_sched_barrier(0);
            if (k < cnt) {
                const int cam = r[0].cam;
                ObsEval<MODEL, NP, true, ROBUST, SOFT, UNITW> e;
                e.eval(a, cam, mp, T.cc(cam), T.tab(cam), r[0].ob, r[0].w, X, Y, Z);
                if constexpr (MODEL == RPC) e.store_jac(a, io0 + k);
                // (this strided 16-byte store is 68 us of the weighted / robust kernel's 250 at 200 x 1M x 10M -- measured by leaving it
                // out; non-temporal: 439 us; whole 32-byte sectors from two buffered slots: 243 us.  Its place -- inside the point's
                // merged record -- is what the Schur kernels' gathers need, section 3 of DESIGN.md)
                if (a.sc) a.sc[io0 + k] = make_double2(e.sw[0], e.sw[1]);
                if (f) f[pos] = make_double2(e.ftrue[0], e.ftrue[1]);  // only the camera-major pass reads it (k_cam_sums)
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
                if constexpr (CAMSUMS) {
                    // camera block: fixed-point LDS atomics (ds_add_u64) into this workgroup's table
                    const int row = (cam << a.rep_shift) | (lane & ((1 << a.rep_shift) - 1));
                    unsigned long long* acc = s_acc + (size_t)row * CUS;
                    // range check: ONE comparison per observation on the unsigned maximum of the shifted high words of its terms (a
                    // compare + ballot per term were 26 of the kernel's ~205 instructions per observation)
                    unsigned umax = 0u;
#pragma unroll
                    for (int i = 0; i < NP; ++i)
                        if (!(const_t && i >= 3)) {
                            const double y = ldexp(e.Jc[0][i] * e.Jc[0][i] + e.Jc[1][i] * e.Jc[1][i], 2 * fea[i]) + FX_MAGIC;
                            umax = max(umax, (unsigned)(__double2hiint(y) + fc1));
                            SATBA_CAM_ADD(acc + i, (unsigned long long)__double_as_longlong(y));
                        }
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        // g_c in two 64-bit limbs: the integer part of the scaled term, and its exact remainder (|rem| <= 1/2) times
                        // 2^40.  The gradient is what the minimiser is defined by: its cancellation at the solution happens exactly, in
                        // integers, and k_lin_finish rounds once, relative to the SUM.  One limb of 46 .. 50 bits left the tight runs
                        // 1e-5 from the reference along flat directions, where float64 sums in a fixed order reach 1e-7
                        const double ts = ldexp(e.Jc[0][i] * e.fs[0] + e.Jc[1][i] * e.fs[1], fea[i] + feb);
                        const double y = ts + FX_MAGIC;
                        umax = max(umax, (unsigned)(__double2hiint(y) + fc1));
                        const double y2 = ldexp(ts - (y - FX_MAGIC), FX_LO_SHIFT) + FX_MAGIC;
                        SATBA_CAM_ADD(acc + NP + i, (unsigned long long)__double_as_longlong(y));
                        SATBA_CAM_ADD(acc + 2 * NP + i, (unsigned long long)__double_as_longlong(y2));
                    }
                    bad |= __ballot(umax >= (unsigned)fc2);
                }
            }
#pragma unroll
            for (int j = 0; j < PF; ++j) r[j] = r[j + 1];
        }
        if (su.sh) {  // several lanes per point: their sums are combined, the lane of slot 0 stores
#pragma unroll
            for (int k = 0; k < 9; ++k) v[k] = slice_point_sum(v[k], su.sh);
        }
        if (has && su.g == 0) {
            if constexpr (UNITW) {  // fixed points: their blocks are masked here
#pragma unroll
                for (int k = 0; k < 9; ++k) v[k] *= mp;
            }
            double2* vo = reinterpret_cast<double2*>(V + 6 * (size_t)q);  // 48-byte rows of an array aligned to 256 bytes
            vo[0] = make_double2(v[0], v[1]); vo[1] = make_double2(v[2], v[3]); vo[2] = make_double2(v[4], v[5]);
            double* go = gp + 3 * (size_t)q;
            go[0] = v[6]; go[1] = v[7]; go[2] = v[8];
            gmax = fmax(gmax, fmax(fabs(v[6]), fmax(fabs(v[7]), fabs(v[8]))));
            if (a.prep_scale) {  // scipy:optimize/_lsq/common.py:598-610 for this point's three variables (k_prepare_vec's arithmetic)
                const size_t ib = (size_t)a.n_c + 3 * (size_t)q;
                const double dg[3] = {v[0], v[3], v[5]}, xv[3] = {X, Y, Z};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double si = sqrt(dg[k]);
                    if (prep_first) si = (si == 0.0) ? 1.0 : si;
                    else si = fmax(si, a.prep_scale[ib + k]);
                    a.prep_scale[ib + k] = si;
                    const double h = v[6 + k] / si;
                    a.prep_gh[ib + k] = h;
                    a.prep_ghs[ib + k] = h / si;
                    s_gh += h * h;
                    const double xs = xv[k] * si;
                    s_xs += xs * xs;
                }
            }
        }
    }, a.rev);
    // per-workgroup epilogue
    gmax = wave_max(gmax);
    if (lane == 0 && gmax > 0.0) atomic_max_pos(hdr_gpmax, gmax);
    if constexpr (CAMSUMS) {
        if (bad != 0ull && lane == 0) atomicOr(a.fx_flag, 1);
#ifdef SATBA_ABLATE_CAM_ATOMICS
        if (abl == 0x5a5a5a5a5a5a5a5aull) atomicOr(a.fx_flag, 4);
#endif
        __syncthreads();
        unsigned long long* out = reinterpret_cast<unsigned long long*>(part) + (size_t)blockIdx.x * a.M * 3 * NP;
        for (int i = threadIdx.x; i < a.M * 3 * NP; i += THREADS) {
            const int cam = i / (3 * NP), k = i % (3 * NP);
            unsigned long long t = 0ull;
            for (int r = 0; r < (1 << a.rep_shift); ++r) t += s_acc[(size_t)((cam << a.rep_shift) | r) * CUS + k];
            out[i] = t;  // fixed cameras (unmasked on the unit-weight path) are zeroed by k_lin_finish
        }
    }
    // cost, and the point sums of the fused prepare part (zero when not fused) in the linearize header
    double cv[3] = {0.5 * cost, s_gh, s_xs};
    double* const dst[3] = {hdr_cost, hdr_cost + SATBA_HDR_PREP_GH, hdr_cost + SATBA_HDR_PREP_XS};
    grid_sum<3>(cv, dst, rb);
}

// sum the per-workgroup camera partials (diag U_c | g_c per camera; 64-bit fixed point, so any order gives the same bits), take
// off the n_obs(camera) copies of the conversion constant, scale back and expand to the exchange payload: U (M x NP x NP,
// diagonal only, rest zero), g_c (M x NP).  64 outputs per workgroup, 16 waves each summing a strided subset of the
// workgroups.  hdr_flag (the linearize header's FX slot) receives 1 when a term of this shard left its range.
__global__ __launch_bounds__(1024) void k_lin_finish(int M, int NP, int nblocks, const double* __restrict__ part, double* __restrict__ U,
                                                     double* __restrict__ gc, const int* __restrict__ cam_ofs,
                                                     const double* __restrict__ camc, int n_cam_fix, int const_t,
                                                     const double* __restrict__ fx, const int* __restrict__ fxe,
                                                     const int* __restrict__ fx_flag, double* __restrict__ hdr_flag, const int* gate) {
    SATBA_GATE(gate);
    const int W = 3 * NP;  // diag U_c | g_c high limbs | g_c low limbs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + lane;  // output (camera, k), k < 2 NP
    const int cam = idx / (2 * NP), k = idx % (2 * NP);
    const unsigned long long* pu = reinterpret_cast<const unsigned long long*>(part);
    __shared__ unsigned long long s_sum[2][16][64];
    unsigned long long s = 0ull, s2 = 0ull;
    if (idx < M * 2 * NP)
        for (int b = wave; b < nblocks; b += 16) {
            const unsigned long long* row = pu + ((size_t)b * M + cam) * W;
            s += row[k];
            if (k >= NP) s2 += row[k + NP];
        }
    s_sum[0][wave][lane] = s;
    s_sum[1][wave][lane] = s2;
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0 && *fx_flag) *hdr_flag = 1.0;
    if (wave != 0 || idx >= M * 2 * NP) return;
    s = 0ull; s2 = 0ull;
#pragma unroll
    for (int w = 0; w < 16; ++w) { s += s_sum[0][w][lane]; s2 += s_sum[1][w][lane]; }
    const unsigned long long n_obs = (unsigned long long)(cam_ofs[cam + 1] - cam_ofs[cam]);
    double v = (double)(long long)(s - n_obs * FX_MAGIC_BITS);
    if (k >= NP) v += ldexp((double)(long long)(s2 - n_obs * FX_MAGIC_BITS), -FX_LO_SHIFT);  // one rounding, relative to the sum
    v *= fx[k];
    if (cam < n_cam_fix) v = 0.0;
    if (k >= NP) {
        gc[cam * NP + (k - NP)] = v;
        return;
    }
    if (const_t && k >= 3) {  // lin_const_t: translation entries of diag(U_c) in closed form
        const double* cc = camc + (size_t)cam * CAMC;
        const double fxx = cc[CAMX + 2], fy = cc[CAMX + 3], sk = cc[CAMX + 4];
        const double cnt = cam >= n_cam_fix ? (double)(cam_ofs[cam + 1] - cam_ofs[cam]) : 0.0;
        v = cnt * (k == 3 ? fxx * fxx : sk * sk + fy * fy);
    }
    U[(size_t)cam * NP * NP + k * NP + k] = v;
}

// Scales of the fixed-point camera sums of k_linearize, one workgroup, launched in front of it (the same launch clears the
// exchange header and the linearize payload, n_clear doubles at `clear`).  Slot k < NP is the diag(U_c) entry k (terms
// (s0 Jc0k)^2 + (s1 Jc1k)^2 >= 0), slot NP + k the g_c entry (terms s0 Jc0k fs0 + s1 Jc1k fs1).  Bounds:
//   JB_k = 2 w_max max |Jc[.][k]| over the cameras and the 8 corners of the (inflated) bounding box of the points -- exact for the
//          affine model (Jc is linear in X), representative for the other two; row scales are <= w_max (rho' + 2 rho'' z <= 1);
//   FB   = bound of |rho'(f^2) f|: sqrt(2 cost(x)) for the linear loss, f_scale for the robust ones;
//   |U term| <= 2 JB_k^2, |g term| <= 2 JB_k FB.
// Range of a converted term: |r| < 2^Q with Q = min(50, 62 - ceil(log2 n_max)) (conversion range; the final sum of n_max terms
// stays inside 63 bits).  Exponents: JB_k 2^a_k < 2^h and FB 2^b < 2^h with h = (Q - 1) / 2 (integer division), so that the U
// terms, scaled by 2^(2 a_k), and the g terms, scaled by 2^(a_k + b), stay below 2^Q.
// fxe: a_0 .. a_{NP-1} | b | c1 | c2  -- the range check of k_linearize is (hi32(y) + c1) <u c2, i.e. r >> 32 in [-L, L), L = 2^(Q-32);
// fx:  [2 NP] inverse scales 2^-(2 a_k), 2^-(a_k + b).  The bounds need not be rigorous: every term is checked.
// (the body: 256 threads of one workgroup; also run by a workgroup of k_lm_accept_scales, satba_lmdev.h)
template <int MODEL, int NP>
__device__ __forceinline__ void lin_scales_body(int M, const double* __restrict__ camc, const double* __restrict__ rpc,
                                                const double* __restrict__ bbox, double w_max, int loss, double f_scale,
                                                const double* __restrict__ cost, double n_max, double shrink,
                                                double* __restrict__ fx, int* __restrict__ fxe, int* __restrict__ fx_flag,
                                                double* __restrict__ clear, int n_clear) {
    __shared__ unsigned long long s_max[NP];
    __shared__ int s_a[NP + 1];
    for (int i = threadIdx.x; i < n_clear; i += 256) clear[i] = 0.0;
    if (threadIdx.x < NP) s_max[threadIdx.x] = 0ull;
    if (threadIdx.x == 0) *fx_flag = 0;
    __syncthreads();
    double c[3], h[3], m[NP];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        c[k] = 0.5 * (bbox[k] + bbox[3 + k]);
        h[k] = (bbox[3 + k] - bbox[k]) + 100.0;  // twice the half extent + 100 m: the points move during the solve
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) m[i] = 0.0;
    // one thread per (camera, corner) (round 5: a thread per camera walked its eight corners one after the other -- eight RPC chains in a
    // row were 17 us of a 0.67 ms iteration at 50 cameras; a maximum does not depend on the order)
    for (int idx = threadIdx.x; idx < 8 * M; idx += 256) {
        const int cam = idx >> 3, corner = idx & 7;
        const double X = c[0] + ((corner & 1) ? h[0] : -h[0]), Y = c[1] + ((corner & 2) ? h[1] : -h[1]), Z = c[2] + ((corner & 4) ? h[2] : -h[2]);
        double u, v, Jc[2][NP], Jp[2][3];
        project<MODEL, NP, true>(camc + (size_t)cam * CAMC, MODEL == RPC ? rpc + (size_t)cam * 90 : nullptr, X, Y, Z, false, u, v, Jc, Jp);
#pragma unroll
        for (int i = 0; i < NP; ++i) m[i] = fmax(m[i], fmax(fabs(Jc[0][i]), fabs(Jc[1][i])));
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const double t = wave_max(m[i]);  // maxima of non-negative values (fmax drops NaNs): their bit patterns order like integers
        if ((threadIdx.x & 63) == 0 && t > 0.0) atomicMax(&s_max[i], (unsigned long long)__double_as_longlong(t));
    }
    __syncthreads();
    int lg_n = 0;
    while (lg_n < 40 && ldexp(1.0, lg_n) < n_max) ++lg_n;  // ceil(log2 n_max)
    const int Q = max(34, min(50, 62 - lg_n)), hq = (Q - 1) / 2;
    if (threadIdx.x <= NP) {
        const int i = threadIdx.x;
        double bound = i < NP ? shrink * 2.0 * w_max * __longlong_as_double((long long)s_max[i < NP ? i : 0])
                              : (loss == 0 ? 1.0001 * sqrt(2.0 * *cost) : f_scale);
        int x = 0, a;
        if (!(bound < 1e300)) a = -900;          // non-finite cost or Jacobian: the solve stops on the cost; every term becomes 0
        else if (!(bound > 1e-300)) a = 900;     // nothing to add in this column
        else { (void)frexp(bound, &x); a = max(-900, min(900, hq - x)); }  // bound < 2^x
        s_a[i] = a;
        fxe[i] = a;
    }
    __syncthreads();
    if (threadIdx.x < 2 * NP) {
        const int k = threadIdx.x, i = k % NP;
        fx[k] = ldexp(1.0, -(k < NP ? 2 * s_a[i] : s_a[i] + s_a[NP]));
    }
    if (threadIdx.x == 0) {
        const int L = 1 << (Q - 32);
        fxe[NP + 1] = (int)((unsigned)L - 0x43380000u);
        fxe[NP + 2] = 2 * L;
    }
}
template <int MODEL, int NP>
__global__ __launch_bounds__(256) void k_lin_scales(int M, const double* __restrict__ camc, const double* __restrict__ rpc,
                                                    const double* __restrict__ bbox, double w_max, int loss, double f_scale,
                                                    const double* __restrict__ cost, double n_max, double shrink,
                                                    double* __restrict__ fx, int* __restrict__ fxe, int* __restrict__ fx_flag,
                                                    double* __restrict__ clear, int n_clear, const int* gate) {
    SATBA_GATE(gate);
    lin_scales_body<MODEL, NP>(M, camc, rpc, bbox, w_max, loss, f_scale, cost, n_max, shrink, fx, fxe, fx_flag, clear, n_clear);
}

// bounding box of the points of x (lo xyz | hi xyz), one workgroup; no points: zeros
__global__ __launch_bounds__(1024) void k_bbox(int N, const double* __restrict__ pts, double* __restrict__ bbox) {
    __shared__ double s_lo[16][3], s_hi[16][3];
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int q = threadIdx.x; q < N; q += 1024) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double v = pts[3 * (size_t)q + k];
            lo[k] = fmin(lo[k], v); hi[k] = fmax(hi[k], v);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double a = -wave_max(-lo[k]), b = wave_max(hi[k]);
        if (lane == 0) { s_lo[wave][k] = a; s_hi[wave][k] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double a = 1e300, b = -1e300;
        for (int w = 0; w < 16; ++w) { a = fmin(a, s_lo[w][threadIdx.x]); b = fmax(b, s_hi[w][threadIdx.x]); }
        if (N == 0) { a = 0.0; b = 0.0; }
        bbox[threadIdx.x] = a;
        bbox[3 + threadIdx.x] = b;
    }
}

// ------------------------------------------------------------------------------------------------ camera-major camera sums
struct CamMajor {
    const int* __restrict__ cam_ofs;  // M + 1
    const int* __restrict__ pt;       // K: internal point of every entry
    const int* __restrict__ pos;      // K: its ELL position (residuals)
    const int* __restrict__ io;       // K: its io index (row scales, stored RPC blocks)
};
#ifndef SATBA_LINC_THREADS
#define SATBA_LINC_THREADS 256
#endif
constexpr int LINC_THREADS = SATBA_LINC_THREADS;

// error-free addition (Knuth): hi + t = s + e exactly; the error is collected in lo
__device__ __forceinline__ void two_sum_acc(double& hi, double& lo, double t) {
    const double s = hi + t, bb = s - hi;
    lo += (hi - (s - bb)) + (t - bb);
    hi = s;
}
__device__ __forceinline__ void wave_sum2(double& hi, double& lo) {  // lane 0 ends up with the wave's (hi, lo)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const double oh = __shfl_down(hi, d), ol = __shfl_down(lo, d);
        two_sum_acc(hi, lo, oh);
        lo += ol;
    }
}

// Full U_c = J_c^T J_c (upper triangle) and g_c = J_c^T f of the linearisation k_linearize has just stored (f, row scales,
// RPC blocks), by a camera-major pass: grid (M, chunks), every thread accumulates cam_acc_len(NP) sums of its camera in
// registers over a strided slice of the camera's list; fixed summation order.  Used (a) for the parity tests' view of the
// full blocks, (b) SATBA_DETERMINISTIC runs, (c) camera counts whose accumulator table does not fit the LDS.
// g_c is summed with compensation (round 5): every addition is error free (two_sum_acc), the errors travel in a second word through
// the lanes, the waves and the chunk partials (part: cam_acc_len(NP) + NP doubles per (camera, chunk)) and are added once at the
// end.  The gradient cancels to ~1e-9 of its terms at the solution; plain float64 sums left the tight affine_C2_R / soft_l1 solve
// 1.3e-6 of |f| short of the minimiser the fixed-point sums of k_linearize (exact) reach to 6e-9 (the CPU oracle, plain sums, stops
// at the same 1.3e-6).
template <int MODEL, int NP>
__global__ __launch_bounds__(LINC_THREADS) void k_cam_sums(ObsArgs a, CamMajor c, const double2* __restrict__ f, double* __restrict__ part) {
    SATBA_GATE(a.gate);
    constexpr int CU = cam_acc_len(NP);
    const int cam = blockIdx.x, chunk = blockIdx.y, n_chunks = gridDim.y;
    const int b = c.cam_ofs[cam], e = c.cam_ofs[cam + 1];
    const long long len = e - b;
    const int lo = b + (int)(len * chunk / n_chunks), hi = b + (int)(len * (chunk + 1) / n_chunks);
    const double* cc = a.camc + (size_t)cam * CAMC;
    const double mc = (cam >= a.n_cam_fix) ? 1.0 : 0.0;
    double acc[CU], glo[NP];
#pragma unroll
    for (int k = 0; k < CU; ++k) acc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < NP; ++k) glo[k] = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += LINC_THREADS) {
        const int pos = c.pos[i], q = c.pt[i], io = c.io[i];
        double2 ff;
        double Jc[2][NP], Jp[2][3];
        if constexpr (MODEL == RPC) {  // the residuals and D' k_linearize stored (row scales included)
            ff = f[pos];
            ObsEval<MODEL, NP, true> e2;
            const double* px = a.x + a.n_c + 3 * (size_t)q;
            e2.jac(a, io, cam, 1.0, cc, nullptr, px[0], px[1], px[2]);
#pragma unroll
            for (int k = 0; k < NP; ++k) { Jc[0][k] = e2.Jc[0][k]; Jc[1][k] = e2.Jc[1][k]; }
        } else {  // the projection is evaluated for the Jacobian anyway: the residual comes with it (no 16 B / observation store)
            const double* px = a.x + a.n_c + 3 * (size_t)q;
            double u, v;
            project<MODEL, NP, true>(cc, nullptr, px[0], px[1], px[2], false, u, v, Jc, Jp);
            const double2 ob = a.e_obs[pos];
            const double w = a.unit ? 1.0 : a.e_w[pos];
            ff = make_double2(w * (u - ob.x), w * (v - ob.y));
            double s0 = mc, s1 = mc;
            if (a.sc) { const double2 t = a.sc[io]; s0 *= t.x; s1 *= t.y; }
#pragma unroll
            for (int k = 0; k < NP; ++k) { Jc[0][k] *= s0; Jc[1][k] *= s1; }
        }
        double fs0 = ff.x, fs1 = ff.y;
        if (a.loss != 0) {
            double r, js;
            robust(a.loss, a.f_scale, ff.x, r, fs0, js, MODEL != RPC);
            robust(a.loss, a.f_scale, ff.y, r, fs1, js, MODEL != RPC);
        }
        int k = 0;
#pragma unroll
        for (int r = 0; r < NP; ++r)
#pragma unroll
            for (int s = r; s < NP; ++s) acc[k++] += Jc[0][r] * Jc[0][s] + Jc[1][r] * Jc[1][s];
#pragma unroll
        for (int r = 0; r < NP; ++r, ++k) two_sum_acc(acc[k], glo[r], Jc[0][r] * fs0 + Jc[1][r] * fs1);
    }
    constexpr int NTRI = NP * (NP + 1) / 2;
    __shared__ double s_red[LINC_THREADS / 64][CU + NP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NTRI; ++k) {
        const double t = wave_sum(acc[k]);
        if (lane == 0) s_red[wave][k] = t;
    }
#pragma unroll
    for (int r = 0; r < NP; ++r) {
        wave_sum2(acc[NTRI + r], glo[r]);
        if (lane == 0) { s_red[wave][NTRI + r] = acc[NTRI + r]; s_red[wave][CU + r] = glo[r]; }
    }
    __syncthreads();
    if (threadIdx.x < CU) {
        double* out = part + ((size_t)cam * n_chunks + chunk) * (CU + NP);
        double t = 0.0, tl = 0.0;
        if (threadIdx.x < NTRI) {
            for (int wv = 0; wv < LINC_THREADS / 64; ++wv) t += s_red[wv][threadIdx.x];
        } else {
            for (int wv = 0; wv < LINC_THREADS / 64; ++wv) { two_sum_acc(t, tl, s_red[wv][threadIdx.x]); tl += s_red[wv][CU + threadIdx.x - NTRI]; }
            out[CU + threadIdx.x - NTRI] = tl;
        }
        out[threadIdx.x] = t;
    }
}

// chunk partials -> U (M x NP x NP, both triangles), g_c (M x NP)
__global__ void k_cam_sums_finish(int M, int NP, int n_chunks, const double* __restrict__ part, double* __restrict__ U, double* __restrict__ gc,
                                  const int* gate) {
    SATBA_GATE(gate);
    const int CU = cam_acc_len(NP);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * CU) return;
    const int cam = idx / CU, k = idx % CU;
    const int ntri = NP * (NP + 1) / 2, W = CU + NP;  // k_cam_sums: cam_acc_len sums, then the error words of the g_c entries
    if (k >= ntri) {
        double t = 0.0, tl = 0.0;
        for (int ch = 0; ch < n_chunks; ++ch) {
            const double* row = part + ((size_t)cam * n_chunks + ch) * W;
            two_sum_acc(t, tl, row[k]);
            tl += row[CU + k - ntri];
        }
        gc[cam * NP + (k - ntri)] = t + tl;
        return;
    }
    double t = 0.0;
    for (int ch = 0; ch < n_chunks; ++ch) t += part[((size_t)cam * n_chunks + ch) * W + k];
    int r = 0, rem = k;
    while (rem >= NP - r) { rem -= NP - r; ++r; }
    const int s = r + rem;
    U[(size_t)cam * NP * NP + r * NP + s] = t;
    U[(size_t)cam * NP * NP + s * NP + r] = t;
}

// ------------------------------------------------------------------------------------------------ prepare
// x_scale="jac" (scipy:optimize/_lsq/common.py:598-610): scale_inv = column norms of J = sqrt(diag(J^T J)),
// zeros -> 1 on the first evaluation, running maximum afterwards; g_h = g / scale_inv.
// hdr[1] = |g_h|^2, hdr[3] = |x * scale_inv|^2 (camera part only if lead), hdr[4] = lead * |g_c|_inf
__global__ __launch_bounds__(256) void k_prepare_vec(int n, int n_c, int NP, int first, double lead,
                                                     const double* __restrict__ U, const double* __restrict__ gc_red,
                                                     const double* __restrict__ V, const double* __restrict__ x,
                                                     double* __restrict__ g, double* __restrict__ scale_inv,
                                                     double* __restrict__ gh, double* __restrict__ ghs, RedBuf rb, double* __restrict__ hdr,
                                                     const double* __restrict__ keep, const int* __restrict__ first_dev, const int* gate) {
    // ghs = g_h / scale_inv: the unscaled direction of g_h, input of the Jacobian-vector product that follows
    SATBA_GATE(gate);
    if (first_dev) first = *first_dev;  // device-resident loop: the first linearisation of a solve initialises the scaling
    if (keep[SATBA_K_FX] != 0.0) return;
    // n == n_c: the point entries were done by k_linearize (ObsArgs::prep_*); their sums wait in keep[2], keep[4] (k_prepare_stash)
    const bool fused = n == n_c;  // the camera sums of this linearisation overflowed their fixed-point range: the caller
                                          // repeats it with k_cam_sums; the running maximum of scale_inv must not see the garbage
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
    m_gc = wave_max(m_gc);
    if ((threadIdx.x & 63) == 0 && m_gc > 0.0) atomic_max_pos(hdr + 4, lead * m_gc);
    if (fused && blockIdx.x == 0 && threadIdx.x == 0) { s_gh += keep[2]; s_xs += keep[4]; }
    double v[2] = {s_gh, s_xs};
    double* const dst[2] = {hdr + 1, hdr + 3};
    grid_sum<2>(v, dst, rb);
}

// ------------------------------------------------------------------------------------------------ Jacobian-vector products
// per-camera constants of the affine form of k_jvp / k_backsub: B (6) | b (2) | A (6) -- seven 16-byte slots, read as seven ds_read_b128
// (DirRow).  Round 6: with an odd stride of 15 doubles the fourteen reads were ds_read2_b64 pairs (128 B / clk, banks modulo 32).
constexpr int JVP_ROW = 14;
static_assert(JVP_ROW % 2 == 0 && (JVP_ROW / 2) % 2 == 1, "direction-table rows: an odd number of 16-byte slots");
struct DirRow {
    double v[14];
    __device__ inline void load(const double* __restrict__ tab, int cam) {
        const double2* r = static_cast<const double2*>(__builtin_assume_aligned(tab + (size_t)cam * JVP_ROW, 16));
#pragma unroll
        for (int m = 0; m < 7; ++m) { const double2 t = r[m]; v[2 * m] = t.x; v[2 * m + 1] = t.y; }
    }
};
#ifndef SATBA_JVP_THREADS
#define SATBA_JVP_THREADS 512
#endif
constexpr int JVP_THREADS = SATBA_JVP_THREADS;
// slots of a point whose camera index (and row scales) are in flight ahead of the one being worked on, in the table forms of k_jvp and
// k_backsub.  Not what bounds them: depths 2 / 4 / 8 give 49.9 / 50.0 / 49.9 us (k_jvp) and 55.4 / 54.9 / 55.6 us (k_backsub) at
// 200 x 1M x 10M (round 5).  Their time is the LDS: the table row of a random camera per observation -- 64 lanes on ~50 different rows.
// Round 5 read it as fourteen 8-byte values, which the compiler paired into ds_read2_b64 (8 LDS cycles each, banks modulo 32, ~3 passes
// on random rows: ~197 cycles per wave and iteration, which is what 50 us were); round 6 reads seven aligned ds_read_b128 (4 cycles
// each, banks modulo 64): k_jvp 50 -> 37 us, k_backsub 56 -> 46 us in the loop (profiles/r6_b128_tables.txt)
#ifndef SATBA_CAM_PF
#define SATBA_CAM_PF 2
#endif

// affine cameras: J_c v_c = B_c X + b_c with B_c = sum_i v_ci D_ci, b_c = K-columns . v_cT, and J_p = A_c.  Every workgroup
// derives the 14 constants of each camera once (three evaluations of the projector's Jacobian at the unit vectors) into an
// LDS table of 112-byte rows; an observation then costs seven 16-byte LDS reads and 14-18 multiply-adds instead of the Jacobian
// evaluation.  vc: the camera part of the vector (unscaled variables), n_c doubles.
// vs (or null): vc is in scaled variables, the direction is vc / vs (as the generic path forms it: times the reciprocal)
template <int NP>
__device__ inline void affine_dir_table(const ObsArgs& a, const double* __restrict__ vc, double* tab, int nthreads, const double* __restrict__ vs = nullptr) {
    for (int c = threadIdx.x; c < a.M; c += nthreads) {
        const double* cc = a.camc + (size_t)c * CAMC;
        double u, v, Jc[2][NP], Jp[2][3], B[2][3], b[2] = {0.0, 0.0};
        const double mc = (c >= a.n_cam_fix) ? 1.0 : 0.0;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            project<AFFINE, NP, true>(cc, nullptr, m == 0 ? 1.0 : 0.0, m == 1 ? 1.0 : 0.0, m == 2 ? 1.0 : 0.0, false, u, v, Jc, Jp);
            B[0][m] = 0.0; B[1][m] = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double vi = vs ? vc[c * NP + i] * (1.0 / vs[c * NP + i]) : vc[c * NP + i];
                B[0][m] += Jc[0][i] * vi; B[1][m] += Jc[1][i] * vi;
            }
        }
#pragma unroll
        for (int i = 3; i < NP; ++i) {
            const double vi = vs ? vc[c * NP + i] * (1.0 / vs[c * NP + i]) : vc[c * NP + i];
            b[0] += Jc[0][i] * vi; b[1] += Jc[1][i] * vi;
        }
        double* row = tab + (size_t)c * JVP_ROW;
#pragma unroll
        for (int m = 0; m < 3; ++m) { row[m] = mc * B[0][m]; row[3 + m] = mc * B[1][m]; row[8 + m] = Jp[0][m]; row[11 + m] = Jp[1][m]; }
        row[6] = mc * b[0]; row[7] = mc * b[1];
    }
    __syncthreads();
}

// The same tables in global memory (a.dir_tab), one launch of one workgroup in front of k_jvp / k_backsub<..., DG = true>: camera counts
// whose tables do not fit the LDS (two tables of 120 bytes per camera: from ~640 cameras on).  v2 (or null): a second direction
template <int NP>
__global__ __launch_bounds__(1024) void k_affine_dir_tab(ObsArgs a, const double* __restrict__ v1, const double* __restrict__ v2, const double* __restrict__ vs) {
    SATBA_GATE(a.gate);
    affine_dir_table<NP>(a, v1, a.dir_tab, 1024, vs);
    if (v2) affine_dir_table<NP>(a, v2, a.dir_tab + (size_t)a.M * JVP_ROW, 1024, vs);
}

// For NV vectors given in scaled variables (v = q / scale_inv): sums of (J v_a) . (J v_b) over the observations.
// NV = 1: out[0] = |J v1|^2.   NV = 2: out[0] = |J v1|^2, out[1] = (J v1).(J v2), out[2] = |J v2|^2.
// PRE (NV = 1): q1 is already divided by scale_inv (k_prepare_vec)
// DG (affine cameras): the direction tables are read from a.dir_tab (k_affine_dir_tab) instead of being built in the LDS
template <int MODEL, int NP, int NV, bool CL, bool RL, bool PRE, bool DG = false>
__global__ __launch_bounds__(JVP_THREADS) void k_jvp(ObsArgs a, const double* __restrict__ q1, const double* __restrict__ q2,
                                                     const double* __restrict__ scale_inv, RedBuf rb, double* __restrict__ out) {
    SATBA_GATE(a.gate);
    extern __shared__ __attribute__((aligned(16))) double s_dyn_jvp[];
    const int lane = threadIdx.x & 63;
    constexpr int WAVES = JVP_THREADS / 64;
    double s11 = 0.0, s12 = 0.0, s22 = 0.0;
    if constexpr (MODEL == AFFINE && PRE && NV == 1) {
        const double* tab = DG ? a.dir_tab : s_dyn_jvp;  // M x JVP_ROW
        if constexpr (!DG) affine_dir_table<NP>(a, q1, s_dyn_jvp, JVP_THREADS);
        for_each_slice(a.n_slices << a.sh, WAVES, [&](const int u) {
            const SliceUnit su(a, u, lane);
            const int q = su.q;
            const bool has = q < a.N;
            const int cnt = has ? a.pt_cnt[q] : 0;
            double X = 0.0, Y = 0.0, Z = 0.0, v0 = 0.0, v1 = 0.0, v2 = 0.0;
            if (has) {
                const size_t ip = (size_t)a.n_c + 3 * (size_t)q;
                X = a.x[ip]; Y = a.x[ip + 1]; Z = a.x[ip + 2];
                const double mp = (a.perm[q] >= a.n_pts_fix) ? 1.0 : 0.0;
                v0 = mp * q1[ip]; v1 = mp * q1[ip + 1]; v2 = mp * q1[ip + 2];
            }
            int pos = su.pos;
            const int io0 = has ? a.ipt_ofs[q] : 0;
            // cameras (and row scales) of the next two slots in flight
            constexpr int D = SATBA_CAM_PF;
            int cq[D];
            double2 sq[D];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int kd = su.slot(d);
                cq[d] = 0; sq[d] = make_double2(1.0, 1.0);
                if (kd < cnt) { cq[d] = a.e_cam[pos + d * su.step]; if (a.sc) sq[d] = a.sc[io0 + kd]; }
            }
            for (int tt = 0; tt < su.nt; ++tt, pos += su.step) {
                const int k = su.slot(tt), kn = su.slot(tt + D);
                int cn = 0;
                double2 sn = make_double2(1.0, 1.0);
                if (kn < cnt) { cn = a.e_cam[pos + D * su.step]; if (a.sc) sn = a.sc[io0 + kn]; }
                __builtin_amdgcn_sched_barrier(0);
                const int c0 = cq[0];
                const double2 s0 = sq[0];
                if (k < cnt) {
                    DirRow R;
                    R.load(tab, c0);
                    const double* row = R.v;
                    const double j0 = s0.x * (row[0] * X + row[1] * Y + row[2] * Z + row[6] + row[8] * v0 + row[9] * v1 + row[10] * v2);
                    const double j1 = s0.y * (row[3] * X + row[4] * Y + row[5] * Z + row[7] + row[11] * v0 + row[12] * v1 + row[13] * v2);
                    s11 += j0 * j0 + j1 * j1;
                }
#pragma unroll
                for (int d = 0; d + 1 < D; ++d) { cq[d] = cq[d + 1]; sq[d] = sq[d + 1]; }
                cq[D - 1] = cn; sq[D - 1] = sn;
            }
        }, a.rev);
    } else if constexpr (MODEL == AFFINE && !PRE && NV == 2) {
        // the two directions of the explicit-products pattern (g_h and gn_h parallel: 30 % of the soft_l1 iterations at 200 cameras):
        // two direction tables instead of a Jacobian evaluation per observation (208 -> ~110 us at 10 M observations)
        const double* tab1 = DG ? a.dir_tab : s_dyn_jvp;
        const double* tab2 = tab1 + (size_t)a.M * JVP_ROW;
        if constexpr (!DG) {
            affine_dir_table<NP>(a, q1, s_dyn_jvp, JVP_THREADS, scale_inv);
            affine_dir_table<NP>(a, q2, s_dyn_jvp + (size_t)a.M * JVP_ROW, JVP_THREADS, scale_inv);
        }
        for_each_slice(a.n_slices << a.sh, WAVES, [&](const int u) {
            const SliceUnit su(a, u, lane);
            const int q = su.q;
            const bool has = q < a.N;
            const int cnt = has ? a.pt_cnt[q] : 0;
            double X = 0.0, Y = 0.0, Z = 0.0, p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0};
            if (has) {
                const size_t ip = (size_t)a.n_c + 3 * (size_t)q;
                X = a.x[ip]; Y = a.x[ip + 1]; Z = a.x[ip + 2];
                const double mp = (a.perm[q] >= a.n_pts_fix) ? 1.0 : 0.0;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double si = 1.0 / scale_inv[ip + j];
                    p1[j] = mp * (q1[ip + j] * si);
                    p2[j] = mp * (q2[ip + j] * si);
                }
            }
            int pos = su.pos;
            const int io0 = has ? a.ipt_ofs[q] : 0;
            constexpr int D = SATBA_CAM_PF;
            int cq[D];
            double2 sq[D];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int kd = su.slot(d);
                cq[d] = 0; sq[d] = make_double2(1.0, 1.0);
                if (kd < cnt) { cq[d] = a.e_cam[pos + d * su.step]; if (a.sc) sq[d] = a.sc[io0 + kd]; }
            }
            for (int tt = 0; tt < su.nt; ++tt, pos += su.step) {
                const int k = su.slot(tt), kn = su.slot(tt + D);
                int cn = 0;
                double2 sn = make_double2(1.0, 1.0);
                if (kn < cnt) { cn = a.e_cam[pos + D * su.step]; if (a.sc) sn = a.sc[io0 + kn]; }
                __builtin_amdgcn_sched_barrier(0);
                const int c0 = cq[0];
                const double2 s0 = sq[0];
                if (k < cnt) {
                    DirRow R1, R2;
                    R1.load(tab1, c0);
                    R2.load(tab2, c0);
                    const double *r1 = R1.v, *r2 = R2.v;
                    const double a0 = r1[8], a1 = r1[9], a2 = r1[10], a3 = r1[11], a4 = r1[12], a5 = r1[13];  // J_p (the same in both tables)
                    const double u0 = s0.x * (r1[0] * X + r1[1] * Y + r1[2] * Z + r1[6] + a0 * p1[0] + a1 * p1[1] + a2 * p1[2]);
                    const double u1 = s0.y * (r1[3] * X + r1[4] * Y + r1[5] * Z + r1[7] + a3 * p1[0] + a4 * p1[1] + a5 * p1[2]);
                    const double w0 = s0.x * (r2[0] * X + r2[1] * Y + r2[2] * Z + r2[6] + a0 * p2[0] + a1 * p2[1] + a2 * p2[2]);
                    const double w1 = s0.y * (r2[3] * X + r2[4] * Y + r2[5] * Z + r2[7] + a3 * p2[0] + a4 * p2[1] + a5 * p2[2]);
                    s11 += u0 * u0 + u1 * u1;
                    s12 += u0 * w0 + u1 * w1;
                    s22 += w0 * w0 + w1 * w1;
                }
#pragma unroll
                for (int d = 0; d + 1 < D; ++d) { cq[d] = cq[d + 1]; sq[d] = sq[d + 1]; }
                cq[D - 1] = cn; sq[D - 1] = sn;
            }
        }, a.rev);
    } else {
        CamTables<CL, RL> T;
        T.stage(a, s_dyn_jvp, JVP_THREADS);
        for_each_slice(a.n_slices << a.sh, WAVES, [&](const int u) {
            const SliceUnit su(a, u, lane);
            const int q = su.q;
            const bool has = q < a.N;
            const int cnt = has ? a.pt_cnt[q] : 0;
            double X = 0.0, Y = 0.0, Z = 0.0, mp = 0.0, p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0};
            if (has) {
                const size_t ip = (size_t)a.n_c + 3 * (size_t)q;
                X = a.x[ip]; Y = a.x[ip + 1]; Z = a.x[ip + 2];
                mp = (a.perm[q] >= a.n_pts_fix) ? 1.0 : 0.0;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if constexpr (PRE) p1[j] = q1[ip + j];
                    else {
                        const double si = 1.0 / scale_inv[ip + j];
                        p1[j] = q1[ip + j] * si;
                        if (NV == 2) p2[j] = q2[ip + j] * si;
                    }
                }
            }
            int pos = su.pos;
            const int io0 = has ? a.ipt_ofs[q] : 0;
            for (int tt = 0; tt < su.nt; ++tt, pos += su.step) {
                const int k = su.slot(tt);
                if (k < cnt) {
                    const int cam = a.e_cam[pos];
                    ObsEval<MODEL, NP, true> e;
                    e.jac(a, io0 + k, cam, mp, T.cc(cam), T.tab(cam), X, Y, Z);
                    const size_t ic = (size_t)cam * NP;
                    double j1[2] = {0, 0}, j2[2] = {0, 0};
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        double v1;
                        if constexpr (PRE) v1 = q1[ic + i];
                        else {
                            const double si = 1.0 / scale_inv[ic + i];
                            v1 = q1[ic + i] * si;
                            if (NV == 2) { const double v2 = q2[ic + i] * si; j2[0] += e.Jc[0][i] * v2; j2[1] += e.Jc[1][i] * v2; }
                        }
                        j1[0] += e.Jc[0][i] * v1; j1[1] += e.Jc[1][i] * v1;
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        j1[0] += e.Jp[0][j] * p1[j]; j1[1] += e.Jp[1][j] * p1[j];
                        if (NV == 2) { j2[0] += e.Jp[0][j] * p2[j]; j2[1] += e.Jp[1][j] * p2[j]; }
                    }
                    s11 += j1[0] * j1[0] + j1[1] * j1[1];
                    if (NV == 2) {
                        s12 += j1[0] * j2[0] + j1[1] * j2[1];
                        s22 += j2[0] * j2[0] + j2[1] * j2[1];
                    }
                }
            }
        }, a.rev);
    }
    if constexpr (NV == 2) {
        double v[3] = {s11, s12, s22};
        double* const dst[3] = {out, out + 1, out + 2};
        grid_sum<3>(v, dst, rb);
    } else {
        double v[1] = {s11};
        double* const dst[1] = {out};
        grid_sum<1>(v, dst, rb);
    }
}

// ------------------------------------------------------------------------------------------------ reduced system helpers
// Reduced system in scaled variables: S <- diag(scale) S diag(scale), rhs <- scale * rhs (scale = 1 / scale_inv).
// x_scale="jac" makes the scaled matrix unit-diagonal up to the damping, which keeps the dense factorisation
// well conditioned although the raw camera blocks span ~12 orders of magnitude (angles vs translations).
// (the same launch clears the dense solver's status word and flags, n_clear ints at `clear`: one fill less)
// col_lo, col_hi: only the columns of that range (the overlapped factorisation scales every range of columns as it arrives);
// the right-hand side and the flags go with the range that starts at column 0
__global__ __launch_bounds__(256) void k_scale_system(int n_c, const double* __restrict__ scale_inv, double* __restrict__ S,
                                                      const double* __restrict__ rhs, double* __restrict__ rhs_scaled, int* __restrict__ clear,
                                                      int n_clear, const int* gate, int col_lo, int col_hi) {
    SATBA_GATE(gate);
    const size_t lo = (size_t)col_lo * n_c, nn = (size_t)col_hi * n_c;
    const bool head = col_lo == 0;
    if (head)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n_clear; i += (size_t)gridDim.x * blockDim.x) clear[i] = 0;
    for (size_t i = lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nn + (head ? n_c : 0); i += (size_t)gridDim.x * blockDim.x) {
        if (i < nn) {
            const int r = (int)(i % n_c), c = (int)(i / n_c);
            if (r >= c) S[i] /= scale_inv[r] * scale_inv[c];
        } else {
            rhs_scaled[i - nn] = rhs[i - nn] / scale_inv[i - nn];  // the solve works on a copy: the payload is all-reduced data
        }
    }
}

// dc = dc_h / scale_inv (unscaled camera step for the back-substitution).  The same launch prepares the header of the
// solve phase (nothing touches it between here and k_backsub): zero, Cholesky status in slot 4, the scalars
// kept from the earlier phases of this iteration in slots SATBA_HDR_KEEP.. (rank 0 only: headers are summed over ranks)
// done (or null): the dense solve runs on ANOTHER stream (front_schur_solve, beside the pair kernel) and posts done_epoch there when
// dc_h is complete (k_trsv_back_mw) -- every workgroup waits for the word instead of the stream for an event (13 us from the end of
// the substitution to the start of this kernel, measured), and reads what that stream wrote with agent-scope loads
__global__ void k_unscale(int n_c, const double* __restrict__ scale_inv, const double* __restrict__ dch, double* __restrict__ dc,
                          int hdr_len, double* __restrict__ hdr, const int* __restrict__ fail_flag, double lead,
                          const double* __restrict__ keep, int keep_at, int keep_len, const int* gate, const int* __restrict__ done = nullptr,
                          int done_epoch = 0) {
    SATBA_GATE(gate);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int timed_out = 0;
    if (done) {
        int spins = 0;
        while ((int)(__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - done_epoch) < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1 << 22)) { timed_out = 2 | 4; break; }  // (~1 s: the other stream's kernels bound their own waits; bit 2: a wait between the two streams)
        }
    }
    if (i < n_c) dc[i] = (done ? __hip_atomic_load(dch + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : dch[i]) / scale_inv[i];
    if (i < hdr_len) {
        double v = 0.0;
        // (0: factorised; bit 0 not positive definite, bit 1 a wait timed out, bit 2: it was a wait of the concurrent front -- for the kernel
        // on the other stream)
        if (i == 4) v = lead * (double)((done ? __hip_atomic_load(fail_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *fail_flag) | timed_out);
        if (i >= keep_at && i < keep_at + keep_len) v = lead * keep[i - keep_at];
        hdr[i] = v;
    }
}

// start of the prepare phase, one launch: the (already all-reduced) linearize payload U | g_c is copied out of the
// exchange buffer, keep[0] = cost and keep[1] = max_rank |g_p|_inf are taken from its header, the header is zeroed
__global__ __launch_bounds__(1024) void k_prepare_stash(int nU, int n_c, int world, int hdr_fixed, int hdr_len, double* __restrict__ xb,
                                                        double* __restrict__ U, double* __restrict__ gc, double* __restrict__ keep, const int* gate) {
    SATBA_GATE(gate);
    const double* payload = xb + hdr_len;
    for (int i = threadIdx.x; i < nU + n_c; i += blockDim.x) {
        if (i < nU) U[i] = payload[i];
        else gc[i - nU] = payload[i];
    }
    if (threadIdx.x == 0) {
        double m = 0.0;
        for (int r = 0; r < world; ++r) m = fmax(m, xb[hdr_fixed + r]);
        keep[0] = xb[0];
        keep[1] = m;
        keep[2] = xb[SATBA_HDR_PREP_GH]; keep[4] = xb[SATBA_HDR_PREP_XS];  // (replaced by the totals in the Schur phase)
        keep[SATBA_K_FX] = xb[SATBA_HDR_FX];
    }
    __syncthreads();
    if ((int)threadIdx.x < hdr_len) xb[threadIdx.x] = 0.0;
}

// k_prepare_stash and the camera entries of k_prepare_vec in ONE launch of one workgroup: what the prepare phase is when k_linearize
// has done its point part (ObsArgs::prep_*, the loops of satba_capi.hip) and there are at most 1 024 camera unknowns -- one launch
// less in front of every iteration (4.4 us of 140 at 10 cameras).  Same values as the two kernels; the two sums are formed by this
// workgroup alone (fixed order), the point part's sums (header slots SATBA_HDR_PREP_GH / _XS) added last.
__global__ __launch_bounds__(1024) void k_prepare_cams(int nU, int n_c, int NP, int world, int hdr_fixed, int hdr_len, int first, double lead,
                                                       double* __restrict__ xb, double* __restrict__ U, double* __restrict__ gc,
                                                       double* __restrict__ keep, const double* __restrict__ x, double* __restrict__ g,
                                                       double* __restrict__ scale_inv, double* __restrict__ gh, double* __restrict__ ghs,
                                                       const int* __restrict__ first_dev, const int* gate) {
    SATBA_GATE(gate);
    __shared__ double s_k[3], s_red[3][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* payload = xb + hdr_len;
    for (int i = tid; i < nU + n_c; i += 1024) {
        if (i < nU) U[i] = payload[i];
        else gc[i - nU] = payload[i];
    }
    if (tid == 0) {
        double m = 0.0;
        for (int r = 0; r < world; ++r) m = fmax(m, xb[hdr_fixed + r]);
        keep[0] = xb[0];
        keep[1] = m;
        s_k[0] = xb[SATBA_HDR_PREP_GH]; s_k[1] = xb[SATBA_HDR_PREP_XS]; s_k[2] = xb[SATBA_HDR_FX];
        keep[2] = s_k[0]; keep[4] = s_k[1];  // (replaced by the totals in the Schur phase)
        keep[SATBA_K_FX] = s_k[2];
    }
    __syncthreads();
    if (tid < hdr_len) xb[tid] = 0.0;
    if (s_k[2] != 0.0) return;  // the fixed-point camera sums overflowed: the caller repeats the linearisation (k_prepare_vec)
    if (first_dev) first = *first_dev;
    double s_gh = 0.0, s_xs = 0.0, m_gc = 0.0;
    for (int i = tid; i < n_c; i += 1024) {
        const int cam = i / NP, k = i % NP;
        const double diag = payload[(size_t)cam * NP * NP + k * NP + k], gi = payload[nU + i];
        g[i] = gi;
        m_gc = fmax(m_gc, fabs(gi));
        double si = sqrt(diag);
        if (first) si = (si == 0.0) ? 1.0 : si;
        else si = fmax(si, scale_inv[i]);
        scale_inv[i] = si;
        const double h = gi / si;
        gh[i] = h;
        ghs[i] = h / si;
        s_gh += lead * h * h;
        const double xs = x[i] * si;
        s_xs += lead * xs * xs;
    }
    s_gh = wave_sum(s_gh); s_xs = wave_sum(s_xs); m_gc = wave_max(m_gc);
    if (lane == 0) { s_red[0][wave] = s_gh; s_red[1][wave] = s_xs; s_red[2][wave] = m_gc; }
    __syncthreads();  // (also: the header is cleared)
    if (tid == 0) {
        double a = 0.0, b = 0.0, m = 0.0;
        for (int w = 0; w < 16; ++w) { a += s_red[0][w]; b += s_red[1][w]; m = fmax(m, s_red[2][w]); }
        xb[1] = a + s_k[0];
        xb[3] = b + s_k[1];
        xb[4] = lead * m;
    }
}

// ------------------------------------------------------------------------------------------------ K5 back-substitution
// t_p = sum_obs Jp^T (Jc dc[cam]) per point in the lane's registers, then the point part of the Gauss-Newton step in scaled
// variables, gn_h = scale_inv_p * Vinv (g_p - t), and the Gram matrix of (g_h, gn_h): hdr[1..3] = a, b, c.  The camera part
// gn_h[0 .. n_c) = dc_h is copied by workgroup 0.  (Round 1 needed a staging buffer and a second kernel for the per-point sums.)
#ifndef SATBA_BS_THREADS
#define SATBA_BS_THREADS 512
#endif
constexpr int BS_THREADS = SATBA_BS_THREADS;
template <int MODEL, int NP, bool CL, bool RL, bool DG = false>
__global__ __launch_bounds__(BS_THREADS) void k_backsub(ObsArgs a, const double* __restrict__ dc, const double* __restrict__ dch,
                                                        double lead, const double* __restrict__ Vinv, const double* __restrict__ g,
                                                        const double* __restrict__ scale_inv, const double* __restrict__ gh,
                                                        double* __restrict__ gn, RedBuf rb, double* __restrict__ hdr) {
    SATBA_GATE(a.gate);
    extern __shared__ __attribute__((aligned(16))) double s_dyn_bs[];
    const int lane = threadIdx.x & 63;
    constexpr int WAVES = BS_THREADS / 64;
    double sa = 0.0, sb = 0.0, sc = 0.0;
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < a.n_c; i += BS_THREADS) {
            const double v = dch[i], h = gh[i];
            gn[i] = v;
            sa += lead * h * h; sb += lead * h * v; sc += lead * v * v;
        }
    }
    CamTables<CL, RL> T;
    const double* tab = DG ? a.dir_tab : s_dyn_bs;  // (DG: built by k_affine_dir_tab in front)
    if constexpr (MODEL == AFFINE) { if constexpr (!DG) affine_dir_table<NP>(a, dc, s_dyn_bs, BS_THREADS); }
    else T.stage(a, s_dyn_bs, BS_THREADS);
    for_each_slice(a.n_slices << a.sh, WAVES, [&](const int u) {
        const SliceUnit su(a, u, lane);
        const int q = su.q;
        const bool has = q < a.N;
        const int cnt = has ? a.pt_cnt[q] : 0;
        double X = 0.0, Y = 0.0, Z = 0.0, mp = 0.0;
        if (has) {
            const double* px = a.x + a.n_c + 3 * (size_t)q;
            X = px[0]; Y = px[1]; Z = px[2];
            mp = (a.perm[q] >= a.n_pts_fix) ? 1.0 : 0.0;
        }
        double t[3] = {0.0, 0.0, 0.0};
        int pos = su.pos;
        const int io0 = has ? a.ipt_ofs[q] : 0;
        if constexpr (MODEL == AFFINE) {
            constexpr int D = SATBA_CAM_PF;
            int cq[D];
            double2 sq[D];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int kd = su.slot(d);
                cq[d] = 0; sq[d] = make_double2(1.0, 1.0);
                if (kd < cnt) { cq[d] = a.e_cam[pos + d * su.step]; if (a.sc) sq[d] = a.sc[io0 + kd]; }
            }
            for (int tt = 0; tt < su.nt; ++tt, pos += su.step) {
                const int k = su.slot(tt), kn = su.slot(tt + D);
                int cn = 0;
                double2 sn = make_double2(1.0, 1.0);
                if (kn < cnt) { cn = a.e_cam[pos + D * su.step]; if (a.sc) sn = a.sc[io0 + kn]; }
                __builtin_amdgcn_sched_barrier(0);
                const int cam = cq[0];
                const double2 s2 = sq[0];
#pragma unroll
                for (int d = 0; d + 1 < D; ++d) { cq[d] = cq[d + 1]; sq[d] = sq[d + 1]; }
                cq[D - 1] = cn; sq[D - 1] = sn;
                if (k < cnt) {
                    DirRow R;
                    R.load(tab, cam);
                    const double* row = R.v;
                    // both blocks of an observation carry its row scale
                    const double u0 = mp * s2.x * s2.x * (row[0] * X + row[1] * Y + row[2] * Z + row[6]);
                    const double u1 = mp * s2.y * s2.y * (row[3] * X + row[4] * Y + row[5] * Z + row[7]);
#pragma unroll
                    for (int j = 0; j < 3; ++j) t[j] += row[8 + j] * u0 + row[11 + j] * u1;
                }
            }
        } else {
            for (int tt = 0; tt < su.nt; ++tt, pos += su.step) {
                const int k = su.slot(tt);
                if (k < cnt) {
                    const int cam = a.e_cam[pos];
                    ObsEval<MODEL, NP, true> e;
                    e.jac(a, io0 + k, cam, mp, T.cc(cam), T.tab(cam), X, Y, Z);
                    double u0 = 0.0, u1 = 0.0;
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        const double d = dc[cam * NP + i];  // global gather (L1 hits)
                        u0 += e.Jc[0][i] * d;
                        u1 += e.Jc[1][i] * d;
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j) t[j] += e.Jp[0][j] * u0 + e.Jp[1][j] * u1;
                }
            }
        }
        if (su.sh) {  // several lanes per point: W_p^T dc is the sum of their parts, the lane of slot 0 finishes the point
#pragma unroll
            for (int j = 0; j < 3; ++j) t[j] = slice_point_sum(t[j], su.sh);
        }
        if (has && su.g == 0) {
            const double* vi = Vinv + 6 * (size_t)q;
            const size_t ib = (size_t)a.n_c + 3 * (size_t)q;
            const double r0 = g[ib] - t[0], r1 = g[ib + 1] - t[1], r2 = g[ib + 2] - t[2];
            const double d[3] = {vi[0] * r0 + vi[1] * r1 + vi[2] * r2, vi[1] * r0 + vi[3] * r1 + vi[4] * r2,
                                 vi[2] * r0 + vi[4] * r1 + vi[5] * r2};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double v = d[k] * scale_inv[ib + k], h = gh[ib + k];
                gn[ib + k] = v;
                sa += h * h; sb += h * v; sc += v * v;
            }
        }
    }, a.rev);
    double v[3] = {sa, sb, sc};
    double* const dst[3] = {hdr + 1, hdr + 2, hdr + 3};
    grid_sum<3>(v, dst, rb);
}

// ------------------------------------------------------------------------------------------------ subspace / trial vectors
// q1 = s g_h, w = gn_h - alpha g_h;  hdr[1] = w.w, hdr[2] = w.q1, hdr[6] = g_h.w
__global__ __launch_bounds__(256) void k_subspace_vec(int n, int n_c, double lead, double alpha, double s,
                                                      const double* __restrict__ gh, const double* __restrict__ gn,
                                                      double* __restrict__ q1, double* __restrict__ wv, RedBuf rb,
                                                      double* __restrict__ hdr, const double* __restrict__ args_dev, const int* gate) {
    SATBA_GATE(gate);
    if (args_dev) { alpha = args_dev[0]; s = args_dev[1]; }  // device-resident loop
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
    grid_sum<3>(v, dst, rb);
}

// ------------------------------------------------------------------------------------------------ inspection
// materialised, weighted, row-scaled Jacobian blocks in the CALLER's observation order (parity tests)
template <int MODEL, int NP>
__global__ void k_jacobian(ObsArgs a, const int* __restrict__ pts_ind, const int* __restrict__ rank, const int* __restrict__ obs_pos,
                           double* __restrict__ Jc, double* __restrict__ Jp) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < a.K; o += (long long)gridDim.x * blockDim.x) {
        const int p = pts_ind[o], q = rank[p], pos = obs_pos[o], cam = a.e_cam[pos];
        const double* px = a.x + a.n_c + 3 * (size_t)q;
        ObsEval<MODEL, NP, true> e;
        e.eval(a, cam, (p >= a.n_pts_fix) ? 1.0 : 0.0, a.camc + (size_t)cam * CAMC, (MODEL == RPC) ? a.rpc + (size_t)cam * 90 : nullptr,
               a.e_obs[pos], a.e_w[pos], px[0], px[1], px[2]);
        for (int r = 0; r < 2; ++r) {
            for (int i = 0; i < NP; ++i) Jc[(o * 2 + r) * NP + i] = e.Jc[r][i];
            for (int j = 0; j < 3; ++j) Jp[(o * 2 + r) * 3 + j] = e.Jp[r][j];
        }
    }
}

}  // namespace satba
