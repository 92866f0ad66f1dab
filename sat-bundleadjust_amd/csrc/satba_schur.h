// satba_schur.h -- K3: Schur complement of the point blocks, by camera-pair intersection.  No atomics on the data path,
// everything accumulates in registers.
//
//   S_ij -= sum over points p seen by BOTH cameras i and j of  W_ip Vinv_p W_jp^T,   W = Jc^T Jp  (NP x 3)
//
// One wavefront owns one (camera pair, point-range chunk) item: it streams the pair's static list of shared points
// (satba_layout.h; a coalesced 12-byte stream: point, the ELL positions of its two observations), gathers the packed
// 128-byte point record (X, Vinv) of every hit, evaluates both Jacobians -- both cameras are wave-uniform, their constants
// sit in scalar registers -- and accumulates the NP x NP block in registers; one wave reduce-scatter per item, the chunk
// partials are added by k_schur_finish.  The diagonal blocks (including the full J_c^T J_c) and the right-hand side
// come from a camera-major pass (k_schur_diag) that also accumulates in registers.
// History (round 1): global float64 atomics 72.5 ms, LDS column panels 6.2 ms, visibility-bitmap intersection 1.8-3.0 ms,
// static pair lists 0.90 ms, chunked dispatch + 128-byte records 0.59 ms at 200 x 1M x 10M.
#pragma once
#include "satba_kernels.h"

namespace satba {

// ------------------------------------------------------------------------------------------------ point blocks
// (V_p + lam Dp^2)^-1 per point, symmetric 3x3 stored as xx xy xz yy yz zz
constexpr int PV_STRIDE = 16;  // doubles per packed point record: 12 used, padded to 16 so that a record is exactly one
                               // 128-byte line (96-byte records straddle lines: 1.5 lines per gather; Schur 0.945 -> 0.845 ms)
// PV: packed per-point record X(3) | Vinv(6) | g_p(3) for the gather-heavy Schur kernels
// The copy of Vinv inside PV is multiplied by the point's fixed mask (0 for a frozen point, 1 otherwise): every Schur term
// contains Vinv_p exactly once, so the Schur kernels need no mask of their own (a gather of perm[] per hit otherwise).
// Start of satba_schur_auto: the (already all-reduced) prepare header -> keep[1] = |g|_inf, keep[2..4] = |g_h|^2, |J_h g_h|^2,
// |x_h|^2; trust radius (scipy trf.py:440-442 when Delta <= 0: first iteration) -> keep[6]; damping of the Gauss-Newton system
// from the Cauchy step (scipy trf.py:473-477, common.py:302-322) -> keep[5].  Returns the damping.  Every thread of k_vinv
// evaluates it from the header (a dozen operations; a one-thread kernel in front cost a launch), one thread writes `keep`.
__device__ inline double schur_lambda(const double* __restrict__ hdr, double Delta, double lam_floor, double* __restrict__ keep, bool write) {
    const double gh_sq = hdr[1], jg_sq = hdr[2], xs_sq = hdr[3];
    if (!(Delta > 0.0)) {
        Delta = sqrt(xs_sq);
        if (Delta == 0.0) Delta = 1.0;
    }
    // minimum of a t^2 + b t on [0, ub]
    const double a = 0.5 * jg_sq, b = -gh_sq, ub = Delta / sqrt(gh_sq);
    double best = fmin(0.0, a * ub * ub + b * ub);
    if (a != 0.0) {
        const double ext = -0.5 * b / a;
        if (0.0 < ext && ext < ub) best = fmin(best, a * ext * ext + b * ext);
    }
    double lam = -best / (Delta * Delta);
    if (!(lam >= lam_floor)) lam = lam_floor;  // also catches NaN (zero gradient)
    if (write) {
        keep[1] = fmax(keep[1], hdr[4]);
        keep[2] = gh_sq; keep[3] = jg_sq; keep[4] = xs_sq;
        keep[5] = lam;
        keep[6] = Delta;
    }
    return lam;
}

// hdr_auto (optional): the damping comes from the prepare header (schur_lambda, Delta / lam_floor / keep as there) instead of `lam`
// Stores: a lane's 48-byte Vinv row and 128-byte record are strided over the wave (24 and 64 lines per store instruction: the
// texture path handles one line per cycle); the wave's 64 rows / records are contiguous in memory, so they are transposed through
// LDS (row strides 48 and 144 bytes: the 16 lanes of a ds_write_b128 phase fall into disjoint banks) and written as three / eight
// fully coalesced 16-byte stores per lane.  Workgroups of 256 threads, N need not be a multiple of 64.
#ifndef SATBA_VINV_THREADS
#define SATBA_VINV_THREADS 256
#endif
constexpr int VINV_THREADS = SATBA_VINV_THREADS;
__global__ __launch_bounds__(VINV_THREADS) void k_vinv(int N, double lam, const double* __restrict__ hdr_auto, double Delta, double lam_floor,
                                                       double* __restrict__ keep, const double* __restrict__ V,
                                                       const double* __restrict__ scale_inv_p, double* __restrict__ Vinv,
                                                       const double* __restrict__ xp, const double* __restrict__ gp, double* __restrict__ PV,
                                                       const int* __restrict__ perm, int n_pts_fix, const double* __restrict__ Delta_dev,
                                                       const double* __restrict__ lam_force, const int* gate, const int* __restrict__ w_fix = nullptr) {
    SATBA_GATE(gate);
    if (Delta_dev) Delta = *Delta_dev;  // device-resident loop: the trust radius lives in the loop's state (<= 0: first iteration)
    if (lam_force && *lam_force > 0.0) {  // ... and so does the escalated damping after a failed factorisation (the prepare header is gone then)
        hdr_auto = nullptr;
        lam = *lam_force;
        if (blockIdx.x == 0 && threadIdx.x == 0) keep[5] = lam;
    }
    __shared__ double2 s_t[VINV_THREADS / 64][64 * 9];  // per wave: 64 records x 144 bytes (the Vinv rows use the front of it)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p0 = (blockIdx.x * (VINV_THREADS / 64) + wave) * 64;  // first point of this wave
    if (p0 >= N) return;
    const int p = min(p0 + lane, N - 1);  // lanes past the end repeat the last point (their stores are dropped below)
    if (hdr_auto) lam = schur_lambda(hdr_auto, Delta, lam_floor, keep, p0 + lane == 0);
    const double2* v2 = reinterpret_cast<const double2*>(V + 6 * (size_t)p);
    const double2 va = v2[0], vb = v2[1], vc = v2[2];
    const double* s = scale_inv_p + 3 * (size_t)p;
    const double a = va.x + lam * s[0] * s[0], b = va.y, c = vb.x;
    const double d = vb.y + lam * s[1] * s[1], e = vc.x, f = vc.y + lam * s[2] * s[2];
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double idet = 1.0 / (a * c00 + b * c01 + c * c02);
    const double o0 = c00 * idet, o1 = c01 * idet, o2 = c02 * idet;
    const double o3 = (a * f - c * c) * idet, o4 = (b * c - a * e) * idet, o5 = (a * d - b * b) * idet;
    const double x0 = xp[3 * (size_t)p], x1 = xp[3 * (size_t)p + 1], x2 = xp[3 * (size_t)p + 2];
    const double g0 = gp[3 * (size_t)p], g1 = gp[3 * (size_t)p + 1], g2 = gp[3 * (size_t)p + 2];
    const double mp = (perm[p] >= n_pts_fix) ? 1.0 : 0.0;
    double2* t = s_t[wave];
    const int n_here = min(64, N - p0);  // points of this wave
    // Vinv rows: 3 pieces per point
    t[3 * lane] = make_double2(o0, o1); t[3 * lane + 1] = make_double2(o2, o3); t[3 * lane + 2] = make_double2(o4, o5);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    {
        double2* out = reinterpret_cast<double2*>(Vinv + 6 * (size_t)p0);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int g = 64 * k + lane;
            const double2 v = t[g];
            if (g < 3 * n_here) out[g] = v;
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // records: 8 pieces per point (6 used), LDS row stride 9 pieces
    static_assert(PV_STRIDE == 16, "records are eight 16-byte pieces");
    double2* row = t + 9 * lane;
    row[0] = make_double2(x0, x1); row[1] = make_double2(x2, mp * o0); row[2] = make_double2(mp * o1, mp * o2);
    row[3] = make_double2(mp * o3, mp * o4); row[4] = make_double2(mp * o5, g0); row[5] = make_double2(g1, g2);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (w_fix) {  // merged records (Layout::w_fix): PV is the buffer W, record p's six pieces start at piece w_fix[p] -- 96 contiguous bytes per record
        double2* out = reinterpret_cast<double2*>(PV);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int g = 64 * k + lane, r = g >> 3, q = g & 7;
            const double2 v = t[9 * r + (q < 6 ? q : 0)];
            const int at = w_fix[min(p0 + r, N - 1)];
            if (q < 6 && r < n_here) out[(size_t)at + q] = v;
        }
        return;
    }
    {
        double2* out = reinterpret_cast<double2*>(PV + PV_STRIDE * (size_t)p0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int g = 64 * k + lane, r = g >> 3, q = g & 7;
            const double2 v = t[9 * r + (q < 6 ? q : 0)];
            if (q < 6 && r < n_here) out[g] = v;  // the two pad pieces of a record are never read
        }
    }
}

// one work item of k_schur_pairs, flattened: without it a wave starts with three dependent loads (item -> pair_ij, pair_ofs)
struct SchurItem {
    long long lo, hi;  // entry range of the (pair, chunk) list
    int i, j;          // cameras, i < j; i < 0: padding
    int pair, chunk;
};
struct SchurArgs {
    const double2* __restrict__ PV;          // N x 8 double2: X0 X1 | X2 v00 | v01 v02 | v11 v12 | v22 g0 | g1 g2 | pad
    const long long* __restrict__ pair_ofs;  // n_pairs x (n_chunks + 1): offsets into the lists
    const int* __restrict__ pair_pts;        // points shared by each camera pair, ascending (static per problem)
    const int* __restrict__ pair_pi;         // io indices of the point's observation in camera i / j
    const int* __restrict__ pair_pj;
    const int2* __restrict__ pair_ij;        // pair index -> (i, j), i < j
    double* __restrict__ pair_part;          // n_chunks x n_pairs x NP*NP partial blocks (n_chunks > 1)
    const int2* __restrict__ items;          // work items in dispatch order: (pair, chunk), pair < 0: padding
    const struct SchurItem* __restrict__ desc;  // the same items with everything a wave needs to start (k_schur_item_desc)
    int n_chunks;
    int diag_xcd = 0;                        // k_schur_diag: chunks dealt to the XCDs (see there)
    // weighted / robust runs, affine and perspective cameras (Layout::w_fix): PV points at the merged records W (piece offsets instead
    // of point indices), the pair lists are pair_rec (piece of X0 of the hit's record) and pair_kk (distances of its two scales in
    // front of it, (c - k_i) | (c - k_j) << 16); zero_fix: an all-zero record.  The diagonal blocks and the right-hand side are items
    // of k_schur_pairs then (item.i == item.j: the camera; lo .. hi: its entries in cm_rec / cm_sc; chunk: its slot among the
    // camera's n_dg partials dg_part, which k_schur_finish adds up like k_schur_diag's)
    int wmode = 0, zero_fix = 0, n_dg = 1;
    const int* __restrict__ pair_rec = nullptr;
    const int* __restrict__ pair_kk = nullptr;
    const int* __restrict__ cm_rec = nullptr;
    const int* __restrict__ cm_sc = nullptr;
    double* __restrict__ dg_part = nullptr;
    // ... beside the factorisation (arrive != null): the diagonal items publish their partials and count themselves in dg_cnt[camera] (M
    // ints, zero between launches); the camera's LAST item in dispatch order (slot n_dg - 1: every other one has been started before
    // it) waits for that count, adds the partials up the way k_schur_finish does (schur_diag_total / schur_diag_store: the same
    // arithmetic in the same order), writes the diagonal block and the camera's entries of the right-hand side and counts the camera
    // in `arrive` -- the factorisation expects M - c counts of camera c then (C3Args::arr_extra) and takes the right-hand side tile by tile
    int* dg_cnt = nullptr;
    double dg_lam = 0.0, dg_lead = 1.0;
    const double* dg_lam_dev = nullptr;
    const double* __restrict__ dg_gc = nullptr;
    const double* __restrict__ dg_scale_inv = nullptr;
    double* __restrict__ dg_rhs = nullptr;
    // factorisation running beside the pair kernel (satba_chol3.h, C3Args::arrive): every item publishes its block (write-through
    // stores, drain) and counts itself in the word of its camera row i; word M = arrive_epoch once the kernels in front of this one
    // (diagonal blocks, right-hand side) are complete.  Words SCHUR_ARRIVE_STRIDE ints apart (one per 128-byte line); null: off
    int* arrive = nullptr;
    int arrive_epoch = 0;
    // ... with point-range chunks (weighted / robust runs): every (pair, chunk) item publishes its partial block and counts itself in
    // pair_cnt[pair] (n_pairs ints, zero between launches); the item of the LAST chunk -- the last of its pair in dispatch order: every
    // other one has been started before it -- waits for that count, adds the partials in chunk order (k_schur_finish's arithmetic),
    // publishes the block and counts the pair in `arrive`.  fail: the factorisation's status word (bit 1: a wait timed out)
    int* pair_cnt = nullptr;
    int* fail = nullptr;
};
constexpr int SCHUR_ARRIVE_STRIDE = 32;

// Wave "reduce-scatter": N (power of two) values per lane are summed over the 64 lanes with N - 1 + (6 - log2 N)
// shuffles instead of 6 N: at every step a lane keeps one half of its values and trades the other half with its
// partner.  On return v[0] of the lanes with (lane & (64/N - 1)) == 0 ... holds total number rs_index<N>(lane).
template <int N>
__device__ inline double wave_reduce_scatter(const double (&v)[N], int lane, int mask) {
    if constexpr (N == 1) {
        double t = v[0];
        for (int m = mask; m > 0; m >>= 1) t += __shfl_xor(t, m);
        return t;
    } else {
        constexpr int H = N / 2;
        const bool upper = (lane & mask) != 0;
        double w[H];
#pragma unroll
        for (int k = 0; k < H; ++k) {
            const double keep = upper ? v[k + H] : v[k];
            const double send = upper ? v[k] : v[k + H];
            w[k] = keep + __shfl_xor(send, mask);
        }
        return wave_reduce_scatter<H>(w, lane, mask >> 1);
    }
}
// index (in 0 .. N-1) of the total a lane ends up with
template <int N>
__device__ inline int rs_index(int lane) {
    int idx = 0, mask = 32;
    for (int h = N / 2; h >= 1; h >>= 1, mask >>= 1)
        if (lane & mask) idx += h;
    return idx;
}

__global__ void k_schur_item_desc(long long n_items, const int2* __restrict__ items, const int2* __restrict__ pair_ij,
                                  const long long* __restrict__ pair_ofs, int n_chunks, SchurItem* __restrict__ desc,
                                  const int* __restrict__ dg_ofs = nullptr, int n_dg = 1) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_items) return;
    const int2 it = items[e];
    SchurItem d;
    if (it.x == -1) { d.lo = d.hi = 0; d.i = d.j = -1; d.pair = -1; d.chunk = 0; }
    else if (it.x < -1) {  // diagonal item of camera -2 - it.x, slot it.y among its n_dg partials
        const int cam = -2 - it.x;
        d.lo = dg_ofs[(size_t)cam * n_dg + it.y];
        d.hi = dg_ofs[(size_t)cam * n_dg + it.y + 1];
        d.i = d.j = cam; d.pair = -1; d.chunk = it.y;
    } else {
        const int2 ij = pair_ij[it.x];
        const int c0 = it.y < 0 ? 0 : it.y, c1 = it.y < 0 ? n_chunks : it.y + 1;  // chunk < 0: the whole list of the pair
        d.lo = pair_ofs[(long long)it.x * (n_chunks + 1) + c0];
        d.hi = pair_ofs[(long long)it.x * (n_chunks + 1) + c1];
        d.i = ij.x; d.j = ij.y; d.pair = it.x; d.chunk = c0;
    }
    desc[e] = d;
}

// End of a (pair, chunk) item beside the factorisation (SchurArgs::pair_cnt).
// mine: the item's entry of its partial block; stride: doubles between the chunks' partial blocks
__device__ __forceinline__ void schur_pair_publish(double* mine, size_t stride, int n_chunks, int chunk, int* cnt, int* arrive_row, int* fail,
                                                double* s_out, double total, bool writer) {
    const int lane = threadIdx.x & 63;
    if (writer) __hip_atomic_store(mine, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (chunk + 1 < n_chunks) {
        if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (lane == 0) {  // the last chunk's item: every other item of the pair has been started before this one
        int spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_chunks - 1) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1 << 22)) { atomicOr(fail, 2); break; }
        }
        __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_wave_barrier();
    if (writer) {
        const double* first = mine - (size_t)chunk * stride;
        double t = 0.0;
        for (int ch = 0; ch < n_chunks; ++ch) t += __hip_atomic_load(first + (size_t)ch * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(s_out, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(arrive_row, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The sum of a camera's n_chunks partials of output k (k_schur_diag's, or the diagonal items' of k_schur_pairs), by eight neighbouring
// lanes: lane `sub` adds every eighth partial, the eight sums are combined by an xor tree -- one fixed order, wherever it runs.
// ATOMIC: the partials were published by other workgroups of the SAME launch (agent-scope loads).
template <bool ATOMIC>
__device__ __forceinline__ double schur_diag_total(const double* __restrict__ part, int cam, int n_chunks, int CU, int k, int sub, bool live) {
    double t = 0.0;
    if (live)
        for (int ch = sub; ch < n_chunks; ch += 8) {
            const double* q = part + ((size_t)cam * n_chunks + ch) * CU + k;
            t += ATOMIC ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
        }
    t += __shfl_xor(t, 1);
    t += __shfl_xor(t, 2);
    t += __shfl_xor(t, 4);
    return t;
}
// ... and where it goes: output k < NP (NP + 1) / 2 is entry (r, q >= r) of the diagonal block (+ (lead) lam Dc^2 on the diagonal; both
// triangles of the block are written), the others are the camera's entries of the right-hand side (+ (lead) g_c).  rhs_scaled (or null):
// the system goes out in scaled variables (k_schur_finish)
template <bool ATOMIC>
__device__ __forceinline__ void schur_diag_store(int cam, int k, double t, int NP, int n_c, double lam, double lead, const double* __restrict__ gc,
                                                 const double* __restrict__ scale_inv, double* __restrict__ S, double* __restrict__ rhs,
                                                 double* __restrict__ rhs_scaled) {
    auto st = [](double* q, double v) { if (ATOMIC) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *q = v; };
    const int ntri = NP * (NP + 1) / 2;
    if (k >= ntri) {
        const int col = cam * NP + (k - ntri);
        const double v = lead * gc[col] + t;
        st(rhs + col, v);
        if (rhs_scaled) rhs_scaled[col] = v / scale_inv[col];
        return;
    }
    int r = 0, rem = k;
    while (rem >= NP - r) { rem -= NP - r; ++r; }
    const int q = r + rem;
    double v = 0.0;
    if (q == r) {
        const double si = scale_inv[cam * NP + r];
        v = lead * lam * si * si;
    }
    double out = v + t;
    if (rhs_scaled) out /= scale_inv[cam * NP + q] * scale_inv[cam * NP + r];
    st(S + (size_t)(cam * NP + q) + (size_t)(cam * NP + r) * n_c, out);
    if (q != r) st(S + (size_t)(cam * NP + r) + (size_t)(cam * NP + q) * n_c, out);
}

template <int MODEL, int NP, int THREADS>
__device__ __forceinline__ void schur_diag_walk(const ObsArgs& a, const double2* __restrict__ base, const int* __restrict__ lst_rec,
                                                const int* __restrict__ lst_sc, int rmul, int zero_rec, int cam, int lo, int hi, int tid,
                                                double2* my, double (&acc)[cam_acc_len(NP)]);

// 1-D grid, 4 waves per workgroup, one (pair, chunk) item each, taken from an item table in DISPATCH order that is built
// for the chip's topology (satba_capi.hip: schur_item_table): workgroup b runs on XCD b % 8 (observed placement; only speed
// depends on it), and every XCD works through the pairs (i, j) of ONE camera i and ONE point-range chunk at a time.  All those
// items gather from the records of camera i's points in that chunk (~1.6 MB at 200 x 1M x 10M), each record is needed by ~9
// of them, and the 4 MB L2 of the XCD keeps it: round 1 dispatched chunk-major over all pairs and every XCD streamed the
// whole 32 MB window of the chunk through its L2 (hit rate 34 %, 4.4 GB fetched per launch for 0.13 GB of records).
// UNITW: every weight is 1 and the loss is linear -- raw Jacobians, nothing is fetched per observation.
// Otherwise (weighted / robust): unit-weight Jacobians times the row scales k_linearize stored (a.sc) -- the scales multiply
// the 2 x 2 middle matrix (4 products instead of 32 on the blocks).  RPC: the stored Jacobian blocks are gathered instead
// of being recomputed (they carry scales and masks).
#ifndef SATBA_PAIRS_OCC_U
#define SATBA_PAIRS_OCC_U 3
#endif
#ifndef SATBA_PAIRS_OCC_W
#define SATBA_PAIRS_OCC_W 3  // weighted / robust affine kernel (round 5: with 2 waves per SIMD and no spills the Schur phase is 1.21 instead of 1.09 ms)
#endif
#ifndef SATBA_PAIRS_OCC_MIN_O
#define SATBA_PAIRS_OCC_MIN_O 1  // perspective / RPC kernels: least waves per SIMD the register allocator must leave room for (218 - 252 registers: two
                                 // waves.  Round 5, forced to 3 -- 168 registers, 88 - 340 bytes of scratch in the hit loop: C5 1 419 against 1 699 it/s, P3 1 209 / 2 324)
#endif
#define SATBA_PAIRS_WAVES_ATTR(MODEL, UNITW) __attribute__((amdgpu_waves_per_eu(((MODEL) == AFFINE) ? ((UNITW) ? SATBA_PAIRS_OCC_U : SATBA_PAIRS_OCC_W) : SATBA_PAIRS_OCC_MIN_O, ((MODEL) == AFFINE) ? ((UNITW) ? SATBA_PAIRS_OCC_U : SATBA_PAIRS_OCC_W) : 3)))
// bidx: the workgroup's index among the pair workgroups of the launch; s_coop: 4 x 64 x 7 double2 of LDS -- per wave 64 records x 80 (112: with the
// scales) bytes, or 64 Jacobian rows x 112 (the cooperative gathers are transposed here); s_idx: weighted / robust, the three gather indices of a wave's 64 hits
template <int MODEL, int NP, bool UNITW>
__device__ __forceinline__ void schur_pairs_body(const ObsArgs& a, const SchurArgs& s, double* __restrict__ S, const unsigned bidx, double2* s_coop, unsigned (*s_idx)[3 * 64]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n_pairs = (long long)a.M * (a.M - 1) / 2;
    const SchurItem* dp = s.desc + (bidx * 4u + (unsigned)wave);
    const int i = __builtin_amdgcn_readfirstlane(dp->i);  // wave-uniform by construction: lets the camera constants use scalar loads
    if (i < 0) return;
    const int j = __builtin_amdgcn_readfirstlane(dp->j);
    const int chunk = __builtin_amdgcn_readfirstlane(dp->chunk);
    const long long pair = __builtin_amdgcn_readfirstlane(dp->pair);
    if constexpr (!UNITW && MODEL != RPC) {
        if (i == j) {  // a diagonal item (SchurArgs::wmode): partial diagonal block and right-hand side of camera i over the entries lo .. hi of its list
            constexpr int CU = cam_acc_len(NP), CPAD = CU <= 16 ? 16 : 32;
            double dacc[CU];
#pragma unroll
            for (int k = 0; k < CU; ++k) dacc[k] = 0.0;
            schur_diag_walk<MODEL, NP, 64>(a, s.PV, s.cm_rec, s.cm_sc, 1, s.zero_fix, i, (int)dp->lo, (int)dp->hi, lane,
                                           reinterpret_cast<double2*>(reinterpret_cast<char*>(s_coop) + wave * (64 * 112)), dacc);
            double flat[CPAD];
            const double cam_mask = i < a.n_cam_fix ? 0.0 : 1.0;
#pragma unroll
            for (int e = 0; e < CPAD; ++e) flat[e] = (e < CU) ? dacc[e] : 0.0;
            const double total = cam_mask * wave_reduce_scatter<CPAD>(flat, lane, 32);
            const int e = rs_index<CPAD>(lane);
            double* const mine = s.dg_part + ((size_t)i * s.n_dg + chunk) * CU + e;
            const bool writer = (lane & (64 / CPAD - 1)) == 0 && e < CU;
            if (!s.arrive) {  // k_schur_finish adds the partials up
                if (writer) *mine = total;
                return;
            }
            if (writer) __hip_atomic_store(mine, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (bidx == 0 && threadIdx.x == 0)  // (the first item of the launch: everything in front of this kernel on the stream is complete)
                __hip_atomic_store(s.arrive + (size_t)SCHUR_ARRIVE_STRIDE * a.M, s.arrive_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int* cnt = s.dg_cnt + i;
            if (chunk + 1 < s.n_dg) {
                if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            if (lane == 0) {  // the camera's last item: every other one has been started before this one
                const long long t0 = wall_clock64();
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < s.n_dg - 1) {
                    __builtin_amdgcn_s_sleep(4);
                    if (wall_clock64() - t0 > 100000000ll) { atomicOr(s.fail, 2); break; }  // (1 s: a workgroup of this launch never ran)
                }
                __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_wave_barrier();
            const double lam = s.dg_lam_dev ? *s.dg_lam_dev : s.dg_lam;
#pragma unroll 1
            for (int o0 = 0; o0 < CU; o0 += 8) {  // eight outputs a pass, eight lanes an output (k_schur_finish's arrangement)
                const int k = o0 + (lane >> 3), sub = lane & 7;
                const bool live = k < CU;
                const double t = schur_diag_total<true>(s.dg_part, i, s.n_dg, CU, live ? k : 0, sub, live);
                if (live && sub == 0) schur_diag_store<true>(i, k, t, NP, a.n_c, lam, s.dg_lead, s.dg_gc, s.dg_scale_inv, S, s.dg_rhs, nullptr);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(s.arrive + (size_t)SCHUR_ARRIVE_STRIDE * i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    const double* cci = a.camc + (size_t)i * CAMC;
    const double* ccj = a.camc + (size_t)j * CAMC;

    double acc[NP][NP];
#pragma unroll
    for (int r = 0; r < NP; ++r)
#pragma unroll
        for (int q = 0; q < NP; ++q) acc[r][q] = 0.0;

    struct Rec { double2 r0, r1, r2, r3; double r4; };  // packed point record: X0 X1 | X2 v00 | v01 v02 | v11 v12 | v22
    // Affine cameras: J_c = A [D(X) | I] with A = (fx sk; 0 fy) and D(X) = rows 0, 1 of (dR/da X, dR/db X, dR/dg X), and
    // J_p = A R(rows 0, 1) does not depend on the point.  With m' = A_i^T [J_pi Vinv J_pj^T] A_j (2 x 2) the pair block is
    //     [D_i | I]^T m' [D_j | I]
    // -- 120 multiply-adds per hit instead of 198 for the generic form below (two full Jacobian evaluations, 2 x 5 blocks).
    // P_i, P_j: unit weights: A^T A R (so that m' = P_i Vinv P_j^T directly); otherwise A R, and A_i^T . A_j is applied after
    // the row scales.  Wave-uniform, computed once per item.
    double Pi_[2][3], Pj_[2][3], Ai_[3] = {0.0, 0.0, 0.0}, Aj_[3] = {0.0, 0.0, 0.0};
    double tri[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, trj[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // cos, sin of the three angles
    if constexpr (MODEL == AFFINE) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { tri[k] = cci[k]; trj[k] = ccj[k]; }
        // (wave-uniform values computed by the vector ALU go back to scalar registers: 12 doubles that live through the whole loop)
        auto uni = [](double x) {
            return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
        };
        auto cam_p = [&](const double* cc, double (&P)[2][3], double (&Au)[3]) {
            const double fx = cc[CAMX + 2], fy = cc[CAMX + 3], sk = cc[CAMX + 4];
            Au[0] = fx; Au[1] = sk; Au[2] = fy;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double p0 = fx * cc[6 + k] + sk * cc[9 + k], p1 = fy * cc[9 + k];  // A R
                if constexpr (UNITW) { P[0][k] = uni(fx * p0); P[1][k] = uni(sk * p0 + fy * p1); }  // A^T (A R)
                else { P[0][k] = uni(p0); P[1][k] = uni(p1); }
            }
        };
        cam_p(cci, Pi_, Ai_);
        cam_p(ccj, Pj_, Aj_);
    }
    // rows 0, 1 of the three angle derivatives of R X (satba_models.h: rotate), d[c][k]
    auto affine_d = [&](const double (&cc)[6], double X, double Y, double Z, double (&d)[2][3]) {
        const double ca = cc[0], sa = cc[1], cb = cc[2], sb = cc[3], cg = cc[4], sg = cc[5];
        const double y1y = ca * Y - sa * Z, y1z = sa * Y + ca * Z;
        const double y2x = cb * X + sb * y1z, y2z = -sb * X + cb * y1z;
        const double ax = sb * y1y;
        d[0][0] = cg * ax + sg * y1z; d[1][0] = sg * ax - cg * y1z;
        d[0][1] = cg * y2z;           d[1][1] = sg * y2z;
        d[0][2] = -(sg * y2x + cg * y1y); d[1][2] = cg * y2x - sg * y1y;
    };
    // JROWS (RPC): D' = row-scaled d(col,row)/dX' of both observations comes in ti / tj (six doubles each, gathered cooperatively by the
    // caller); the blocks are rebuilt from it, the point and the two cameras' rotations (rpc_jac_from_d)
    constexpr bool JROWS = MODEL == RPC;
    constexpr int JLEN = 6;
    auto compute = [&](const Rec& rc, int pi, int pj, const double2& scl_i, const double2& scl_j, const double* tip, const double* tjp) {
        const double X = rc.r0.x, Y = rc.r0.y, Z = rc.r1.x;
        const double v00 = rc.r1.y, v01 = rc.r2.x, v02 = rc.r2.y, v11 = rc.r3.x, v12 = rc.r3.y, v22 = rc.r4;
        if constexpr (MODEL == AFFINE) {
            double T[2][3];  // P_i Vinv
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                T[r][0] = Pi_[r][0] * v00 + Pi_[r][1] * v01 + Pi_[r][2] * v02;
                T[r][1] = Pi_[r][0] * v01 + Pi_[r][1] * v11 + Pi_[r][2] * v12;
                T[r][2] = Pi_[r][0] * v02 + Pi_[r][1] * v12 + Pi_[r][2] * v22;
            }
            double m[2][2];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) m[r][c] = T[r][0] * Pj_[c][0] + T[r][1] * Pj_[c][1] + T[r][2] * Pj_[c][2];
            if constexpr (!UNITW) {
                const double ax = scl_i.x * scl_i.x, ay = scl_i.y * scl_i.y;
                const double bx = scl_j.x * scl_j.x, by = scl_j.y * scl_j.y;
                m[0][0] *= ax * bx; m[0][1] *= ax * by; m[1][0] *= ay * bx; m[1][1] *= ay * by;
                // A_i^T m A_j
                const double t00 = Ai_[0] * m[0][0], t01 = Ai_[0] * m[0][1];
                const double t10 = Ai_[1] * m[0][0] + Ai_[2] * m[1][0], t11 = Ai_[1] * m[0][1] + Ai_[2] * m[1][1];
                m[0][0] = t00 * Aj_[0]; m[0][1] = t00 * Aj_[1] + t01 * Aj_[2];
                m[1][0] = t10 * Aj_[0]; m[1][1] = t10 * Aj_[1] + t11 * Aj_[2];
            }
            double Di[2][3], Dj[2][3];
            affine_d(tri, X, Y, Z, Di);
            affine_d(trj, X, Y, Z, Dj);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const double y0 = m[0][0] * Dj[0][q] + m[0][1] * Dj[1][q];
                const double y1 = m[1][0] * Dj[0][q] + m[1][1] * Dj[1][q];
#pragma unroll
                for (int r = 0; r < 3; ++r) acc[r][q] = fma(-Di[0][r], y0, fma(-Di[1][r], y1, acc[r][q]));
                if constexpr (NP == 5) { acc[3][q] -= y0; acc[4][q] -= y1; }
            }
            if constexpr (NP == 5) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) acc[r][3 + t] = fma(-Di[0][r], m[0][t], fma(-Di[1][r], m[1][t], acc[r][3 + t]));
                    acc[3][3 + t] -= m[0][t];
                    acc[4][3 + t] -= m[1][t];
                }
            }
            return;
        }
        double Jci[2][NP], Jpi[2][3], Jcj[2][NP], Jpj[2][3];
        if constexpr (MODEL == RPC) {
            const double Di[2][3] = {{tip[0], tip[1], tip[2]}, {tip[3], tip[4], tip[5]}};
            const double Dj[2][3] = {{tjp[0], tjp[1], tjp[2]}, {tjp[3], tjp[4], tjp[5]}};
            rpc_jac_from_d<NP>(cci, X, Y, Z, Di, Jci, Jpi);
            rpc_jac_from_d<NP>(ccj, X, Y, Z, Dj, Jcj, Jpj);
        } else {
            double u, v;
            project<MODEL, NP, true>(cci, nullptr, X, Y, Z, false, u, v, Jci, Jpi);
            project<MODEL, NP, true>(ccj, nullptr, X, Y, Z, false, u, v, Jcj, Jpj);
        }
        // Mm = Jp_i Vinv Jp_j^T (2 x 2)
        double A[2][3];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            A[r][0] = Jpi[r][0] * v00 + Jpi[r][1] * v01 + Jpi[r][2] * v02;
            A[r][1] = Jpi[r][0] * v01 + Jpi[r][1] * v11 + Jpi[r][2] * v12;
            A[r][2] = Jpi[r][0] * v02 + Jpi[r][1] * v12 + Jpi[r][2] * v22;
        }
        double m00 = A[0][0] * Jpj[0][0] + A[0][1] * Jpj[0][1] + A[0][2] * Jpj[0][2];
        double m01 = A[0][0] * Jpj[1][0] + A[0][1] * Jpj[1][1] + A[0][2] * Jpj[1][2];
        double m10 = A[1][0] * Jpj[0][0] + A[1][1] * Jpj[0][1] + A[1][2] * Jpj[0][2];
        double m11 = A[1][0] * Jpj[1][0] + A[1][1] * Jpj[1][1] + A[1][2] * Jpj[1][2];
        if constexpr (MODEL != RPC && !UNITW) {
            // row scales s of the two observations (weights, robust loss): both blocks of an observation carry them, so the
            // pair block is Jc_i^T [diag(s_i^2) (Jp_i Vinv Jp_j^T) diag(s_j^2)] Jc_j  (the fixed-point mask rides in Vinv)
            const double ax = scl_i.x * scl_i.x, ay = scl_i.y * scl_i.y;
            const double bx = scl_j.x * scl_j.x, by = scl_j.y * scl_j.y;
            m00 *= ax * bx; m01 *= ax * by; m10 *= ay * bx; m11 *= ay * by;
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const double y0 = m00 * Jcj[0][q] + m01 * Jcj[1][q];
            const double y1 = m10 * Jcj[0][q] + m11 * Jcj[1][q];
#pragma unroll
            for (int r = 0; r < NP; ++r) acc[r][q] = fma(-Jci[0][r], y0, fma(-Jci[1][r], y1, acc[r][q]));
        }
    };

    {
        const long long lo = dp->lo, hi = dp->hi;
        constexpr bool POS = !UNITW || MODEL == RPC;  // positions (and, weighted / robust, scales) ride along
        constexpr bool SCL = !UNITW && MODEL != RPC;
        // Cooperative record gathers.  A gather instruction costs the texture path one tag lookup per distinct line it touches;
        // with lane = hit every one of the five loads of a record touched 64 lines (two of the five removed: -0.13 ms of 0.55
        // at 200 x 1M x 10M).  Here the 320 16-byte pieces of the 64 records of an iteration are dealt to the lanes in order
        // (load t, lane l: piece 64 t + l = piece (64 t + l) % 5 of record (64 t + l) / 5), so an instruction touches 13-14 lines.
        // The pieces are written to LDS lane-linearly -- which IS the record-major image with stride 80 bytes, and with that
        // stride the 16 lanes of a ds_read_b128 phase fall into disjoint banks -- and every lane reads its own record back.
        // All lanes run the same number of iterations (cooperating lanes must be active): lanes past the end of the list are
        // pointed at record N, which is all zeros -- Vinv = 0 makes every term vanish.
        const int n_it = (int)((hi - lo + 63) >> 6);
        if (n_it > 0) {
            const long long last = hi - 1;
            // the lists are streamed once: non-temporal loads keep them from displacing the point records in L2
            auto ld = [&](const int* arr, long long k) { return __builtin_nontemporal_load(arr + (k < last ? k : last)); };
            // weighted / robust (SCL): the merged records (SchurArgs::wmode) -- `p` is the piece offset of the hit's record, `pi` the packed
            // distances of its two scales in front of it
            auto ldp = [&](long long k) { const int v = ld(SCL ? s.pair_rec : s.pair_pts, k); return (k < hi) ? v : (SCL ? s.zero_fix : a.N); };
            auto ldk = [&](long long k) { const int v = ld(s.pair_kk, k); return (k < hi) ? v : 0x00010001; };
            // piece g = 64 t + lane of the 64 NPC pieces (64 records x NPC) in load t: record g / NPC, piece g % NPC.
            // Weighted / robust (SCL): NPC = 7 -- pieces 5 and 6 of a "record" are the row scales of the hit's two observations
            // (round 5: they sit in the SAME variable-length record, in front of X0, Layout::w_fix; before that in a separate array in io order).  Gathered with lane = hit they were
            // two more instructions of 64 lines each per iteration; dealt to the lanes with the record pieces an instruction
            // touches ~9 records x 2 lines.
            constexpr int NPC = SCL ? 7 : 5;
            struct Coop { double2 c0, c1, c2, c3, c4, c5, c6; };  // (named members: an array here ends up in scratch)
            int rsrc[NPC];
            const double2* psrc[NPC];
            unsigned pcv[NPC];  // piece of the record (unit weights: a 32-bit offset from the uniform base instead of five 64-bit pointers)
            int kind[NPC];  // LDS slot of the index that addresses the piece: 64 x (0 record, 1 / 2 scale of observation i / j) + source lane
#pragma unroll
            for (int t = 0; t < NPC; ++t) {
                const int g = 64 * t + lane, pc = g % NPC;
                rsrc[t] = g / NPC;
                pcv[t] = (unsigned)pc;
                kind[t] = 64 * (pc < 5 ? 0 : pc - 4) + g / NPC;
                psrc[t] = pc < 5 ? s.PV + pc : reinterpret_cast<const double2*>(a.sc);
            }
            unsigned* sidx = s_idx[wave];
            auto coop_load = [&](int p, int pi, int pj) {
                Coop o;
                if constexpr (SCL) {
                    // the three indices of every hit go through LDS (one bpermute per piece would need all three per source lane)
                    asm volatile("" ::: "memory");
                    sidx[lane] = (unsigned)p; sidx[64 + lane] = (unsigned)p - ((unsigned)pi & 0xffffu); sidx[128 + lane] = (unsigned)p - ((unsigned)pi >> 16);
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    const unsigned o0 = sidx[kind[0]], o1 = sidx[kind[1]], o2 = sidx[kind[2]], o3 = sidx[kind[3]], o4 = sidx[kind[4]],
                                   o5 = sidx[kind[NPC - 2]], o6 = sidx[kind[NPC - 1]];
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    o.c0 = psrc[0][o0]; o.c1 = psrc[1][o1]; o.c2 = psrc[2][o2]; o.c3 = psrc[3][o3]; o.c4 = psrc[4][o4];
                    o.c5 = psrc[NPC - 2][o5]; o.c6 = psrc[NPC - 1][o6];
                } else {
#define SATBA_CL(t) s.PV[(unsigned)(PV_STRIDE / 2) * (unsigned)__shfl(p, rsrc[t]) + pcv[t]]
                    o.c0 = SATBA_CL(0); o.c1 = SATBA_CL(1); o.c2 = SATBA_CL(2); o.c3 = SATBA_CL(3); o.c4 = SATBA_CL(4);
#undef SATBA_CL
                }
                return o;
            };
            char* my = reinterpret_cast<char*>(s_coop) + wave * (64 * 112);
            double2* wr = reinterpret_cast<double2*>(my) + lane;  // piece g at 16 g bytes: record stride 16 NPC bytes
            const double2* mine = reinterpret_cast<const double2*>(my + lane * (16 * NPC));
            double2 si_cur = make_double2(1.0, 1.0), sj_cur = si_cur;
            auto transpose = [&](const Coop& o) {
                Rec r;
                asm volatile("" ::: "memory");
                wr[0] = o.c0; wr[64] = o.c1; wr[128] = o.c2; wr[192] = o.c3; wr[256] = o.c4;
                if constexpr (SCL) { wr[320] = o.c5; wr[384] = o.c6; }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                r.r0 = mine[0]; r.r1 = mine[1]; r.r2 = mine[2]; r.r3 = mine[3]; r.r4 = mine[4].x;
                if constexpr (SCL) { si_cur = mine[5]; sj_cur = mine[6]; }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                return r;
            };
            // RPC: the stored D' (six doubles = 3 pieces per observation in a 64-byte row, io order) of both observations is gathered the
            // same way (piece 64 t + lane: row (64 t + lane) / 3, piece (64 t + lane) % 3: 22 half-lines per instruction instead of 64
            // lines, six instructions per iteration; rounds 2 - 4: twelve, on 128-byte rows of Jc | Jp) and transposed through the same
            // LDS buffer, row stride 3 pieces (odd: conflict-free 16-byte reads)
            struct JCoop { double2 c0, c1, c2; };
            struct JRow { double v[JLEN]; };
            int jsrc[3];
            const double2* jp[3];
            double2* jw[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int g = 64 * t + lane;
                jsrc[t] = g / 3; jp[t] = reinterpret_cast<const double2*>(a.Jpm) + g % 3; jw[t] = reinterpret_cast<double2*>(my) + g;
            }
            auto jcoop_load = [&](int io) {
                JCoop o;
                constexpr size_t RS = jrow_stride(NP) / 2;  // row stride in 16-byte pieces
                o.c0 = jp[0][RS * (size_t)__shfl(io, jsrc[0])];
                o.c1 = jp[1][RS * (size_t)__shfl(io, jsrc[1])];
                o.c2 = jp[2][RS * (size_t)__shfl(io, jsrc[2])];
                return o;
            };
            const double2* jmine = reinterpret_cast<const double2*>(my) + lane * 3;
            auto jtranspose = [&](const JCoop& o) {
                JRow r;
                asm volatile("" ::: "memory");
                *jw[0] = o.c0; *jw[1] = o.c1; *jw[2] = o.c2;
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double2 t = jmine[k]; r.v[2 * k] = t.x; r.v[2 * k + 1] = t.y; }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                return r;
            };
            long long idx = lo + lane;
            int p_nxt = ldp(idx + 64);
            int pi_cur = 0, pj_cur = 0, pi_nxt = 0, pj_nxt = 0;
            if constexpr (SCL) {
                pi_cur = ldk(idx); pi_nxt = ldk(idx + 64);
            } else if constexpr (POS) {
                pi_cur = ld(s.pair_pi, idx); pj_cur = ld(s.pair_pj, idx);
                pi_nxt = ld(s.pair_pi, idx + 64); pj_nxt = ld(s.pair_pj, idx + 64);
            }
            Coop c_cur = coop_load(ldp(idx), pi_cur, pj_cur);
            JCoop ji_cur, jj_cur;
            if constexpr (JROWS) { ji_cur = jcoop_load(pi_cur); jj_cur = jcoop_load(pj_cur); }
            for (int it = 0; it < n_it; ++it) {
                const Rec r_cur = transpose(c_cur);  // first: its wait covers only loads of the previous iteration (sets si_cur, sj_cur)
                JRow ti, tj;
                if constexpr (JROWS) { ti = jtranspose(ji_cur); tj = jtranspose(jj_cur); }
                // indices run two iterations ahead, records one: neither latency is on the critical path
                const int p_nn = ldp(idx + 128);
                int pi_nn = 0, pj_nn = 0;
                if constexpr (SCL) pi_nn = ldk(idx + 128);
                else if constexpr (POS) { pi_nn = ld(s.pair_pi, idx + 128); pj_nn = ld(s.pair_pj, idx + 128); }
                const Coop c_nxt = coop_load(p_nxt, pi_nxt, pj_nxt);
                JCoop ji_nxt, jj_nxt;
                if constexpr (JROWS) { ji_nxt = jcoop_load(pi_nxt); jj_nxt = jcoop_load(pj_nxt); }
                // keep the gathers above the arithmetic: without the barrier the scheduler sinks them below compute() to
                // save registers and every iteration pays the full memory latency
                __builtin_amdgcn_sched_barrier(0);
                compute(r_cur, pi_cur, pj_cur, si_cur, sj_cur, ti.v, tj.v);
                __builtin_amdgcn_sched_barrier(0);
                p_nxt = p_nn; c_cur = c_nxt;
                if constexpr (JROWS) { ji_cur = ji_nxt; jj_cur = jj_nxt; }
                pi_cur = pi_nxt; pj_cur = pj_nxt; pi_nxt = pi_nn; pj_nxt = pj_nn;
                idx += 64;
            }
        }
    }

    // wave reduction of the NP x NP block; block (row j, col i) of the column-major lower triangle
    constexpr int NB2 = NP * NP;
    constexpr int NPAD = NB2 <= 16 ? 16 : (NB2 <= 32 ? 32 : 64);
    double flat[NPAD];
#pragma unroll
    for (int e = 0; e < NPAD; ++e) flat[e] = (e < NB2) ? acc[e / NP][e % NP] : 0.0;
    // fixed-camera masks
    const double cam_mask = (i < a.n_cam_fix || j < a.n_cam_fix) ? 0.0 : 1.0;
    const double total = cam_mask * wave_reduce_scatter<NPAD>(flat, lane, 32);
    const int e = rs_index<NPAD>(lane);
    const bool writer = (lane & (64 / NPAD - 1)) == 0 && e < NB2;  // one lane per total (NPAD = 64: every lane)
    // (the epilogue's address arithmetic starts here: without the fences it was hoisted above the hit loop -- 191 registers instead of
    // 152 in the weighted kernel, two waves per SIMD)
    int e_ = e, i_ = i, j_ = j, chunk_ = chunk;
    long long pair_ = pair;
    asm volatile("" : "+v"(e_), "+s"(i_), "+s"(j_), "+s"(chunk_), "+s"(pair_));
    // the first item tells the factorisation that this kernel runs: everything in front of it on the stream (diagonal blocks, right-hand side) is complete
    if (s.arrive && bidx == 0 && threadIdx.x == 0)
        __hip_atomic_store(s.arrive + (size_t)SCHUR_ARRIVE_STRIDE * a.M, s.arrive_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double* const s_out = S + (size_t)(j_ * NP + e_ % NP) + (size_t)(i_ * NP + e_ / NP) * a.n_c;  // S[(j,q), (i,r)] = (W_i Vinv W_j^T)[r][q], e = r NP + q
    if constexpr (!UNITW) {
        if (s.arrive && s.n_chunks > 1) {  // beside the factorisation, chunk partials (SchurArgs::pair_cnt)
            schur_pair_publish(s.pair_part + ((size_t)chunk_ * n_pairs + pair_) * NB2 + e_, (size_t)n_pairs * NB2, s.n_chunks, chunk_, s.pair_cnt + pair_,
                               s.arrive + (size_t)SCHUR_ARRIVE_STRIDE * i_, s.fail, s_out, total, writer);
            return;
        }
    }
    if (writer) {
        if (s.n_chunks > 1) s.pair_part[((size_t)chunk_ * n_pairs + pair_) * NB2 + e_] = total;
        else if (!s.arrive) *s_out = total;
        else __hip_atomic_store(s_out, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (s.arrive && s.n_chunks == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(s.arrive + (size_t)SCHUR_ARRIVE_STRIDE * i_, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int MODEL, int NP, bool UNITW>
__global__ __launch_bounds__(256) SATBA_PAIRS_WAVES_ATTR(MODEL, UNITW) void k_schur_pairs(ObsArgs a, SchurArgs s, double* __restrict__ S) {
    SATBA_GATE(a.gate);
    __shared__ double2 s_coop[4 * 64 * 7];
    __shared__ unsigned s_idx[4][3 * 64];
    schur_pairs_body<MODEL, NP, UNITW>(a, s, S, blockIdx.x, s_coop, s_idx);
}


// Diagonal blocks and right-hand side: camera-major pass, registers only.
//   S_ii += sum_p (Jc^T Jc - W_ip Vinv W_ip^T),   rhs_i -= sum_p W_ip Vinv g_p.
// The J_c^T J_c term is the U_c block, which the linearize kernel therefore does not have to accumulate.
// schur_diag_walk: the entries lo .. hi - 1 of camera cam's list, THREADS threads (tid) striding over them, every thread
// accumulating cam_acc_len(NP) sums in registers; shared by k_schur_diag (a workgroup per (camera, chunk)) and by the diagonal items
// of k_schur_pairs (one wave).  base: the records (PV, or the merged records W of the weighted / robust runs, Layout::w_fix);
// lst_rec: per entry the point (rmul = 8: fixed-stride records) or the record's piece offset (rmul = 1); lst_sc: per entry the index of
// its row scales in a.sc (RPC: of its stored Jacobian rows); zero_rec: piece offset of an all-zero record; my: 64 x 7 double2 of LDS
// per wave.
template <int MODEL, int NP, int THREADS>
__device__ __forceinline__ void schur_diag_walk(const ObsArgs& a, const double2* __restrict__ base, const int* __restrict__ lst_rec,
                                                const int* __restrict__ lst_sc, int rmul, int zero_rec, int cam, int lo, int hi, int tid,
                                                double2* my, double (&acc)[cam_acc_len(NP)]) {
    const double* cc = a.camc + (size_t)cam * CAMC;
    // Software pipeline: point indices run two iterations ahead, the point records one.  The records are gathered cooperatively
    // (see k_schur_pairs): the 384 16-byte pieces of a wave's 64 records are dealt to the lanes in order (load t, lane l: piece
    // (64 t + l) % 6 of record (64 t + l) / 6), so a gather instruction touches 11 lines instead of 64, and transposed through
    // LDS (record stride 7 pieces = 112 bytes: the 16 lanes of a ds_read_b128 phase fall into disjoint banks).  All threads run
    // the same number of iterations; threads past the end of the list are pointed at record N (zeros) and masked.
    struct Rec { double2 r0, r1, r2, r3, r4, r5; };
    const int lane = tid & 63;
    int rsrc[6];
    const double2* psrc[6];
    double2* wdst[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int g = 64 * t + lane;
        rsrc[t] = g / 6; psrc[t] = base + g % 6; wdst[t] = my + (g / 6) * 7 + g % 6;
    }
    struct Coop { double2 c0, c1, c2, c3, c4, c5; };
    auto coop_load = [&](int p) {
        Coop o;
        o.c0 = psrc[0][(size_t)__shfl(p, rsrc[0])];
        o.c1 = psrc[1][(size_t)__shfl(p, rsrc[1])];
        o.c2 = psrc[2][(size_t)__shfl(p, rsrc[2])];
        o.c3 = psrc[3][(size_t)__shfl(p, rsrc[3])];
        o.c4 = psrc[4][(size_t)__shfl(p, rsrc[4])];
        o.c5 = psrc[5][(size_t)__shfl(p, rsrc[5])];
        return o;
    };
    const double2* mine = my + lane * 7;
    auto transpose = [&](const Coop& o) {
        Rec r;
        asm volatile("" ::: "memory");
        *wdst[0] = o.c0; *wdst[1] = o.c1; *wdst[2] = o.c2; *wdst[3] = o.c3; *wdst[4] = o.c4; *wdst[5] = o.c5;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        r.r0 = mine[0]; r.r1 = mine[1]; r.r2 = mine[2]; r.r3 = mine[3]; r.r4 = mine[4]; r.r5 = mine[5];
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        return r;
    };
    // affine cameras (block-uniform): J_c = A [D(X) | I], J_p = A R  (see k_schur_pairs)
    double Pm[2][3], Au[3] = {0.0, 0.0, 0.0}, tr[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if constexpr (MODEL == AFFINE) {
        const double fx = cc[CAMX + 2], fy = cc[CAMX + 3], sk = cc[CAMX + 4];
        Au[0] = fx; Au[1] = sk; Au[2] = fy;
#pragma unroll
        for (int k = 0; k < 3; ++k) { Pm[0][k] = fx * cc[6 + k] + sk * cc[9 + k]; Pm[1][k] = fy * cc[9 + k]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) tr[k] = cc[k];
    }
    const int n_it = (hi - lo + THREADS - 1) / THREADS;
    if (n_it > 0) {
        // (a record is addressed by its first 16-byte piece: rmul = 8 for the fixed-stride records PV, 1 when the list holds piece offsets)
        auto ldp = [&](int q) { const int v = lst_rec[q < hi ? q : hi - 1]; return q < hi ? rmul * v : zero_rec; };
        int pos = lo + tid;
        int p_nxt = ldp(pos + THREADS);
        Coop cur = coop_load(ldp(pos));
        // weighted / robust runs (affine, perspective): the row scales of the observation, io order.  Indices two iterations ahead,
        // scales one, like the records (round 2 loaded them where they were used: a dependent index -> scale chain in every
        // iteration, 0.28 ms against 0.13 for the unit-weight kernel)
        const bool scl = MODEL != RPC && a.sc != nullptr;
        auto ldio = [&](int q) { return lst_sc[q < hi ? q : hi - 1]; };
        int io_nxt = scl ? ldio(pos + THREADS) : 0;
        double2 sc_cur = scl ? a.sc[ldio(pos)] : make_double2(1.0, 1.0);
        for (int it = 0; it < n_it; ++it, pos += THREADS) {
            const Rec rc = transpose(cur);  // first: its wait covers only loads of the previous iteration
            const int p_nn = ldp(pos + 2 * THREADS);
            const int io_nn = scl ? ldio(pos + 2 * THREADS) : 0;
            const double2 sc_nxt = scl ? a.sc[io_nxt] : make_double2(1.0, 1.0);
            const Coop nxt = coop_load(p_nxt);
            __builtin_amdgcn_sched_barrier(0);
            const double2 sc_now = sc_cur;
            sc_cur = sc_nxt; io_nxt = io_nn;
            const double2 r0 = rc.r0, r1 = rc.r1, r2 = rc.r2, r3 = rc.r3, r4 = rc.r4, r5 = rc.r5;
            p_nxt = p_nn; cur = nxt;
            const bool valid = pos < hi;
            const int posc = valid ? pos : hi - 1;
            const double vm = valid ? 1.0 : 0.0;
            double sx = vm, sy = vm;  // squared row scales times the fixed-point mask, applied to the 2 x 2 middle matrix
            double ux = vm, uy = vm;  // squared row scales on the J_c^T J_c term (no point mask there)
            const double v00 = r1.y, v01 = r2.x, v02 = r2.y, v11 = r3.x, v12 = r3.y, v22 = r4.x;
            if constexpr (MODEL == AFFINE) {
                if (a.sc) { const double2 t = sc_now; ux = vm * t.x * t.x; uy = vm * t.y * t.y; }
                sx = ux; sy = uy;  // the fixed-point mask rides in the record's Vinv
                double T[2][3];  // J_p Vinv
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    T[r][0] = Pm[r][0] * v00 + Pm[r][1] * v01 + Pm[r][2] * v02;
                    T[r][1] = Pm[r][0] * v01 + Pm[r][1] * v11 + Pm[r][2] * v12;
                    T[r][2] = Pm[r][0] * v02 + Pm[r][1] * v12 + Pm[r][2] * v22;
                }
                // N = diag(u) - diag(s u) (J_p Vinv J_p^T), the 2 x 2 middle matrix of J_c^T J_c - W Vinv W^T; then A^T N A
                const double n00 = ux - sx * ux * (T[0][0] * Pm[0][0] + T[0][1] * Pm[0][1] + T[0][2] * Pm[0][2]);
                const double n01 = -sx * uy * (T[0][0] * Pm[1][0] + T[0][1] * Pm[1][1] + T[0][2] * Pm[1][2]);
                const double n11 = uy - sy * uy * (T[1][0] * Pm[1][0] + T[1][1] * Pm[1][1] + T[1][2] * Pm[1][2]);
                const double ag0 = sx * (T[0][0] * r4.y + T[0][1] * r5.x + T[0][2] * r5.y);
                const double ag1 = sy * (T[1][0] * r4.y + T[1][1] * r5.x + T[1][2] * r5.y);
                // A = (fx sk; 0 fy):  A^T N A, A^T ag
                const double t00 = Au[0] * n00, t01 = Au[0] * n01;
                const double t10 = Au[1] * n00 + Au[2] * n01, t11 = Au[1] * n01 + Au[2] * n11;
                const double e00 = t00 * Au[0], e01 = t00 * Au[1] + t01 * Au[2], e11 = t10 * Au[1] + t11 * Au[2];
                const double b0 = Au[0] * ag0, b1 = Au[1] * ag0 + Au[2] * ag1;
                // rows 0, 1 of the three angle derivatives of R X
                double D[2][3];
                {
                    const double X = r0.x, Y = r0.y, Z = r1.x;
                    const double ca = tr[0], sa = tr[1], cb = tr[2], sb = tr[3], cg = tr[4], sg = tr[5];
                    const double y1y = ca * Y - sa * Z, y1z = sa * Y + ca * Z;
                    const double y2x = cb * X + sb * y1z, y2z = -sb * X + cb * y1z;
                    const double ax = sb * y1y;
                    D[0][0] = cg * ax + sg * y1z; D[1][0] = sg * ax - cg * y1z;
                    D[0][1] = cg * y2z;           D[1][1] = sg * y2z;
                    D[0][2] = -(sg * y2x + cg * y1y); D[1][2] = cg * y2x - sg * y1y;
                }
                // [D | I]^T E [D | I], upper triangle row by row (the order of cam_acc_len: (r, q >= r)), then the right-hand side
                double y0[3], y1[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) { y0[q] = e00 * D[0][q] + e01 * D[1][q]; y1[q] = e01 * D[0][q] + e11 * D[1][q]; }
                int k = 0;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int q = r; q < 3; ++q) acc[k++] += D[0][r] * y0[q] + D[1][r] * y1[q];
                    if constexpr (NP == 5) { acc[k++] += y0[r]; acc[k++] += y1[r]; }
                }
                if constexpr (NP == 5) { acc[k++] += e00; acc[k++] += e01; acc[k++] += e11; }
#pragma unroll
                for (int r = 0; r < 3; ++r) acc[k++] -= D[0][r] * b0 + D[1][r] * b1;
                if constexpr (NP == 5) { acc[k++] -= b0; acc[k++] -= b1; }
            } else {
                double Jc[2][NP], Jp[2][3];
                if constexpr (MODEL == RPC) {  // from D' the linearize kernel stored (the row scales ride in it; the fixed-point mask in the record's Vinv)
                    ObsEval<MODEL, NP, true> e2;
                    e2.load_d(a, lst_sc[posc]);
                    rpc_jac_from_d<NP>(cc, r0.x, r0.y, r1.x, e2.Dr, Jc, Jp);
#pragma unroll
                    for (int k = 0; k < NP; ++k) { Jc[0][k] *= vm; Jc[1][k] *= vm; }
                    ux = uy = sx = sy = 1.0;
                } else {
                    double u, v;
                    project<MODEL, NP, true>(cc, nullptr, r0.x, r0.y, r1.x, false, u, v, Jc, Jp);
                    if (a.sc) { const double2 t = sc_now; ux = vm * t.x * t.x; uy = vm * t.y * t.y; }
                    sx = ux; sy = uy;  // the fixed-point mask rides in the record's Vinv
                }
                double A[2][3];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    A[r][0] = Jp[r][0] * v00 + Jp[r][1] * v01 + Jp[r][2] * v02;
                    A[r][1] = Jp[r][0] * v01 + Jp[r][1] * v11 + Jp[r][2] * v12;
                    A[r][2] = Jp[r][0] * v02 + Jp[r][1] * v12 + Jp[r][2] * v22;
                }
                // W Vinv W^T = Jc^T [diag(s) Jp Vinv Jp^T diag(s)] Jc with s = row scale^2 (both blocks carry the row scale) -- one
                // more factor s on each side comes from Jc: (s_r Jp_r) Vinv (s_q Jp_q)^T sandwiched by (s Jc)
                const double m00 = sx * ux * (A[0][0] * Jp[0][0] + A[0][1] * Jp[0][1] + A[0][2] * Jp[0][2]);
                const double m01 = sx * uy * (A[0][0] * Jp[1][0] + A[0][1] * Jp[1][1] + A[0][2] * Jp[1][2]);
                const double m11 = sy * uy * (A[1][0] * Jp[1][0] + A[1][1] * Jp[1][1] + A[1][2] * Jp[1][2]);
                // rhs: W Vinv g_p = Jc^T diag(s) Jp Vinv g_p
                const double ag0 = sx * (A[0][0] * r4.y + A[0][1] * r5.x + A[0][2] * r5.y);
                const double ag1 = sy * (A[1][0] * r4.y + A[1][1] * r5.x + A[1][2] * r5.y);
                int k = 0;
#pragma unroll
                for (int r = 0; r < NP; ++r) {
                    const double y0 = m00 * Jc[0][r] + m01 * Jc[1][r];
                    const double y1 = m01 * Jc[0][r] + m11 * Jc[1][r];
#pragma unroll
                    for (int q = r; q < NP; ++q) {
                        acc[k] -= Jc[0][q] * y0 + Jc[1][q] * y1;
                        acc[k] += ux * Jc[0][r] * Jc[0][q] + uy * Jc[1][r] * Jc[1][q];
                        ++k;
                    }
                }
#pragma unroll
                for (int r = 0; r < NP; ++r) acc[k++] -= Jc[0][r] * ag0 + Jc[1][r] * ag1;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// grid (M, chunks): the workgroups of one chunk (the same slice of every camera's point-sorted list, i.e. about the same
// point range) are dispatched together and share their point records in L2.  part [M][chunks][CU]
// L: the workgroup's index among the M x n_chunks diagonal workgroups of the launch; s_coop: (LINC_THREADS / 64) x 64 x 7 double2, s_red: the waves' sums
template <int MODEL, int NP>
__device__ __forceinline__ void schur_diag_body(const ObsArgs& a, const CamMajor& c, const SchurArgs& s, double* __restrict__ part, const int L, const int M,
                                                const int n_chunks, double2* s_coop, double (*s_red)[cam_acc_len(NP)]) {
    constexpr int CU = cam_acc_len(NP);
    // s.diag_xcd (chunk count a multiple of 8): workgroup L of the launch runs on XCD L % 8 (observed placement); XCD x takes
    // the chunks = x (mod 8), every camera's slice of one chunk in a row, so that the records (and, weighted runs, the scale lines)
    // of a point range are fetched by ONE XCD instead of by the ~6 that hold one of the point's cameras
    int cam = L % M, chunk = L / M;
    if (s.diag_xcd) {
        const int x = L & 7, q = L >> 3;
        cam = q % M; chunk = (q / M) * 8 + x;
    }
    const int b = c.cam_ofs[cam], e = c.cam_ofs[cam + 1];
    const long long len = e - b;
    const int lo = b + (int)(len * chunk / n_chunks), hi = b + (int)(len * (chunk + 1) / n_chunks);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc[CU];
#pragma unroll
    for (int k = 0; k < CU; ++k) acc[k] = 0.0;
    schur_diag_walk<MODEL, NP, LINC_THREADS>(a, s.PV, c.pt, c.io, s.wmode ? 1 : PV_STRIDE / 2, s.wmode ? s.zero_fix : (PV_STRIDE / 2) * a.N, cam, lo, hi,
                                             (int)threadIdx.x, s_coop + wave * (64 * 7), acc);
    if (cam < a.n_cam_fix) {  // fixed camera (block-uniform)
#pragma unroll
        for (int k = 0; k < CU; ++k) acc[k] = 0.0;
    }
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
template <int MODEL, int NP>
__global__ __launch_bounds__(LINC_THREADS) void k_schur_diag(ObsArgs a, CamMajor c, SchurArgs s, double* __restrict__ part) {
    SATBA_GATE(a.gate);
    __shared__ double2 s_coop[(LINC_THREADS / 64) * 64 * 7];
    __shared__ double s_red[LINC_THREADS / 64][cam_acc_len(NP)];
    schur_diag_body<MODEL, NP>(a, c, s, part, (int)(blockIdx.y * gridDim.x + blockIdx.x), (int)gridDim.x, (int)gridDim.y, s_coop, s_red);
}
// Diagonal pass and pair kernel in ONE launch (round 6; fronts with the factorisation behind them, not beside them): the two are
// independent -- both read k_vinv's records -- and below ~10 000 pairs neither fills the chip (50 cameras: 400 + 306 workgroups for 768
// slots), so they run side by side where two launches ran one after the other.  Workgroups [0, n_diag_pad) are the diagonal ones
// (n_diag_pad: n_diag rounded up to a multiple of 8, so that a pair workgroup's index keeps its XCD), the rest the pair kernel's.
static_assert(LINC_THREADS == 256, "k_schur_both: both bodies run 256 threads");
template <int MODEL, int NP, bool UNITW>
__global__ __launch_bounds__(256) SATBA_PAIRS_WAVES_ATTR(MODEL, UNITW) void k_schur_both(ObsArgs a, CamMajor c, SchurArgs s, double* __restrict__ part, double* __restrict__ S,
                                                                                      int n_diag, int n_diag_pad, int M, int n_chunks) {
    SATBA_GATE(a.gate);
    __shared__ double2 s_coop[4 * 64 * 7];
    __shared__ unsigned s_idx[4][3 * 64];
    __shared__ double s_red[4][cam_acc_len(NP)];
    if ((int)blockIdx.x < n_diag_pad) {
        if ((int)blockIdx.x < n_diag) schur_diag_body<MODEL, NP>(a, c, s, part, (int)blockIdx.x, M, n_chunks, s_coop, s_red);
        return;
    }
    schur_pairs_body<MODEL, NP, UNITW>(a, s, S, blockIdx.x - (unsigned)n_diag_pad, s_coop, s_idx);
}

// End of the Schur phase, one launch (three in rounds 1-3: k_schur_init in front of the phase, k_schur_pairs_reduce and k_schur_diag_finish
// behind it -- 9 us of a 140 us iteration at 10 cameras):
//   workgroups < nb_diag   S_ii = (lead) lam Dc^2 on the diagonal + the chunk partials of k_schur_diag (both triangles of the block are
//                          written), rhs_i = (lead) g_c + partials: eight threads per output, each over every eighth chunk, combined in a
//                          fixed order (up to 64 chunks: a single thread's dependent loads were 18 us with 32 chunks); the exchange header
//                          (hdr_len doubles at xb), which the phases after this one accumulate into, is cleared here (k_vinv, in front
//                          of the phase, still reads it)
//   workgroups >= nb_diag  (red_chunks > 1: the pair kernel left point-range partials) S block of each pair = sum of its chunk partials,
//                          chunk order: repeatable
// Every block of the lower triangle is written by a kernel of the Schur phase (the off-diagonal ones here or by k_schur_pairs, also for
// pairs without a common point), so S is not cleared first (an 8 MB fill per iteration at 200 cameras x 5); the strict upper triangle is
// never read.  lam_dev (optional): the damping is read from device memory (keep[5], written by k_vinv) instead of the argument.
// rhs_scaled (or null): every entry of S passes through this launch and the dense solve follows at once (one rank): the system goes out in
// scaled variables -- S / (scale_inv_r scale_inv_c), rhs_scaled = rhs / scale_inv, k_scale_system's arithmetic -- and the solver's
// status word and flags (n_clear ints at clear) are cleared: that launch is gone, too.
__global__ __launch_bounds__(256) void k_schur_finish(int M, int NP, int n_c, int n_chunks, const double* __restrict__ part, double lam,
                                                      const double* __restrict__ lam_dev, double lead, const double* __restrict__ gc,
                                                      const double* __restrict__ scale_inv, double* __restrict__ S, double* __restrict__ rhs,
                                                      double* __restrict__ xb, int hdr_len, int nb_diag, int red_chunks,
                                                      const int2* __restrict__ pair_ij, const double* __restrict__ pair_part, const int* gate,
                                                      double* __restrict__ rhs_scaled = nullptr, int* __restrict__ clear = nullptr, int n_clear = 0) {
    SATBA_GATE(gate);
    if ((int)blockIdx.x >= nb_diag) {
        const long long n_pairs = (long long)M * (M - 1) / 2;
        const int NB2 = NP * NP;
        const long long idx = (long long)(blockIdx.x - nb_diag) * blockDim.x + threadIdx.x;
        if (idx >= n_pairs * NB2) return;
        const long long pair = idx / NB2;
        const int e = (int)(idx % NB2);
        double t = 0.0;
        int ch = 0;
        for (; ch + 4 <= red_chunks; ch += 4) {  // (four loads in flight; the additions stay in chunk order)
            const double p0 = pair_part[((size_t)ch * n_pairs + pair) * NB2 + e], p1 = pair_part[((size_t)(ch + 1) * n_pairs + pair) * NB2 + e];
            const double p2 = pair_part[((size_t)(ch + 2) * n_pairs + pair) * NB2 + e], p3 = pair_part[((size_t)(ch + 3) * n_pairs + pair) * NB2 + e];
            t += p0; t += p1; t += p2; t += p3;
        }
        for (; ch < red_chunks; ++ch) t += pair_part[((size_t)ch * n_pairs + pair) * NB2 + e];
        const int2 ij = pair_ij[pair];
        const int r = e / NP, q = e % NP;
        if (rhs_scaled) t /= scale_inv[ij.y * NP + q] * scale_inv[ij.x * NP + r];
        S[(size_t)(ij.y * NP + q) + (size_t)(ij.x * NP + r) * n_c] = t;
        return;
    }
    if (lam_dev) lam = *lam_dev;
    const int CU = cam_acc_len(NP);
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, idx = gid >> 3, sub = gid & 7;
    if (gid < hdr_len) xb[gid] = 0.0;
    if (gid < n_clear) clear[gid] = 0;
    if (n_chunks <= 0) return;  // the diagonal blocks and the right-hand side are written by the pair kernel itself (SchurArgs::dg_cnt)
    const bool live = idx < M * CU;
    const int cam = live ? idx / CU : 0, k = live ? idx % CU : 0;
    const double t = schur_diag_total<false>(part, cam, n_chunks, CU, k, sub, live);
    if (!live || sub != 0) return;
    schur_diag_store<false>(cam, k, t, NP, n_c, lam, lead, gc, scale_inv, S, rhs, rhs_scaled);
}

}  // namespace satba
