// satba_schur3.h -- K3 v3: Schur complement by camera-pair intersection.  No atomics on the data path,
// everything accumulates in registers.
//
//   S_ij -= sum over points p seen by BOTH cameras i and j of  W_ip Vinv_p W_jp^T,   W = Jc^T Jp  (NP x 3)
//
// v2 (k_schur_panel) walks observation pairs point by point and scatters NP x NP blocks into an LDS panel with
// ds_add_f64: PMC showed it LDS-atomic bound at 48 % lane utilisation (profiles/r1_pmc_linearize_schur.txt).
// v3 turns the scatter into a gather: one wavefront owns ONE camera pair (i, j), finds the points the two cameras
// share by AND-ing their visibility bitmaps (64 points per word, lane = word), compacts the hits into a small
// per-wave queue, and evaluates 64 hits at a time with every lane busy.  Both cameras are wave-uniform, so their
// constants sit in scalar registers; the NP x NP block accumulates in registers and is reduced across the wave
// once, then stored -- each off-diagonal block of S is written exactly once, no zero-fill, no reduction pass.
// The diagonal blocks and the right-hand side come from a camera-major pass (k_schur_diag) that also
// accumulates in registers.
//
// Index structures (host, once per problem): visibility bitmaps bits[M][NW], rank[M][NW] = number of
// observations of the camera before word w, and the camera-major copy of the observation data; the position of
// (camera, point) in that copy is cam_ofs[c] + rank[c][w] + popcount(bits[c][w] below the point's bit).
#pragma once
#include "satba_kernels.h"
#include "satba_linearize3.h"

namespace satba {

struct Schur3Args {
    const unsigned long long* __restrict__ bits;  // M x NW
    const int* __restrict__ rank;                 // M x NW
    const double* __restrict__ Vinv;              // N x 6
    const double* __restrict__ gp;                // N x 3
    const double2* __restrict__ PV;               // N x 6 double2: X0 X1 | X2 v00 | v01 v02 | v11 v12 | v22 g0 | g1 g2
    const long long* __restrict__ pair_ofs;       // n_pairs + 1 (or null): offsets into pair_pts
    const int* __restrict__ pair_pts;             // points shared by each camera pair, ascending (static per problem)
    double* __restrict__ pair_part;               // n_chunks x n_pairs x NP*NP partial blocks (list path, n_chunks > 1)
    int chunk_mul;                                // the lists are cut into n_chunks * chunk_mul fine chunks (pair_ofs stride + 1)
    const int* __restrict__ pair_pi;              // per list entry: observation index (point-major order) of the point's observation
    const int* __restrict__ pair_pj;              //   in camera i / j (null: not built); per-observation data of the two lies within
                                                  //   the point's contiguous run of observations
    const int2* __restrict__ pair_ij;             // pair index -> (i, j), i < j (null: unranked with a square root)
    int NW;                                       // words per camera
    int n_chunks;                                 // word-range chunks per pair (1: plain stores, >1: atomics)
};

constexpr int S3_QUEUE = 128;  // per-wave hit queue (entries): < 64 pending + at most 64 added per round

// weighted, scaled Jacobian blocks of one (camera, point) from the camera-major copy
// UNITW: every observation weight is 1 and the loss is linear -> nothing has to be fetched per observation
template <int MODEL, int NP, bool ROBUST, bool UNITW = false>
__device__ inline void cm_jacobian(const ObsArgs& a, const CamMajor& c, const double* cc, const double* tab, int cam,
                                   int pos, int pt, double X, double Y, double Z, double Jc[2][NP], double Jp[2][3],
                                   double2* scales = nullptr) {
    double w = 1.0;
    if constexpr (!UNITW) w = c.w[pos];
    double u, v;
    project<MODEL, NP, true>(cc, tab, X, Y, Z, a.f32 != 0, u, v, Jc, Jp);
    double js0 = 1.0, js1 = 1.0;
    if constexpr (ROBUST) {
        const double2 ob = c.obs[pos];
        double r0, r1, fs0, fs1;
        robust(a.loss, a.f_scale, w * (u - ob.x), r0, fs0, js0);
        robust(a.loss, a.f_scale, w * (v - ob.y), r1, fs1, js1);
    }
    const double mc = (cam >= a.n_cam_fix) ? 1.0 : 0.0, mp = (pt >= a.n_pts_fix) ? 1.0 : 0.0;
    const double s0 = w * js0, s1 = w * js1;
#pragma unroll
    for (int k = 0; k < NP; ++k) { Jc[0][k] *= s0 * mc; Jc[1][k] *= s1 * mc; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { Jp[0][k] *= s0 * mp; Jp[1][k] *= s1 * mp; }
    if (scales) *scales = make_double2(s0, s1);
}

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

// grid: one wave per (pair, chunk); 4 waves per workgroup.  pair index -> (i, j), i < j.
// SCL (weighted / robust runs with pair lists that carry observation indices): unit-weight Jacobians times the stored
// row scales -- a separate instantiation, so that it does not carry the registers of the robust evaluation
template <int MODEL, int NP, bool ROBUST, bool UNITW, bool SCL = false>
__device__ __forceinline__ void schur_pairs_body(const ObsArgs& a, const CamMajor& c, const Schur3Args& s, double* __restrict__ S) {
    __shared__ int s_q[4][S3_QUEUE];  // shared points of the two cameras waiting to be evaluated
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n_pairs = (long long)a.M * (a.M - 1) / 2;
    // list path: grid (pairs / 4, chunks) -- chunk-major dispatch order, so that the workgroups running at the same time
    // gather point records from the same slice of the point array (an L2-sized window) instead of the whole array.
    // bitmap path: 1-D grid over (pair, chunk), chunk fastest.  32-bit arithmetic only: the four 64-bit divisions this
    // used to take were ~500 of the ~2000 VALU instructions of a work item.
    unsigned pair_u, chunk_u;
    if (s.pair_ofs) {
        pair_u = blockIdx.x * 4u + (unsigned)wave;
        chunk_u = blockIdx.y;
        if ((long long)pair_u >= n_pairs) return;
    } else {
        const unsigned item = blockIdx.x * 4u + (unsigned)wave;
        if ((long long)item >= n_pairs * s.n_chunks) return;
        pair_u = item / (unsigned)s.n_chunks;
        chunk_u = item % (unsigned)s.n_chunks;
    }
    const long long pair = pair_u;
    const int chunk = (int)chunk_u;
    int i, j;
    if (s.pair_ij) {
        const int2 ij = s.pair_ij[pair_u];
        i = ij.x; j = ij.y;
    } else {
        // unrank pair -> (i, j): pairs of row i start at i*M - i*(i+1)/2
        i = (int)((2.0 * a.M - 1.0 - sqrt((2.0 * a.M - 1.0) * (2.0 * a.M - 1.0) - 8.0 * (double)pair)) * 0.5);
        while ((long long)i * a.M - (long long)i * (i + 1) / 2 > pair) --i;
        while ((long long)(i + 1) * a.M - (long long)(i + 1) * (i + 2) / 2 <= pair) ++i;
        j = i + 1 + (int)(pair - ((long long)i * a.M - (long long)i * (i + 1) / 2));
    }
    i = __builtin_amdgcn_readfirstlane(i);  // wave-uniform by construction: lets the camera constants use scalar loads
    j = __builtin_amdgcn_readfirstlane(j);

    const unsigned long long* bi = s.bits + (size_t)i * s.NW;
    const unsigned long long* bj = s.bits + (size_t)j * s.NW;
    const int* ri = s.rank + (size_t)i * s.NW;
    const int* rj = s.rank + (size_t)j * s.NW;
    const int base_i = c.cam_ofs[i], base_j = c.cam_ofs[j];
    const double* cci = a.camc + (size_t)i * CAMC;  // wave-uniform: scalar loads
    const double* ccj = a.camc + (size_t)j * CAMC;
    const double* tabi = (MODEL == RPC) ? a.rpc + (size_t)i * 90 : nullptr;
    const double* tabj = (MODEL == RPC) ? a.rpc + (size_t)j * 90 : nullptr;

    double acc[NP][NP];
#pragma unroll
    for (int r = 0; r < NP; ++r)
#pragma unroll
        for (int q = 0; q < NP; ++q) acc[r][q] = 0.0;

    struct Rec { double2 r0, r1, r2, r3; double r4; };  // packed point record: X0 X1 | X2 v00 | v01 v02 | v11 v12 | v22
    auto load_rec = [&](int p) {
        const double2* pv = s.PV + (PV_STRIDE / 2) * (size_t)p;  // four 16-byte gathers and one of 8 instead of nine 8-byte ones
        Rec r;
        r.r0 = pv[0]; r.r1 = pv[1]; r.r2 = pv[2]; r.r3 = pv[3]; r.r4 = reinterpret_cast<const double*>(pv + 4)[0];
        return r;
    };
    // listpos: pi, pj = observation indices of (camera i, point) and (camera j, point) from the pair list, scl = the
    // Jacobian row scales of the two observations (weighted / robust runs) -- the Jacobians are then evaluated for unit weight and linear loss and scaled, which is what
    // the weighted / robust evaluation does, without fetching the observation or running the loss function
    auto compute = [&](int p, const Rec& rc, int pi, int pj, const double2& scl_i, const double2& scl_j, bool listpos) {
        if (!listpos) {
            pi = pj = 0;
            if constexpr ((!UNITW && !SCL) || MODEL == RPC) {
                const int w = p >> 6;
                const unsigned long long below = (1ull << (p & 63)) - 1ull;
                pi = base_i + ri[w] + __popcll(bi[w] & below);
                pj = base_j + rj[w] + __popcll(bj[w] & below);
                if constexpr (MODEL == RPC) { pi = c.oidx[pi]; pj = c.oidx[pj]; }  // the stored blocks are in observation order
            }
        }
        const double X = rc.r0.x, Y = rc.r0.y, Z = rc.r1.x;
        const double v00 = rc.r1.y, v01 = rc.r2.x, v02 = rc.r2.y, v11 = rc.r3.x, v12 = rc.r3.y, v22 = rc.r4;
        double Jci[2][NP], Jpi[2][3], Jcj[2][NP], Jpj[2][3];
        if constexpr (MODEL == RPC) {
            // the RPC chain costs 2-3 kflop per Jacobian and every observation sits in (track length - 1) pairs:
            // the blocks the linearize kernel stored are gathered instead of being recomputed
            const double2* qi = reinterpret_cast<const double2*>(a.Jpm + (size_t)pi * (2 * NP + 6));
            const double2* qj = reinterpret_cast<const double2*>(a.Jpm + (size_t)pj * (2 * NP + 6));
            double ti[2 * NP + 6], tj[2 * NP + 6];
#pragma unroll
            for (int k = 0; k < NP + 3; ++k) {
                const double2 vi = qi[k], vj = qj[k];
                ti[2 * k] = vi.x; ti[2 * k + 1] = vi.y; tj[2 * k] = vj.x; tj[2 * k + 1] = vj.y;
            }
#pragma unroll
            for (int k = 0; k < NP; ++k) { Jci[0][k] = ti[k]; Jci[1][k] = ti[NP + k]; Jcj[0][k] = tj[k]; Jcj[1][k] = tj[NP + k]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Jpi[0][k] = ti[2 * NP + k]; Jpi[1][k] = ti[2 * NP + 3 + k];
                Jpj[0][k] = tj[2 * NP + k]; Jpj[1][k] = tj[2 * NP + 3 + k];
            }
        } else if constexpr (SCL || (UNITW && !ROBUST)) {
            // unit weights with the linear loss, or stored row scales (SCL): the raw Jacobians.  The fixed-point mask and
            // the row scales multiply the 2 x 2 middle matrix (4 products instead of 12 + 20 on the blocks), the
            // fixed-camera masks are wave-uniform and multiply the reduced block once (cam_mask below)
            double u, v;
            project<MODEL, NP, true>(cci, tabi, X, Y, Z, a.f32 != 0, u, v, Jci, Jpi);
            project<MODEL, NP, true>(ccj, tabj, X, Y, Z, a.f32 != 0, u, v, Jcj, Jpj);
        } else {
            cm_jacobian<MODEL, NP, ROBUST, UNITW>(a, c, cci, tabi, i, pi, p, X, Y, Z, Jci, Jpi);
            cm_jacobian<MODEL, NP, ROBUST, UNITW>(a, c, ccj, tabj, j, pj, p, X, Y, Z, Jcj, Jpj);
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
        if constexpr ((SCL || (UNITW && !ROBUST)) && MODEL != RPC) {
            const double mp = (p >= a.n_pts_fix) ? 1.0 : 0.0;
            if constexpr (SCL) {
                // row scales s of the two observations (weights, robust loss): both blocks of an observation carry them, so the
                // pair block is Jc_i^T [diag(s_i^2) (Jp_i Vinv Jp_j^T) diag(s_j^2)] Jc_j -- four products on the 2 x 2
                // middle matrix instead of 32 on the Jacobian entries (and fewer live registers)
                const double ax = scl_i.x * scl_i.x * mp, ay = scl_i.y * scl_i.y * mp;
                const double bx = scl_j.x * scl_j.x, by = scl_j.y * scl_j.y;
                m00 *= ax * bx; m01 *= ax * by; m10 *= ay * bx; m11 *= ay * by;
            } else {
                m00 *= mp; m01 *= mp; m10 *= mp; m11 *= mp;
            }
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const double y0 = m00 * Jcj[0][q] + m01 * Jcj[1][q];
            const double y1 = m10 * Jcj[0][q] + m11 * Jcj[1][q];
#pragma unroll
            for (int r = 0; r < NP; ++r) acc[r][q] = fma(-Jci[0][r], y0, fma(-Jci[1][r], y1, acc[r][q]));
        }
    };
    auto process_point = [&](int p, bool valid) {
        if (!valid) return;
        const Rec rc = load_rec(p);
        compute(p, rc, -1, -1, make_double2(1.0, 1.0), make_double2(1.0, 1.0), false);
    };
    auto process = [&](int slot, bool valid) { process_point(valid ? s_q[wave][slot] : 0, valid); };

    if (s.pair_ofs) {
        // precomputed list of the points this camera pair shares (the structure is static across iterations): a
        // coalesced 4-byte stream replaces the bitmap scan (5 GB of bitmap traffic per launch at 200 x 1M)
        // pair_ofs[pair * (n_chunks + 1) + chunk]: start of the pair's points that fall into point-range chunk
        const int CF1 = s.n_chunks * s.chunk_mul + 1;  // this kernel works on groups of chunk_mul fine chunks
        const long long lo = s.pair_ofs[pair * CF1 + chunk * s.chunk_mul], hi = s.pair_ofs[pair * CF1 + (chunk + 1) * s.chunk_mul];
        // software pipeline: the next point's index and record are in flight while the current one is evaluated
        constexpr bool POS = !UNITW || MODEL == RPC;  // positions (and, weighted / robust, scales) ride along
        const bool listpos = POS && s.pair_pi != nullptr;
        long long idx = lo + lane;
        auto ld = [&](const int* arr, long long k) { return (k < hi) ? arr[k] : 0; };
        int p_cur = ld(s.pair_pts, idx), p_nxt = ld(s.pair_pts, idx + 64);
        int pi_cur = 0, pj_cur = 0, pi_nxt = 0, pj_nxt = 0;
        double2 si_cur = make_double2(1.0, 1.0), sj_cur = si_cur;
        if (listpos) {
            pi_cur = ld(s.pair_pi, idx); pj_cur = ld(s.pair_pj, idx);
            pi_nxt = ld(s.pair_pi, idx + 64); pj_nxt = ld(s.pair_pj, idx + 64);
            if constexpr (!UNITW) { si_cur = a.sc[pi_cur]; sj_cur = a.sc[pj_cur]; }
        }
        Rec r_cur = load_rec(p_cur);
        while (idx < hi) {
            // indices run two iterations ahead, records one: neither latency is on the critical path
            const int p_nn = ld(s.pair_pts, idx + 128);
            int pi_nn = 0, pj_nn = 0;
            double2 si_nxt = make_double2(1.0, 1.0), sj_nxt = si_nxt;
            if (listpos) {
                pi_nn = ld(s.pair_pi, idx + 128); pj_nn = ld(s.pair_pj, idx + 128);
                if constexpr (!UNITW) { si_nxt = a.sc[pi_nxt]; sj_nxt = a.sc[pj_nxt]; }
            }
            const Rec r_nxt = load_rec(p_nxt);
            // keep the gathers above the arithmetic: without the barrier the scheduler sinks them below compute() to
            // save registers and every iteration pays the full memory latency
            __builtin_amdgcn_sched_barrier(0);
            compute(p_cur, r_cur, pi_cur, pj_cur, si_cur, sj_cur, listpos);
            __builtin_amdgcn_sched_barrier(0);
            p_cur = p_nxt; p_nxt = p_nn; r_cur = r_nxt;
            pi_cur = pi_nxt; pj_cur = pj_nxt; pi_nxt = pi_nn; pj_nxt = pj_nn;
            si_cur = si_nxt; sj_cur = sj_nxt;
            idx += 64;
        }
    } else {
    const int w_lo = (int)((long long)s.NW * chunk / s.n_chunks), w_hi = (int)((long long)s.NW * (chunk + 1) / s.n_chunks);
    int n_q = 0;  // wave-uniform queue fill
    for (int w0 = w_lo; w0 < w_hi; w0 += 64) {
        const int w = w0 + lane;
        unsigned long long m = (w < w_hi) ? (bi[w] & bj[w]) : 0ull;
        // rounds: every lane with a remaining hit emits ONE entry; slots come from a ballot + popcount (no scan).
        // At most 64 entries are added per round and the queue is drained below 64 after each: no overflow.
        for (;;) {
            const unsigned long long has = __ballot(m != 0);
            if (has == 0) break;
            if (m) {
                const int off = n_q + __popcll(has & ((1ull << lane) - 1ull));
                const int bit = __ffsll((long long)m) - 1;
                s_q[wave][off] = w * 64 + bit;
                m &= m - 1;
            }
            n_q += __popcll(has);
            if (n_q >= 64) {  // evaluate a full wavefront of hits (the most recent 64 keep the queue compact)
                process(n_q - 64 + lane, true);
                n_q -= 64;
            }
        }
    }
    process(lane, lane < n_q);
    }

    // wave reduction of the NP x NP block; block (row j, col i) of the column-major lower triangle
    constexpr int NB2 = NP * NP;
    constexpr int NPAD = NB2 <= 16 ? 16 : (NB2 <= 32 ? 32 : 64);
    double flat[NPAD];
#pragma unroll
    for (int e = 0; e < NPAD; ++e) flat[e] = (e < NB2) ? acc[e / NP][e % NP] : 0.0;
    // camera masks of the unit-weight path (applied to every Jacobian by cm_jacobian on the other paths)
    const double cam_mask = ((SCL || (UNITW && !ROBUST)) && MODEL != RPC && (i < a.n_cam_fix || j < a.n_cam_fix)) ? 0.0 : 1.0;
    const double total = cam_mask * wave_reduce_scatter<NPAD>(flat, lane, 32);
    const int e = rs_index<NPAD>(lane);
    const bool writer = (lane & (64 / NPAD - 1)) == 0 && e < NB2;  // one lane per total (NPAD = 64: every lane)
    if (writer) {
        const int r = e / NP, q = e % NP;
        if (s.pair_ofs && s.n_chunks > 1) {
            s.pair_part[((size_t)chunk * n_pairs + pair) * NB2 + e] = total;
        } else {
            double* dst = S + (size_t)(j * NP + q) + (size_t)(i * NP + r) * a.n_c;  // S[(j,q), (i,r)] = (W_i Vinv W_j^T)[r][q]
            if (s.n_chunks > 1) atomicAdd(dst, total);
            else *dst = total;
        }
    }
}

template <int MODEL, int NP, bool ROBUST, bool UNITW, bool SCL = false>
__global__ __launch_bounds__(256) void k_schur_pairs(ObsArgs a, CamMajor c, Schur3Args s, double* __restrict__ S) {
    schur_pairs_body<MODEL, NP, ROBUST, UNITW, SCL>(a, c, s, S);
}
// ---- list path, lane-group form (experiment, SATBA_SCHUR_STREAM=1).  A wave owns up to 8 camera pairs
// (i, j0 .. j0+7) of ONE camera i and gives each pair 8 of its lanes for the whole launch: lane (g, r) walks hits
// r, r+8, ... of pair g.  Every wave of the grid is resident and all of them go through the point-range chunks in the
// same order and at the same pace, so the waves running at any time gather point records from one L2-sized window
// (PMC: L2 misses 37 M -> 10 M per launch with 3 MB windows), and because a lane never changes its pair, its 25
// accumulators survive from chunk to chunk: one 8-lane reduction per pair at the very end, no per-(pair, chunk)
// work items, no partial blocks, no reduce kernel.  Measured at 200 x 1M x 10M: 1.02 ms against 0.71 ms for the
// one-wave-per-item kernel above -- the memory side is solved, but the 8-lane lists leave 10-15 % of the lanes idle
// at every chunk end, the j constants come from LDS instead of SGPRs, and 2590 long-running waves do not pack the
// 2048 / 3072 wave slots (1.26 rounds at two waves per SIMD; at three the kernel got slower, not faster).  Kept
// as the starting point for the next attempt.
// grid: one wave per group; groups[g] = {i, j0, count}; S3_GW waves per workgroup.
constexpr int S3_MAXC = 64;  // chunks (upper bound)
constexpr int S3_GW = 4;     // waves per workgroup

// LP lanes per pair, PW = 64 / LP pairs per wave (LP = 6: 10 pairs on 60 lanes)
template <int MODEL, int NP, bool ROBUST, bool UNITW, int LP>
__device__ __forceinline__ void schur_pairs_groups_body(const ObsArgs& a, const CamMajor& c, const Schur3Args& s, const int* __restrict__ groups,
                                                         int n_groups, double* __restrict__ S) {
    constexpr int PW = 64 / LP;
    __shared__ long long s_ofs[S3_GW][PW][S3_MAXC + 1];
    __shared__ double s_cj[S3_GW][PW][CAMC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gidx = blockIdx.x * S3_GW + __builtin_amdgcn_readfirstlane(wave);
    if (gidx >= n_groups) return;
    const int i = groups[3 * gidx], j0 = groups[3 * gidx + 1], cnt = groups[3 * gidx + 2];  // wave-uniform (scalar loads)
    const int C = s.n_chunks * s.chunk_mul, C1 = C + 1;
    const long long pair0 = (long long)i * a.M - (long long)i * (i + 1) / 2 + (j0 - i - 1);
    for (int k = lane; k < cnt * C1; k += 64) s_ofs[wave][k / C1][k % C1] = s.pair_ofs[pair0 * C1 + k];
    for (int k = lane; k < cnt * CAMC; k += 64) s_cj[wave][k / CAMC][k % CAMC] = a.camc[(size_t)(j0 + k / CAMC) * CAMC + k % CAMC];
    __builtin_amdgcn_wave_barrier();  // single wave: its LDS operations execute in order
    const int g = lane / LP, r8 = lane % LP;
    const bool member = g < cnt;
    const int gq = member ? g : 0;
    const int j = j0 + gq;
    const double* cci = a.camc + (size_t)i * CAMC;
    const double* ccj = &s_cj[wave][gq][0];
    const double* tabi = (MODEL == RPC) ? a.rpc + (size_t)i * 90 : nullptr;
    const double* tabj = (MODEL == RPC) ? a.rpc + (size_t)j * 90 : nullptr;
    const unsigned long long* bi = s.bits + (size_t)i * s.NW;
    const unsigned long long* bj = s.bits + (size_t)j * s.NW;
    const int* ri = s.rank + (size_t)i * s.NW;
    const int* rj = s.rank + (size_t)j * s.NW;
    int base_i = 0, base_j = 0;
    if constexpr (!UNITW) { base_i = c.cam_ofs[i]; base_j = c.cam_ofs[j]; }

    double acc[NP][NP];
#pragma unroll
    for (int r = 0; r < NP; ++r)
#pragma unroll
        for (int q = 0; q < NP; ++q) acc[r][q] = 0.0;

    struct Rec { double2 r0, r1, r2, r3, r4; };  // packed point record: X0 X1 | X2 v00 | v01 v02 | v11 v12 | v22 g0
    auto load_rec = [&](int p) {
        const double2* pv = s.PV + (PV_STRIDE / 2) * (size_t)p;
        Rec r;
        r.r0 = pv[0]; r.r1 = pv[1]; r.r2 = pv[2]; r.r3 = pv[3]; r.r4 = pv[4];
        return r;
    };
    auto compute = [&](int p, const Rec& rc) {
        int pi = 0, pj = 0;
        if constexpr (!UNITW) {
            const int w = p >> 6;
            const unsigned long long below = (1ull << (p & 63)) - 1ull;
            pi = base_i + ri[w] + __popcll(bi[w] & below);
            pj = base_j + rj[w] + __popcll(bj[w] & below);
        }
        const double X = rc.r0.x, Y = rc.r0.y, Z = rc.r1.x;
        const double v00 = rc.r1.y, v01 = rc.r2.x, v02 = rc.r2.y, v11 = rc.r3.x, v12 = rc.r3.y, v22 = rc.r4.x;
        double Jci[2][NP], Jpi[2][3], Jcj[2][NP], Jpj[2][3];
        cm_jacobian<MODEL, NP, ROBUST, UNITW>(a, c, cci, tabi, i, pi, p, X, Y, Z, Jci, Jpi);
        cm_jacobian<MODEL, NP, ROBUST, UNITW>(a, c, ccj, tabj, j, pj, p, X, Y, Z, Jcj, Jpj);
        double A[2][3];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            A[r][0] = Jpi[r][0] * v00 + Jpi[r][1] * v01 + Jpi[r][2] * v02;
            A[r][1] = Jpi[r][0] * v01 + Jpi[r][1] * v11 + Jpi[r][2] * v12;
            A[r][2] = Jpi[r][0] * v02 + Jpi[r][1] * v12 + Jpi[r][2] * v22;
        }
        const double m00 = A[0][0] * Jpj[0][0] + A[0][1] * Jpj[0][1] + A[0][2] * Jpj[0][2];
        const double m01 = A[0][0] * Jpj[1][0] + A[0][1] * Jpj[1][1] + A[0][2] * Jpj[1][2];
        const double m10 = A[1][0] * Jpj[0][0] + A[1][1] * Jpj[0][1] + A[1][2] * Jpj[0][2];
        const double m11 = A[1][0] * Jpj[1][0] + A[1][1] * Jpj[1][1] + A[1][2] * Jpj[1][2];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const double y0 = m00 * Jcj[0][q] + m01 * Jcj[1][q];
            const double y1 = m10 * Jcj[0][q] + m11 * Jcj[1][q];
#pragma unroll
            for (int r = 0; r < NP; ++r) acc[r][q] = fma(-Jci[0][r], y0, fma(-Jci[1][r], y1, acc[r][q]));
        }
    };

    for (int ch = 0; ch < C; ++ch) {
        // this lane's hits of the chunk: pos, pos + 8, ... < hi; indices two steps ahead, records one step ahead
        long long pos = s_ofs[wave][gq][ch] + r8;
        const long long hi = member ? s_ofs[wave][gq][ch + 1] : 0;
        int p_cur = (pos < hi) ? s.pair_pts[pos] : 0;
        int p_nxt = (pos + LP < hi) ? s.pair_pts[pos + LP] : 0;
        Rec r_cur = load_rec(p_cur);
        while (__any(pos < hi)) {
            const int p_nn = (pos + 2 * LP < hi) ? s.pair_pts[pos + 2 * LP] : 0;
            const Rec r_nxt = load_rec(p_nxt);
            // keep the gathers above the arithmetic: without the barrier the scheduler sinks them below compute()
            // to save registers and every step pays the full memory latency
            __builtin_amdgcn_sched_barrier(0);
            {   // lanes past the end of their list evaluate point 0 with Vinv = 0: a zero contribution without a
                // divergent branch around the accumulators (the branch costs 50 register copies per step)
                Rec rc = r_cur;
                if (!(pos < hi)) { rc.r1.y = 0.0; rc.r2 = make_double2(0.0, 0.0); rc.r3 = make_double2(0.0, 0.0); rc.r4.x = 0.0; }
                compute(p_cur, rc);
            }
            __builtin_amdgcn_sched_barrier(0);
            p_cur = p_nxt; p_nxt = p_nn; r_cur = r_nxt;
            pos += LP;
        }
    }
    // LP-lane reduction per pair; block (row j, col i) of the column-major lower triangle
#pragma unroll
    for (int r = 0; r < NP; ++r)
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            double t = acc[r][q];
#pragma unroll
            for (int d = 1; d < LP; ++d) t += __shfl(acc[r][q], min(lane + d, 63));  // only the sums of the lanes r8 == 0 are used
            if (member && r8 == 0) S[(size_t)(j * NP + q) + (size_t)(i * NP + r) * a.n_c] = t;  // S[(j,q), (i,r)] = (W_i Vinv W_j^T)[r][q]
        }
}

template <int MODEL, int NP, bool ROBUST, bool UNITW>
__global__ __launch_bounds__(64 * S3_GW) void k_schur_pairs_groups(ObsArgs a, CamMajor c, Schur3Args s, const int* __restrict__ groups,
                                                                   int n_groups, double* __restrict__ S) {
    schur_pairs_groups_body<MODEL, NP, ROBUST, UNITW, 8>(a, c, s, groups, n_groups, S);
}
// the same held to three waves per SIMD (<= 168 VGPRs; the affine unit-weight body needs 170): all M(M-1)/16 waves of
// the headline shape are then resident at once -- one round instead of 1.26
template <int MODEL, int NP, bool ROBUST, bool UNITW>
__global__ __launch_bounds__(64 * S3_GW) __attribute__((amdgpu_waves_per_eu(3, 8))) void k_schur_pairs_groups_occ3(
    ObsArgs a, CamMajor c, Schur3Args s, const int* __restrict__ groups, int n_groups, double* __restrict__ S) {
    schur_pairs_groups_body<MODEL, NP, ROBUST, UNITW, 8>(a, c, s, groups, n_groups, S);
}
// six lanes per pair, ten pairs per wave: M(M-1)/20 waves fit the 2048 slots of two waves per SIMD in one round
template <int MODEL, int NP, bool ROBUST, bool UNITW>
__global__ __launch_bounds__(64 * S3_GW) void k_schur_pairs_groups6(ObsArgs a, CamMajor c, Schur3Args s, const int* __restrict__ groups,
                                                                    int n_groups, double* __restrict__ S) {
    schur_pairs_groups_body<MODEL, NP, ROBUST, UNITW, 6>(a, c, s, groups, n_groups, S);
}

// ---- affine cameras, unit weights, linear loss: the pair blocks through MOMENTS of the shared points
// (experiment, SATBA_SCHUR_MOMENTS=1).
// For an affine camera the point Jacobian A_i = dq/dX is a constant 2 x 3 matrix and every column of the camera
// Jacobian is affine-linear in the point, Jc_i(X)[:, c] = E_ic x~ with x~ = (X - c0, 1) and E_ic a constant 2 x 4
// matrix.  Therefore
//   block_ij[r][q] = - sum_p Jc_i[:, r]^T (A_i Vinv_p A_j^T) Jc_j[:, q]
//                  = - sum_{a,b,m,n} (E_ir^T A_i)[a][m]  T_ij[a][b][m][n]  (A_j^T E_jq)[n][b],
//   T_ij[a][b][m][n] = sum_{p in i and j} x~_a x~_b Vinv_p[m][n]   (10 x 6 independent entries),
// and the per-hit work drops from ~200 fp64 operations (two Jacobians, a 2 x 2 product, a 5 x 5 rank-2 update) to
// 6 multiplications and 60 multiply-adds on data that needs no camera at all; the contraction with the cameras
// happens once per pair (k_schur_contract).  Same lane-group layout as above (LP lanes per pair, accumulators live for
// the whole launch).  Measured at 200 x 1M x 10M: exact (same solver trajectory), VALU instructions 277 M -> 98 M per
// launch, but 1.0 ms against 0.71 ms: the waves do not stay in lockstep over the chunks (L2 misses 32 M per launch,
// not the 10 M of a 3 MB window) and with two waves per SIMD nothing hides the gather latency.  The arithmetic is the
// right one; the traversal needs a (soft) chunk barrier or the item kernel's dynamic scheduling -- next round.
constexpr int S3_NT = 60;

template <int LP>
__global__ __launch_bounds__(64 * S3_GW) void k_schur_pairs_moments(int M, int n_pts_fix, Schur3Args s, const int* __restrict__ groups,
                                                                    int n_groups, double c0x, double c0y, double c0z,
                                                                    double* __restrict__ Tbuf) {
    constexpr int PW = 64 / LP;
    __shared__ long long s_ofs[S3_GW][PW][S3_MAXC + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gidx = blockIdx.x * S3_GW + __builtin_amdgcn_readfirstlane(wave);
    if (gidx >= n_groups) return;
    const int i = groups[3 * gidx], j0 = groups[3 * gidx + 1], cnt = groups[3 * gidx + 2];
    const int C = s.n_chunks * s.chunk_mul, C1 = C + 1;
    const long long pair0 = (long long)i * M - (long long)i * (i + 1) / 2 + (j0 - i - 1);
    for (int k = lane; k < cnt * C1; k += 64) s_ofs[wave][k / C1][k % C1] = s.pair_ofs[pair0 * C1 + k];
    __builtin_amdgcn_wave_barrier();  // single wave: its LDS operations execute in order
    const int g = lane / LP, r8 = lane % LP;
    const bool member = g < cnt;
    const int gq = member ? g : 0;

    double t[S3_NT];
#pragma unroll
    for (int k = 0; k < S3_NT; ++k) t[k] = 0.0;

    struct Rec { double2 r0, r1, r2, r3, r4; };  // packed point record: X0 X1 | X2 v00 | v01 v02 | v11 v12 | v22 g0
    auto load_rec = [&](int p) {
        const double2* pv = s.PV + (PV_STRIDE / 2) * (size_t)p;
        Rec r;
        r.r0 = pv[0]; r.r1 = pv[1]; r.r2 = pv[2]; r.r3 = pv[3]; r.r4 = pv[4];
        return r;
    };
    // One loop level carries the accumulators; moving on to the next chunk with work is a small loop of its own at
    // the end of a step (a while loop per chunk makes the compiler keep two copies of the 60 accumulators).
    int ch = -1;
    long long pos = 0, hi = 0;
    int p_cur = 0, p_nxt = 0;
    Rec r_cur = load_rec(0);
    auto next_chunk = [&]() {  // false: no chunk left
        do {
            if (++ch >= C) return false;
            pos = s_ofs[wave][gq][ch] + r8;
            hi = member ? s_ofs[wave][gq][ch + 1] : 0;
        } while (!__any(pos < hi));
        p_cur = (pos < hi) ? s.pair_pts[pos] : 0;
        p_nxt = (pos + LP < hi) ? s.pair_pts[pos + LP] : 0;
        r_cur = load_rec(p_cur);
        return true;
    };
    bool more = next_chunk();
    while (more) {
        const int p_nn = (pos + 2 * LP < hi) ? s.pair_pts[pos + 2 * LP] : 0;
        const Rec r_nxt = load_rec(p_nxt);
        __builtin_amdgcn_sched_barrier(0);  // the gathers stay above the arithmetic
        {
            // lanes past the end of their list, and fixed points (their Jacobian is masked), contribute zero
            const double on = (pos < hi && p_cur >= n_pts_fix) ? 1.0 : 0.0;
            const double x0 = r_cur.r0.x - c0x, x1 = r_cur.r0.y - c0y, x2 = r_cur.r1.x - c0z;
            const double v[6] = {on * r_cur.r1.y, on * r_cur.r2.x, on * r_cur.r2.y, on * r_cur.r3.x, on * r_cur.r3.y, on * r_cur.r4.x};
            const double xx[10] = {x0 * x0, x0 * x1, x0 * x2, x0, x1 * x1, x1 * x2, x1, x2 * x2, x2, 1.0};
#pragma unroll
            for (int k = 0; k < 10; ++k)
#pragma unroll
                for (int m = 0; m < 6; ++m) t[k * 6 + m] = (k == 9) ? t[k * 6 + m] + v[m] : fma(xx[k], v[m], t[k * 6 + m]);
        }
        __builtin_amdgcn_sched_barrier(0);
        p_cur = p_nxt; p_nxt = p_nn; r_cur = r_nxt;
        pos += LP;
        if (!__any(pos < hi)) more = next_chunk();
    }
    // LP-lane reduction per pair
#pragma unroll
    for (int k = 0; k < S3_NT; ++k) {
        double sum = t[k];
#pragma unroll
        for (int d = 1; d < LP; ++d) sum += __shfl(t[k], min(lane + d, 63));  // only the sums of the lanes r8 == 0 are used
        if (member && r8 == 0) Tbuf[(size_t)(pair0 + g) * S3_NT + k] = sum;
        __builtin_amdgcn_sched_barrier(0);  // one value at a time: 300 shuffles in flight would cost the main loop its registers
    }
}

// contraction of the pair moments with the two cameras: one thread per (pair, r, q); the constant matrices of a camera
// come out of the projector itself (it is exactly affine in the point): columns at the origin and at h e_a give the
// linear part, the column at c0 the constant part.
template <int NP>
__global__ __launch_bounds__(256) void k_schur_contract(ObsArgs a, const double* __restrict__ Tbuf, double c0x, double c0y, double c0z,
                                                        double* __restrict__ S) {
    const long long n_pairs = (long long)a.M * (a.M - 1) / 2;
    constexpr int NB2 = NP * NP;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pairs * NB2) return;
    const long long pair = idx / NB2;
    const int e = (int)(idx % NB2), r = e / NP, q = e % NP;
    int i = (int)((2.0 * a.M - 1.0 - sqrt((2.0 * a.M - 1.0) * (2.0 * a.M - 1.0) - 8.0 * (double)pair)) * 0.5);
    while ((long long)i * a.M - (long long)i * (i + 1) / 2 > pair) --i;
    while ((long long)(i + 1) * a.M - (long long)(i + 1) * (i + 2) / 2 <= pair) ++i;
    const int j = i + 1 + (int)(pair - ((long long)i * a.M - (long long)i * (i + 1) / 2));
    // B[al][m] = sum_k E[k][al] A[k][m] for parameter `col` of camera `cam`
    auto cam_matrix = [&](int cam, int col, double B[4][3]) {
        const double* cc = a.camc + (size_t)cam * CAMC;
        const double h = 1048576.0;  // 2^20 m: exact scaling, keeps the difference quotient free of cancellation
        double u, v, Jc0[2][NP], Jc1[2][NP], Jp[2][3], E[2][4];
        project<AFFINE, NP, true>(cc, nullptr, 0.0, 0.0, 0.0, false, u, v, Jc0, Jp);
#pragma unroll
        for (int al = 0; al < 3; ++al) {
            project<AFFINE, NP, true>(cc, nullptr, al == 0 ? h : 0.0, al == 1 ? h : 0.0, al == 2 ? h : 0.0, false, u, v, Jc1, Jp);
            E[0][al] = (Jc1[0][col] - Jc0[0][col]) / h;
            E[1][al] = (Jc1[1][col] - Jc0[1][col]) / h;
        }
        project<AFFINE, NP, true>(cc, nullptr, c0x, c0y, c0z, false, u, v, Jc1, Jp);
        E[0][3] = Jc1[0][col];
        E[1][3] = Jc1[1][col];
        const double mc = (cam >= a.n_cam_fix) ? 1.0 : 0.0;
#pragma unroll
        for (int al = 0; al < 4; ++al)
#pragma unroll
            for (int m = 0; m < 3; ++m) B[al][m] = mc * (E[0][al] * Jp[0][m] + E[1][al] * Jp[1][m]);
    };
    double Bi[4][3], Bj[4][3];
    cam_matrix(i, r, Bi);
    cam_matrix(j, q, Bj);
    const double* T = Tbuf + (size_t)pair * S3_NT;
    const int XI[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}};
    const int VI[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    double sum = 0.0;
#pragma unroll
    for (int al = 0; al < 4; ++al)
#pragma unroll
        for (int be = 0; be < 4; ++be)
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) sum += Bi[al][m] * T[XI[al][be] * 6 + VI[m][n]] * Bj[be][n];
    S[(size_t)(j * NP + q) + (size_t)(i * NP + r) * a.n_c] = -sum;
}

// list path with several point-range chunks: S block of each pair = sum of its chunk partials
__global__ __launch_bounds__(256) void k_schur_pairs_reduce(int M, int NP, int n_c, int n_chunks, const double* __restrict__ part,
                                                            double* __restrict__ S) {
    const long long n_pairs = (long long)M * (M - 1) / 2;
    const int NB2 = NP * NP;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pairs * NB2) return;
    const long long pair = idx / NB2;
    const int e = (int)(idx % NB2);
    double t = 0.0;
    for (int ch = 0; ch < n_chunks; ++ch) t += part[((size_t)ch * n_pairs + pair) * NB2 + e];
    int i = (int)((2.0 * M - 1.0 - sqrt((2.0 * M - 1.0) * (2.0 * M - 1.0) - 8.0 * (double)pair)) * 0.5);
    while ((long long)i * M - (long long)i * (i + 1) / 2 > pair) --i;
    while ((long long)(i + 1) * M - (long long)(i + 1) * (i + 2) / 2 <= pair) ++i;
    const int j = i + 1 + (int)(pair - ((long long)i * M - (long long)i * (i + 1) / 2));
    const int r = e / NP, q = e % NP;
    S[(size_t)(j * NP + q) + (size_t)(i * NP + r) * n_c] = t;
}

// Diagonal blocks and right-hand side: camera-major pass, registers only.
//   S_ii += sum_p (Jc^T Jc [ADDU] - W_ip Vinv W_ip^T),   rhs_i -= sum_p W_ip Vinv g_p.   grid (chunks, M); part [M][chunks][CU]
// ADDU: also add J_c^T J_c (the U_c block), which the linearize kernel then does not have to accumulate.
template <int MODEL, int NP, bool ROBUST, bool ADDU>
__global__ __launch_bounds__(LINC_THREADS) void k_schur_diag(ObsArgs a, CamMajor c, Schur3Args s, double* __restrict__ part) {
    constexpr int CU = cam_acc_len(NP);
    // grid (M, chunks): the workgroups of one chunk (the same slice of every camera's point-sorted list, i.e. about the same
    // point range) are dispatched together and share their point records in L2
    const int cam = blockIdx.x, chunk = blockIdx.y, n_chunks = gridDim.y;
    const int b = c.cam_ofs[cam], e = c.cam_ofs[cam + 1];
    const long long len = e - b;
    const int lo = b + (int)(len * chunk / n_chunks), hi = b + (int)(len * (chunk + 1) / n_chunks);
    const double* cc = a.camc + (size_t)cam * CAMC;
    const double* tab = (MODEL == RPC) ? a.rpc + (size_t)cam * 90 : nullptr;
    double acc[CU];
#pragma unroll
    for (int k = 0; k < CU; ++k) acc[k] = 0.0;
    // Software pipeline: point indices run two iterations ahead, the 96-byte point records one (the loop was a chain
    // of two dependent gathers per iteration, ~18 iterations per thread).  unit: every weight is 1 and the loss is
    // linear -- nothing is fetched per observation, the fixed-point mask multiplies the 2 x 2 middle matrix and the
    // (block-uniform) fixed-camera mask the accumulators at the end.
    struct Rec { double2 r0, r1, r2, r3, r4, r5; };
    auto load_rec = [&](int p) {
        const double2* pv = s.PV + (PV_STRIDE / 2) * (size_t)p;
        Rec r;
        r.r0 = pv[0]; r.r1 = pv[1]; r.r2 = pv[2]; r.r3 = pv[3]; r.r4 = pv[4]; r.r5 = pv[5];
        return r;
    };
    const bool unit = !ROBUST && MODEL != RPC && a.unit != 0;
    auto ldp = [&](int q) { return q < hi ? c.pt[q] : 0; };
    int pos = lo + threadIdx.x;
    int p_cur = ldp(pos), p_nxt = ldp(pos + LINC_THREADS);
    Rec rc = load_rec(p_cur);
    for (; pos < hi; pos += LINC_THREADS) {
        const int p_nn = ldp(pos + 2 * LINC_THREADS);
        const Rec rn = load_rec(p_nxt);
        __builtin_amdgcn_sched_barrier(0);
        const int p = p_cur;
        const double2 r0 = rc.r0, r1 = rc.r1, r2 = rc.r2, r3 = rc.r3, r4 = rc.r4, r5 = rc.r5;
        p_cur = p_nxt; p_nxt = p_nn; rc = rn;
        double Jc[2][NP], Jp[2][3];
        bool have = false;
        double mp = 1.0;
        if constexpr (MODEL == RPC) {
            if (a.Jpm) {  // the blocks the linearize kernel stored, through the camera-major permutation
                const double2* q = reinterpret_cast<const double2*>(a.Jpm + (size_t)c.oidx[pos] * (2 * NP + 6));
                double t[2 * NP + 6];
#pragma unroll
                for (int k = 0; k < NP + 3; ++k) { const double2 v = q[k]; t[2 * k] = v.x; t[2 * k + 1] = v.y; }
#pragma unroll
                for (int k = 0; k < NP; ++k) { Jc[0][k] = t[k]; Jc[1][k] = t[NP + k]; }
#pragma unroll
                for (int k = 0; k < 3; ++k) { Jp[0][k] = t[2 * NP + k]; Jp[1][k] = t[2 * NP + 3 + k]; }
                have = true;
            }
        }
        if (!have) {
            if (unit) {
                double u, v;
                project<MODEL, NP, true>(cc, tab, r0.x, r0.y, r1.x, a.f32 != 0, u, v, Jc, Jp);
                mp = (p >= a.n_pts_fix) ? 1.0 : 0.0;
            } else {
                cm_jacobian<MODEL, NP, ROBUST>(a, c, cc, tab, cam, pos, p, r0.x, r0.y, r1.x, Jc, Jp);
            }
        }
        const double v00 = r1.y, v01 = r2.x, v02 = r2.y, v11 = r3.x, v12 = r3.y, v22 = r4.x;
        double A[2][3];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            A[r][0] = Jp[r][0] * v00 + Jp[r][1] * v01 + Jp[r][2] * v02;
            A[r][1] = Jp[r][0] * v01 + Jp[r][1] * v11 + Jp[r][2] * v12;
            A[r][2] = Jp[r][0] * v02 + Jp[r][1] * v12 + Jp[r][2] * v22;
        }
        const double m00 = mp * (A[0][0] * Jp[0][0] + A[0][1] * Jp[0][1] + A[0][2] * Jp[0][2]);
        const double m01 = mp * (A[0][0] * Jp[1][0] + A[0][1] * Jp[1][1] + A[0][2] * Jp[1][2]);
        const double m11 = mp * (A[1][0] * Jp[1][0] + A[1][1] * Jp[1][1] + A[1][2] * Jp[1][2]);
        const double ag0 = mp * (A[0][0] * r4.y + A[0][1] * r5.x + A[0][2] * r5.y);
        const double ag1 = mp * (A[1][0] * r4.y + A[1][1] * r5.x + A[1][2] * r5.y);
        int k = 0;
#pragma unroll
        for (int r = 0; r < NP; ++r) {
            const double y0 = m00 * Jc[0][r] + m01 * Jc[1][r];
            const double y1 = m01 * Jc[0][r] + m11 * Jc[1][r];
#pragma unroll
            for (int q = r; q < NP; ++q) {
                acc[k] -= Jc[0][q] * y0 + Jc[1][q] * y1;
                if (ADDU) acc[k] += Jc[0][r] * Jc[0][q] + Jc[1][r] * Jc[1][q];
                ++k;
            }
        }
#pragma unroll
        for (int r = 0; r < NP; ++r) acc[k++] -= Jc[0][r] * ag0 + Jc[1][r] * ag1;
        __builtin_amdgcn_sched_barrier(0);
    }
    if (unit && cam < a.n_cam_fix) {
#pragma unroll
        for (int k = 0; k < CU; ++k) acc[k] = 0.0;
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

// S_ii (lower incl. diagonal, both triangles of the block are written) and rhs_i += chunk partials
__global__ void k_schur_diag_finish(int M, int NP, int n_c, int n_chunks, const double* __restrict__ part, double* __restrict__ S,
                                    double* __restrict__ rhs) {
    const int CU = cam_acc_len(NP);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * CU) return;
    const int cam = idx / CU, k = idx % CU;
    double t = 0.0;
    for (int ch = 0; ch < n_chunks; ++ch) t += part[((size_t)cam * n_chunks + ch) * CU + k];
    const int ntri = NP * (NP + 1) / 2;
    if (k >= ntri) {
        rhs[cam * NP + (k - ntri)] += t;
        return;
    }
    int r = 0, rem = k;
    while (rem >= NP - r) { rem -= NP - r; ++r; }
    const int q = r + rem;
    S[(size_t)(cam * NP + q) + (size_t)(cam * NP + r) * n_c] += t;
    if (q != r) S[(size_t)(cam * NP + r) + (size_t)(cam * NP + q) * n_c] += t;
}

}  // namespace satba
