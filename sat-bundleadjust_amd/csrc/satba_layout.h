// satba_layout.h -- index structures of one shard, built ON THE DEVICE from the caller's (pts_ind, cam_ind) lists
// (what ref:bundle_adjust/ba_params.py:139-148 produces: point-major observation lists, cameras ascending inside a point).
//
// Everything the kernels index with is derived here with sort / scan kernels (hipCUB radix sort and prefix sums plus a
// few fill kernels); the host only sizes the allocations.  tests/test_gpu_layout.py checks every array bit for bit
// against a numpy restatement.
//
//   points     internal order = stable sort by track length (number of observations), ascending.
//              perm[q] = caller's local point index of internal point q, rank[] its inverse, pt_cnt[q] the length.
//   sliced ELL 64 consecutive internal points form a slice (one wavefront, lane = point).  Slot k of all 64 points
//              is stored contiguously, so "observation k of my point" is a coalesced access for the wave:
//                  pos(q, k) = slice_base[q / 64] + 64 k + q % 64,       k < pt_cnt[q]
//              A slice is as long as its longest track; because the points are sorted by length the padding is a few
//              slots at the boundaries between length classes (e_cam = -1 there).  Every per-observation array of the
//              solver (residuals, row scales, stored RPC Jacobian blocks) uses these positions.
//   io         internal point-major observation index io(q, k) = ipt_ofs[q] + k: the observations of a point are neighbours.
//              Per-observation data that the Schur kernels GATHER (Jacobian row scales, stored RPC blocks) is kept in this
//              order, so that the two observations of a (camera pair, point) hit share cache lines.
//   cameras    cam_ofs[M + 1], cm_pt[K], cm_pos[K], cm_io[K]: the observations of every camera, internal point ascending
//              (camera-major passes: diagonal Schur blocks, deterministic camera sums).
//   pairs      for every camera pair (i < j) the points both see, ascending, cut into C point-range chunks:
//              pair_ofs[pair (C + 1) + chunk], pair_pts[E], pair_pi[E], pair_pj[E] (io indices of the two
//              observations), pair_ij[pair] = (i, j).
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

namespace satba {

struct Layout {
    int M = 0, N = 0, n_slices = 0, P = 0;  // P: padded ELL length
    long long K = 0, E = 0;                 // observations, pair-list entries
    long long n_pairs = 0;
    int C = 1;                              // point-range chunks of the pair lists
    int *perm = nullptr, *rank = nullptr, *pt_cnt = nullptr, *slice_base = nullptr;
    int* e_cam = nullptr;
    double2* e_obs = nullptr;
    double* e_w = nullptr;
    int *obs_pos = nullptr, *pts_ind = nullptr;
    int* ipt_ofs = nullptr;  // N + 1: internal point-major observation index ("io") of every point's first observation
    int *cam_ofs = nullptr, *cm_pt = nullptr, *cm_pos = nullptr, *cm_io = nullptr;
    long long* pair_ofs = nullptr;
    int2* pair_ij = nullptr;
    int *pair_pts = nullptr, *pair_pi = nullptr, *pair_pj = nullptr;
    long long* hit_ofs = nullptr;  // N + 1: running count of the k (k - 1) / 2 pair entries of the points before q
    // Weighted / robust runs of the affine and perspective models (round 5, built on first use: wl_ready).  The point record of the
    // Schur kernels and the Jacobian row scales of the point's observations share ONE variable-length record in the buffer W of
    // 16-byte pieces, ending on a 128-byte line:
    //     [ pad | s_0 s_1 .. s_{c-1} | X0 X1 | X2 v00 | v01 v02 | v11 v12 | v22 g0 | g1 g2 ]       c = track length
    // so that a visit of camera (track position k) reads one contiguous run s_k .. g: ceil((c - k + 6) / 8) lines, 1.9 on
    // average at ten observations per point, where the separate record (1 line) and io-ordered scale array (1.7) made 2.7.
    // w_fix[q]: piece of X0 (w_fix[N]: an all-zero record); sc_ofs[q] = w_fix[q] - c: piece of s_0 (replaces ipt_ofs in the
    // kernels that walk a point's observations); pair_rec / pair_kk: per pair-list entry w_fix of the point and the distances
    // (c - k_i) | (c - k_j) << 16 of its two scales in front of it; cm_rec / cm_sc: per camera-major entry w_fix of the point and
    // the piece of the observation's scale; dg_ofs: the camera-major lists cut into the diagonal items of k_schur_pairs,
    // [(camera n_dg + chunk dg_spc + sub)], n_dg = C dg_spc per camera.
    bool wl_ready = false;
    long long W_len = 0;     // pieces, the zero record included
    int zero_fix = 0, dg_spc = 1, n_dg = 1;
    int *w_fix = nullptr, *sc_ofs = nullptr, *pair_rec = nullptr, *pair_kk = nullptr, *cm_rec = nullptr, *cm_sc = nullptr, *dg_ofs = nullptr;
};

// error word: 0 ok; otherwise (code << 32 | index of the first offender + 1 is not tracked: code only)
enum { LAY_OK = 0, LAY_E_CAM = 1, LAY_E_PT = 2, LAY_E_ORDER = 3, LAY_E_CAM_ORDER = 4 };

// indices in range, points non-decreasing, cameras strictly ascending inside a point; flags[0] = first error code,
// flags[1] = 1 if some weight differs from 1, flags[2..3] = bit pattern of the largest |weight| (a double; the bit patterns of
// non-negative doubles order like integers)
__global__ void k_lay_validate(long long K, int M, int N, const int* __restrict__ cam, const int* __restrict__ pt,
                               const double* __restrict__ w, int* __restrict__ flags) {
    double wm = 0.0;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < K; o += (long long)gridDim.x * blockDim.x) {
        wm = fmax(wm, fabs(w[o]));
        const int c = cam[o], p = pt[o];
        int err = 0;
        if (c < 0 || c >= M) err = LAY_E_CAM;
        else if (p < 0 || p >= N) err = LAY_E_PT;
        else if (o > 0) {
            const int pp = pt[o - 1];
            if (pp > p) err = LAY_E_ORDER;
            else if (pp == p && cam[o - 1] >= c) err = LAY_E_CAM_ORDER;
        }
        if (err) atomicCAS(flags, 0, err);
        if (w[o] != 1.0) flags[1] = 1;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wm = fmax(wm, __shfl_down(wm, d));
    if ((threadIdx.x & 63) == 0 && wm > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(flags + 2), (unsigned long long)__double_as_longlong(wm));
}

// CSR offsets of a sorted key list: ofs[v] = first index whose key is >= v, for v = 0 .. n_keys (ofs[n_keys] = K)
template <class T>
__global__ void k_lay_offsets(long long K, int n_keys, const int* __restrict__ key, T* __restrict__ ofs) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o <= K; o += (long long)gridDim.x * blockDim.x) {
        const int lo = (o == 0) ? 0 : key[o - 1] + 1;
        const int hi = (o == K) ? n_keys : key[o];
        for (int v = lo; v <= hi; ++v) ofs[v] = (T)o;
    }
}
// the same for 64-bit composite keys that are given by a functor of the index
__global__ void k_lay_offsets_pair(long long E, long long n_keys, const int* __restrict__ pair_sorted, const int* __restrict__ pts,
                                   int N, int C, long long* __restrict__ ofs) {
    auto key = [&](long long r) { return (long long)pair_sorted[r] * (C + 1) + (int)((long long)pts[r] * C / N); };
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r <= E; r += (long long)gridDim.x * blockDim.x) {
        const long long lo = (r == 0) ? 0 : key(r - 1) + 1;
        const long long hi = (r == E) ? n_keys : key(r);
        for (long long v = lo; v <= hi; ++v) ofs[v] = r;
    }
}

__global__ void k_lay_counts(int N, const int* __restrict__ ofs, int* __restrict__ cnt, int* __restrict__ iota) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < N) { cnt[p] = ofs[p + 1] - ofs[p]; iota[p] = p; }
}

__global__ void k_lay_iota(long long n, int* __restrict__ iota) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) iota[i] = (int)i;
}

// rank = perm^-1; slice lengths (the last point of a slice has its longest track); pair-entry counts per point
__global__ void k_lay_rank(int N, const int* __restrict__ perm, const int* __restrict__ cnt_sorted, int* __restrict__ rank,
                           int* __restrict__ slice_slots, long long* __restrict__ hits) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= N) return;
    rank[perm[q]] = q;
    const long long c = cnt_sorted[q];
    hits[q] = c * (c - 1) / 2;
    if ((q & 63) == 63 || q == N - 1) slice_slots[q >> 6] = 64 * (int)c;
}

// scatter the caller's observation arrays into ELL order; internal point-major list (cam, pos, point) for the camera sort
__global__ void k_lay_fill_ell(long long K, const int* __restrict__ cam, const int* __restrict__ pt, const double2* __restrict__ obs,
                               const double* __restrict__ w, const int* __restrict__ ofs, const int* __restrict__ rank,
                               const int* __restrict__ slice_base, const int* __restrict__ ipt_ofs, int* __restrict__ e_cam,
                               double2* __restrict__ e_obs, double* __restrict__ e_w, int* __restrict__ obs_pos,
                               int* __restrict__ io_cam, int* __restrict__ io_pos, int* __restrict__ io_pt) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < K; o += (long long)gridDim.x * blockDim.x) {
        const int p = pt[o], q = rank[p], k = (int)(o - ofs[p]);
        const int pos = slice_base[q >> 6] + 64 * k + (q & 63);
        const int c = cam[o];
        e_cam[pos] = c;
        e_obs[pos] = obs[o];
        e_w[pos] = w[o];
        obs_pos[o] = pos;
        const int io = ipt_ofs[q] + k;
        io_cam[io] = c; io_pos[io] = pos; io_pt[io] = q;
    }
}

__global__ void k_lay_gather2(long long n, const int* __restrict__ idx, const int* __restrict__ a, const int* __restrict__ b,
                              int* __restrict__ oa, int* __restrict__ ob) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = idx[i];
        oa[i] = a[j]; ob[i] = b[j];
    }
}
__global__ void k_lay_gather3(long long n, const int* __restrict__ idx, const int* __restrict__ a, const int* __restrict__ b,
                              const int* __restrict__ c, int* __restrict__ oa, int* __restrict__ ob, int* __restrict__ oc) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = idx[i];
        oa[i] = a[j]; ob[i] = b[j]; oc[i] = c[j];
    }
}

__host__ __device__ inline long long pair_index(long long M, long long a, long long b) { return a * M - a * (a + 1) / 2 + (b - a - 1); }

// all camera pairs of every point, in internal point order: key = pair index, values = point, the io indices of its two observations
__global__ void k_lay_hits(int N, int M, const int* __restrict__ cnt, const int* __restrict__ slice_base, const int* __restrict__ ipt_ofs,
                           const int* __restrict__ e_cam, const long long* __restrict__ hit_ofs, int* __restrict__ key, int* __restrict__ hq,
                           int* __restrict__ hpi, int* __restrict__ hpj) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= N) return;
    const int n = cnt[q], base = slice_base[q >> 6] + (q & 63), io0 = ipt_ofs[q];
    long long at = hit_ofs[q];
    for (int a = 0; a < n; ++a) {
        const int ca = e_cam[base + 64 * a];
        for (int b = a + 1; b < n; ++b) {
            const int cb = e_cam[base + 64 * b];
            key[at] = (int)pair_index(M, ca, cb);
            hq[at] = q; hpi[at] = io0 + a; hpj[at] = io0 + b;
            ++at;
        }
    }
}

// ---- the merged record layout of the weighted / robust runs (Layout::w_fix ...)
__global__ void k_lay_wsize(int N, const int* __restrict__ cnt, int* __restrict__ sz) {  // pieces of record q; entry N: the zero record
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q <= N) sz[q] = q < N ? ((cnt[q] + 6 + 7) & ~7) : 8;
}
__global__ void k_lay_wfix(int N, const int* __restrict__ cnt, const int* __restrict__ sz, const int* __restrict__ wb, int* __restrict__ w_fix,
                           int* __restrict__ sc_ofs) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > N) return;
    const int f = wb[q] + sz[q] - 6;
    w_fix[q] = f;
    sc_ofs[q] = f - (q < N ? cnt[q] : 0);
}
__global__ void k_lay_pair_w(long long E, const int* __restrict__ pts, const int* __restrict__ pi, const int* __restrict__ pj,
                             const int* __restrict__ ipt_ofs, const int* __restrict__ cnt, const int* __restrict__ w_fix,
                             int* __restrict__ rec, int* __restrict__ kk) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long long)gridDim.x * blockDim.x) {
        const int q = pts[e], io0 = ipt_ofs[q], c = cnt[q];
        rec[e] = w_fix[q];
        kk[e] = (c - (pi[e] - io0)) | ((c - (pj[e] - io0)) << 16);
    }
}
__global__ void k_lay_cm_w(long long K, const int* __restrict__ cm_pt, const int* __restrict__ cm_io, const int* __restrict__ ipt_ofs,
                           const int* __restrict__ cnt, const int* __restrict__ w_fix, int* __restrict__ cm_rec, int* __restrict__ cm_sc) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < K; i += (long long)gridDim.x * blockDim.x) {
        const int q = cm_pt[i], f = w_fix[q];
        cm_rec[i] = f;
        cm_sc[i] = f - cnt[q] + (cm_io[i] - ipt_ofs[q]);
    }
}
// dg_ofs[(cam C + ch) spc + s]: first camera-major entry of diagonal item (cam, ch, s): the entries of camera cam whose point lies in
// point-range chunk ch (the chunks of the pair lists: chunk(q) = q C / N), cut into spc equal parts; one thread per (cam, ch)
__global__ void k_lay_diag_items(int M, int C, int spc, int N, const int* __restrict__ cam_ofs, const int* __restrict__ cm_pt, int* __restrict__ dg_ofs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > M * C) return;
    if (t == M * C) { dg_ofs[(size_t)M * C * spc] = cam_ofs[M]; return; }
    const int cam = t / C, ch = t % C;
    auto first_in = [&](int chunk) {  // first entry of the camera whose point's chunk is >= chunk
        int lo = cam_ofs[cam], hi = cam_ofs[cam + 1];
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((int)((long long)cm_pt[mid] * C / N) < chunk) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const int b = first_in(ch), e = first_in(ch + 1);
    for (int s = 0; s < spc; ++s) dg_ofs[(size_t)t * spc + s] = b + (int)((long long)(e - b) * s / spc);
}

__global__ void k_lay_pair_ij(int M, int2* __restrict__ ij) {
    const int i = blockIdx.x;
    for (int j = i + 1 + threadIdx.x; j < M; j += blockDim.x) ij[pair_index(M, i, j)] = make_int2(i, j);
}

}  // namespace satba
