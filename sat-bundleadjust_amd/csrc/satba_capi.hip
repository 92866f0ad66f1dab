// satba_capi.hip -- C ABI of libsatba_hip.so (declared in include/satba.h) over the kernels of satba_kernels.h.
//
// The handle owns every device array of one shard (all cameras, a contiguous range of points, their
// observations).  Phases are launched asynchronously on the handle's stream; the only synchronisation points
// are the functions that return data to the host.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/satba.h"
#include "satba_chol.h"
#include "satba_chol_dag.h"
#include "satba_kernels.h"
#include "satba_linearize3.h"
#include "satba_schur3.h"

using namespace satba;

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(SATBA_E_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct satba_problem {
    int model = 0, M = 0, N = 0, NP = 0, c_p = 0, n_cam_fix = 0, n_pts_fix = 0, rank = 0, world = 1, f32 = 0, device = 0;
    long long K = 0, n_total = 0;
    int n_c = 0, n = 0, hdr = 0;
    int loss = 0;
    int camc_lds = 0;          // camera-constant table fits the per-workgroup LDS budget (40 KB)
    size_t camc_bytes = 0;
    double f_scale = 1.0, lead = 1.0;
    hipStream_t stream = nullptr;
    // observation data
    double2* d_obs = nullptr;
    double* d_w = nullptr;
    int *d_cam = nullptr, *d_pt = nullptr, *d_tile_start = nullptr;
    unsigned char* d_tile_split = nullptr;
    int n_tiles = 0, n_split = 0;
    int *d_split_pts = nullptr, *d_split_o0 = nullptr, *d_split_o1 = nullptr;
    int *d_cam_ofs = nullptr, *d_cam_obs = nullptr, *d_pt_ofs = nullptr;  // camera-major lists, point CSR
    int sch_T = 0, sch_ctiles = 0, sch_chunks = 0, sch_camc_lds = 0;       // Schur panel configuration (T == 0: v1 kernel)
    size_t sch_lds = 0;
    double *d_S_part = nullptr, *d_rhs_part = nullptr;
    double* d_dch = nullptr;   // camera step in scaled variables
    DagWorkspace dag;          // dataflow Cholesky (dag.n_tasks == 0: blocked multi-launch version)
    double *d_cam_static = nullptr, *d_rpc = nullptr;
    // solver state
    double *d_x = nullptr, *d_xnew = nullptr, *d_camc = nullptr, *d_camc_new = nullptr;
    double *d_scale_inv = nullptr, *d_g = nullptr, *d_gh = nullptr, *d_gn = nullptr, *d_q1 = nullptr, *d_wv = nullptr;
    double *d_U = nullptr, *d_gc = nullptr, *d_V = nullptr, *d_Vinv = nullptr, *d_tbuf = nullptr, *d_dc = nullptr;
    double2* d_f = nullptr;
    double* d_part = nullptr;
    int lin_grid = 0;
    // linearize v3 (two register-accumulating passes); lin3_chunks == 0: not used
    int lin3_chunks = 0, lin3_grid = 0;
    double2* d_cm_obs = nullptr;
    double* d_cm_w = nullptr;
    int* d_cm_pt = nullptr;
    double* d_part3 = nullptr;
    // Schur v3 (camera-pair intersection): sch3_chunks == 0: not used
    int sch3_chunks = 0, NW = 0;
    int sch3_groups = 0;    // > 0: lane-group list kernel, number of pair groups
    int sch3_group_pairs = 8;
    int sch3_chunk_mul = 1;   // fine chunks per coarse chunk of the pair lists
    int sch3_groups_m = 0;      // groups of the moments kernel
    bool sch3_moments = false;  // affine + unit weights: pair blocks through point moments (linear loss only)
    double* d_Tbuf = nullptr;
    double* d_Jpm = nullptr;    // RPC: Jacobian blocks of the current linearisation per observation (written by the linearize kernels)
    int *d_pair_pi = nullptr, *d_pair_pj = nullptr;  // observation indices of the two observations of every pair-list entry
    double2* d_sc = nullptr;    // Jacobian row scales of the current linearisation per observation (weighted / robust runs)
    double c0[3] = {0, 0, 0};   // expansion point of the moments
    bool c0_set = false;
    int* d_groups = nullptr;
    unsigned long long* d_bits = nullptr;
    int* d_rank = nullptr;
    double* d_PV = nullptr;    // packed per-point records (N x 12)
    long long* d_pair_ofs = nullptr;  // per camera pair: list of shared points (null: bitmap scan)
    int2* d_pair_ij = nullptr;        // pair index -> (i, j)
    int* d_pair_pts = nullptr;
    double* d_pair_part = nullptr;  // chunk partials of the pair blocks
    int unit_weights = 0;
    int u_full = 1;            // linearize accumulates the full U_c blocks (0: diagonal only, Schur v3 adds the rest)
    int* d_fail = nullptr;
    int chol_mode = 0;  // SATBA_CHOL: 0 double steps (default), 1 two launches per panel, 2 single steps
    double* d_scal = nullptr;  // 8 private scalars (costs of satba_residuals, timing sinks)
    double* d_keep = nullptr;  // SATBA_KEEP_LEN scalars of the running iteration that outlive the per-phase headers
    bool prepared = false;
    double *d_xb_own = nullptr, *d_xb = nullptr;
    long long xb_len = 0;
    double* h_pin = nullptr;  // pinned staging for header reads
    bool linearized = false, have_step = false;
    std::vector<void*> allocs;

    double* payload() const { return d_xb + hdr; }
};

template <class T>
static int dev_alloc(satba_problem* p, T** out, size_t count) {
    void* ptr = nullptr;
    HIP_TRY(hipMalloc(&ptr, (count ? count : 1) * sizeof(T)));
    p->allocs.push_back(ptr);
    *out = static_cast<T*>(ptr);
    return 0;
}

#define TRY(expr)            \
    do {                     \
        int rc_ = (expr);    \
        if (rc_) return rc_; \
    } while (0)

// dispatch on (camera model, parameters per camera, camera table in LDS); valid pairs: affine {3,5},
// perspective / rpc {3,6}
#define SATBA_DISPATCH(p, ...)                                                                                    \
    do {                                                                                                          \
        const int key_ = (p)->model * 10 + (p)->NP + ((p)->camc_lds ? 100 : 0);                                   \
        switch (key_) {                                                                                           \
            case 3:   { constexpr int MODEL = AFFINE, NP = 3; constexpr bool CL = false; (void)CL; __VA_ARGS__; } break;      \
            case 5:   { constexpr int MODEL = AFFINE, NP = 5; constexpr bool CL = false; (void)CL; __VA_ARGS__; } break;      \
            case 13:  { constexpr int MODEL = PERSPECTIVE, NP = 3; constexpr bool CL = false; (void)CL; __VA_ARGS__; } break; \
            case 16:  { constexpr int MODEL = PERSPECTIVE, NP = 6; constexpr bool CL = false; (void)CL; __VA_ARGS__; } break; \
            case 23:  { constexpr int MODEL = RPC, NP = 3; constexpr bool CL = false; (void)CL; __VA_ARGS__; } break;         \
            case 26:  { constexpr int MODEL = RPC, NP = 6; constexpr bool CL = false; (void)CL; __VA_ARGS__; } break;         \
            case 103: { constexpr int MODEL = AFFINE, NP = 3; constexpr bool CL = true; (void)CL; __VA_ARGS__; } break;       \
            case 105: { constexpr int MODEL = AFFINE, NP = 5; constexpr bool CL = true; (void)CL; __VA_ARGS__; } break;       \
            case 113: { constexpr int MODEL = PERSPECTIVE, NP = 3; constexpr bool CL = true; (void)CL; __VA_ARGS__; } break;  \
            case 116: { constexpr int MODEL = PERSPECTIVE, NP = 6; constexpr bool CL = true; (void)CL; __VA_ARGS__; } break;  \
            case 123: { constexpr int MODEL = RPC, NP = 3; constexpr bool CL = true; (void)CL; __VA_ARGS__; } break;          \
            case 126: { constexpr int MODEL = RPC, NP = 6; constexpr bool CL = true; (void)CL; __VA_ARGS__; } break;          \
            default: return fail(SATBA_E_ARG, "unsupported (cam_model, n_params) = (%d, %d)", (p)->model, (p)->NP);           \
        }                                                                                                         \
    } while (0)

static ObsArgs obs_args(const satba_problem* p, bool at_new) {
    ObsArgs a;
    a.obs = p->d_obs; a.w = p->d_w; a.cam = p->d_cam; a.pt = p->d_pt;
    a.tile_start = p->d_tile_start; a.tile_split = p->d_tile_split;
    a.x = at_new ? p->d_xnew : p->d_x;
    a.camc = at_new ? p->d_camc_new : p->d_camc;
    a.rpc = p->d_rpc;
    a.Jpm = at_new ? nullptr : p->d_Jpm;  // stored Jacobian blocks belong to the linearisation at x
    a.sc = (at_new || (p->loss == 0 && p->unit_weights)) ? nullptr : p->d_sc;
    a.K = p->K; a.n_tiles = p->n_tiles; a.M = p->M; a.N = p->N; a.n_c = p->n_c;
    a.n_cam_fix = p->n_cam_fix; a.n_pts_fix = p->n_pts_fix; a.loss = p->loss; a.f32 = p->f32;
    a.f_scale = p->f_scale;
    a.unit = (p->loss == 0 && p->unit_weights) ? 1 : 0;
    return a;
}

static int grid_for(long long work, int block, int cap) {
    long long g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

static int launch_cam_consts(satba_problem* p, bool at_new) {
    hipLaunchKernelGGL(k_cam_consts, dim3((p->M + 63) / 64), dim3(64), 0, p->stream, p->model, p->M, p->NP, p->c_p,
                       at_new ? p->d_xnew : p->d_x, p->d_cam_static, at_new ? p->d_camc_new : p->d_camc);
    HIP_TRY(hipGetLastError());
    return 0;
}

// S z = rhs for the reduced system (S column-major lower, destroyed; b in place)
static int dense_solve(satba_problem* p, double* S, double* b) {
    if (p->dag.n_tasks > 0) {
        HIP_TRY(hipMemsetAsync(p->d_fail, 0, sizeof(int), p->stream));
        HIP_TRY(hipMemsetAsync(p->dag.d_flags, 0, sizeof(int) * p->dag.flag_ints, p->stream));
        DagFlags fl = dag_flags(p->dag);
        hipLaunchKernelGGL(k_chol_dag, dim3(DG_GRID), dim3(DG_THREADS), 0, p->stream, S, p->n_c, b, p->dag.d_tasks, p->dag.n_tasks, fl);
        hipLaunchKernelGGL(k_dag_status, dim3(1), dim3(1), 0, p->stream, fl.ctr, p->d_fail);
        hipLaunchKernelGGL(k_trsv_back, dim3(1), dim3(1024), sizeof(double) * p->n_c, p->stream, S, p->n_c, b);
    } else {
        cholesky_solve(S, p->n_c, b, p->d_fail, p->d_fail + 1, p->chol_mode, p->stream);  // clears d_fail and the step flags
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

static int zero_header(satba_problem* p) {
    HIP_TRY(hipMemsetAsync(p->d_xb, 0, sizeof(double) * p->hdr, p->stream));
    return 0;
}

static int launch_residual(satba_problem* p, bool at_new, double2* f, double* hdr_slot) {
    ObsArgs a = obs_args(p, at_new);
    const int grid = grid_for(p->K, 512, 512);
    if (p->loss == 0 && p->unit_weights)
        SATBA_DISPATCH(p, hipLaunchKernelGGL((k_residual<MODEL, NP, CL, true>), dim3(grid), dim3(512), p->camc_bytes, p->stream, a, f, hdr_slot));
    else
        SATBA_DISPATCH(p, hipLaunchKernelGGL((k_residual<MODEL, NP, CL>), dim3(grid), dim3(512), p->camc_bytes, p->stream, a, f, hdr_slot));
    HIP_TRY(hipGetLastError());
    return 0;
}

// number of workgroup partials the last linearize launch produced
static int lin_partials(const satba_problem* p) { return p->lin_grid; }

template <int MODEL, int NP, bool CL>
static int launch_lin3(satba_problem* p, const ObsArgs& a) {
    Lin3Args s;
    s.pt_ofs = p->d_pt_ofs; s.f = p->d_f; s.V = p->d_V; s.gp = p->d_g + p->n_c;
    s.hdr_cost = p->d_xb + 0; s.hdr_gpmax = p->d_xb + SATBA_HDR_FIXED + p->rank;
    CamMajor cm;
    cm.cam_ofs = p->d_cam_ofs; cm.obs = p->d_cm_obs; cm.w = p->d_cm_w; cm.pt = p->d_cm_pt; cm.oidx = p->d_cam_obs;
    const size_t lds = p->camc_bytes;
    if (p->loss == 0) {
        hipLaunchKernelGGL((k_lin_points<MODEL, NP, false, CL>), dim3(p->lin3_grid), dim3(256), lds, p->stream, a, s);
        hipLaunchKernelGGL((k_lin_cameras<MODEL, NP, false>), dim3(p->lin3_chunks, p->M), dim3(LINC_THREADS), 0, p->stream, a, cm, p->d_part3);
    } else {
        hipLaunchKernelGGL((k_lin_points<MODEL, NP, true, CL>), dim3(p->lin3_grid), dim3(256), lds, p->stream, a, s);
        hipLaunchKernelGGL((k_lin_cameras<MODEL, NP, true>), dim3(p->lin3_chunks, p->M), dim3(LINC_THREADS), 0, p->stream, a, cm, p->d_part3);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

static size_t lin1_lds(const satba_problem* p, bool robust) {
    const int waves = robust ? 8 : 16;
    return sizeof(double) * ((size_t)p->M * cam_acc_stride(p->NP) + (size_t)waves * 9 * LIN_STAGE) + p->camc_bytes;
}

template <int MODEL, int NP, bool CL, bool FULLU>
static int launch_lin1u(satba_problem* p, const ObsArgs& a) {
    double* gpv = p->d_g + p->n_c;
    double* cost = p->d_xb + 0;
    double* gmax = p->d_xb + SATBA_HDR_FIXED + p->rank;
    if (p->loss == 0 && p->unit_weights && MODEL != RPC)  // RPC keeps the generic form (its Jacobian store carries the masks)
        hipLaunchKernelGGL((k_linearize<MODEL, NP, false, CL, FULLU, false, MODEL != RPC>), dim3(p->lin_grid), dim3(LinCfg<false>::THREADS),
                           lin1_lds(p, false), p->stream, a, p->d_f, p->d_V, gpv, p->d_part, cost, gmax);
    else if (p->loss == 0)
        hipLaunchKernelGGL((k_linearize<MODEL, NP, false, CL, FULLU>), dim3(p->lin_grid), dim3(LinCfg<false>::THREADS),
                           lin1_lds(p, false), p->stream, a, p->d_f, p->d_V, gpv, p->d_part, cost, gmax);
    else if (p->loss == SATBA_LOSS_SOFT_L1 && MODEL != RPC)  // the pipeline's robust loss, specialised (RPC needs the registers)
        hipLaunchKernelGGL((k_linearize<MODEL, NP, true, CL, FULLU, MODEL != RPC>), dim3(p->lin_grid), dim3(LinCfg<MODEL == RPC>::THREADS),
                           lin1_lds(p, MODEL == RPC), p->stream, a, p->d_f, p->d_V, gpv, p->d_part, cost, gmax);
    else
        hipLaunchKernelGGL((k_linearize<MODEL, NP, true, CL, FULLU>), dim3(p->lin_grid), dim3(LinCfg<true>::THREADS),
                           lin1_lds(p, true), p->stream, a, p->d_f, p->d_V, gpv, p->d_part, cost, gmax);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int MODEL, int NP, bool CL>
static int launch_lin1(satba_problem* p, const ObsArgs& a) {
    return p->u_full ? launch_lin1u<MODEL, NP, CL, true>(p, a) : launch_lin1u<MODEL, NP, CL, false>(p, a);
}

static int launch_linearize_kernel(satba_problem* p) {
    ObsArgs a = obs_args(p, false);
    if (p->lin3_grid > 0) {
        SATBA_DISPATCH(p, TRY((launch_lin3<MODEL, NP, CL>(p, a))));
        return 0;
    }
    SATBA_DISPATCH(p, TRY((launch_lin1<MODEL, NP, CL>(p, a))));
    return 0;
}

static size_t schur_lds(const satba_problem* p) { return sizeof(double) * ((size_t)4 * 64 * p->NP * 3 + p->n_c); }

template <int MODEL, int NP, bool ADDU>
static int launch_schur3(satba_problem* p, const ObsArgs& a, double* S, double* rhs) {
    CamMajor cm;
    cm.cam_ofs = p->d_cam_ofs; cm.obs = p->d_cm_obs; cm.w = p->d_cm_w; cm.pt = p->d_cm_pt; cm.oidx = p->d_cam_obs;
    Schur3Args s;
    s.bits = p->d_bits; s.rank = p->d_rank; s.Vinv = p->d_Vinv; s.gp = p->d_g + p->n_c; s.NW = p->NW; s.n_chunks = p->sch3_chunks;
    s.PV = reinterpret_cast<const double2*>(p->d_PV);
    s.pair_ofs = p->d_pair_ofs; s.pair_pts = p->d_pair_pts; s.pair_part = p->d_pair_part; s.chunk_mul = p->sch3_chunk_mul;
    s.pair_pi = p->d_pair_pi; s.pair_pj = p->d_pair_pj;
    const long long n_pairs = (long long)p->M * (p->M - 1) / 2;
    const long long items = n_pairs * p->sch3_chunks;
    const int diag_chunks = p->lin3_chunks;
    const bool grp = p->sch3_groups > 0;
    const unsigned ggrid = grp ? (unsigned)((p->sch3_groups + S3_GW - 1) / S3_GW) : 0;
    // list path: 2-D grid (pairs / 4, chunks); bitmap path: 1-D over the (pair, chunk) items
    if (items >= (1ll << 31)) return fail(SATBA_E_ARG, "too many (camera pair, chunk) work items");
    const dim3 igrid = p->d_pair_ofs ? dim3((unsigned)((n_pairs + 3) / 4), (unsigned)p->sch3_chunks) : dim3((unsigned)((items + 3) / 4));
    s.pair_ij = p->d_pair_ij;
    const bool moments = p->sch3_moments && p->loss == 0 && items > 0;
    // diagonal blocks first: for RPC cameras this pass also stores the Jacobian blocks the pair kernel gathers
    if (p->loss == 0) hipLaunchKernelGGL((k_schur_diag<MODEL, NP, false, ADDU>), dim3(p->M, diag_chunks), dim3(LINC_THREADS), 0, p->stream, a, cm, s, p->d_part3);
    else hipLaunchKernelGGL((k_schur_diag<MODEL, NP, true, ADDU>), dim3(p->M, diag_chunks), dim3(LINC_THREADS), 0, p->stream, a, cm, s, p->d_part3);
    bool done = false;
    if constexpr (MODEL == AFFINE) {
        if (moments) {
            hipLaunchKernelGGL((k_schur_pairs_moments<6>), dim3((unsigned)((p->sch3_groups_m + S3_GW - 1) / S3_GW)), dim3(64 * S3_GW), 0, p->stream,
                               p->M, p->n_pts_fix, s, p->d_groups, p->sch3_groups_m, p->c0[0], p->c0[1], p->c0[2], p->d_Tbuf);
            const long long outs = n_pairs * NP * NP;
            hipLaunchKernelGGL((k_schur_contract<NP>), dim3((unsigned)((outs + 255) / 256)), dim3(256), 0, p->stream, a, p->d_Tbuf,
                               p->c0[0], p->c0[1], p->c0[2], S);
            done = true;
        }
    }
    if (done || items == 0) {
    } else if (p->loss == 0 && p->unit_weights) {
        if (grp) {
            if (p->sch3_group_pairs == 10)
                hipLaunchKernelGGL((k_schur_pairs_groups6<MODEL, NP, false, true>), dim3(ggrid), dim3(64 * S3_GW), 0, p->stream, a, cm, s, p->d_groups, p->sch3_groups, S);
            else if constexpr (MODEL == AFFINE)
                hipLaunchKernelGGL((k_schur_pairs_groups_occ3<MODEL, NP, false, true>), dim3(ggrid), dim3(64 * S3_GW), 0, p->stream, a, cm, s, p->d_groups, p->sch3_groups, S);
            else
                hipLaunchKernelGGL((k_schur_pairs_groups<MODEL, NP, false, true>), dim3(ggrid), dim3(64 * S3_GW), 0, p->stream, a, cm, s, p->d_groups, p->sch3_groups, S);
        } else {
            hipLaunchKernelGGL((k_schur_pairs<MODEL, NP, false, true>), igrid, dim3(256), 0, p->stream, a, cm, s, S);
        }
    } else if (!grp && p->d_pair_ofs && p->d_pair_pi && a.sc) {
        // weighted / robust with pair lists that carry the observation indices: unit Jacobians times the stored scales
        hipLaunchKernelGGL((k_schur_pairs<MODEL, NP, false, false, true>), igrid, dim3(256), 0, p->stream, a, cm, s, S);
    } else if (p->loss == 0) {
        if (grp) hipLaunchKernelGGL((k_schur_pairs_groups<MODEL, NP, false, false>), dim3(ggrid), dim3(64 * S3_GW), 0, p->stream, a, cm, s, p->d_groups, p->sch3_groups, S);
        else hipLaunchKernelGGL((k_schur_pairs<MODEL, NP, false, false>), igrid, dim3(256), 0, p->stream, a, cm, s, S);
    } else {
        if (grp) hipLaunchKernelGGL((k_schur_pairs_groups<MODEL, NP, true, false>), dim3(ggrid), dim3(64 * S3_GW), 0, p->stream, a, cm, s, p->d_groups, p->sch3_groups, S);
        else hipLaunchKernelGGL((k_schur_pairs<MODEL, NP, true, false>), igrid, dim3(256), 0, p->stream, a, cm, s, S);
    }
    HIP_TRY(hipGetLastError());
    if (items > 0 && p->d_pair_ofs && p->sch3_chunks > 1 && !grp && !(MODEL == AFFINE && moments)) {
        const long long outs = n_pairs * p->NP * p->NP;
        hipLaunchKernelGGL(k_schur_pairs_reduce, dim3((unsigned)((outs + 255) / 256)), dim3(256), 0, p->stream, p->M, p->NP, p->n_c,
                           p->sch3_chunks, p->d_pair_part, S);
        HIP_TRY(hipGetLastError());
    }
    const int total = p->M * cam_acc_len(p->NP);
    hipLaunchKernelGGL(k_schur_diag_finish, dim3((total + 255) / 256), dim3(256), 0, p->stream, p->M, p->NP, p->n_c, diag_chunks,
                       p->d_part3, S, rhs);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int launch_schur_kernel(satba_problem* p) {
    ObsArgs a = obs_args(p, false);
    double* S = p->payload();
    double* rhs = S + (size_t)p->n_c * p->n_c;
    if (p->sch3_chunks > 0) {  // v3: camera-pair intersection, register accumulation
        if (p->u_full) SATBA_DISPATCH(p, TRY((launch_schur3<MODEL, NP, false>(p, a, S, rhs))));
        else SATBA_DISPATCH(p, TRY((launch_schur3<MODEL, NP, true>(p, a, S, rhs))));
        return 0;
    }
    if (p->sch_T > 0) {  // v2: LDS column panels, no global atomics
        SchurArgs s;
        s.cam_ofs = p->d_cam_ofs; s.cam_obs = p->d_cam_obs; s.pt_ofs = p->d_pt_ofs;
        s.Vinv = p->d_Vinv; s.gp = p->d_g + p->n_c; s.S_part = p->d_S_part; s.rhs_part = p->d_rhs_part;
        s.T = p->sch_T; s.n_ctiles = p->sch_ctiles; s.n_chunks = p->sch_chunks; s.camc_in_lds = p->sch_camc_lds;
        SATBA_DISPATCH(p, hipLaunchKernelGGL((k_schur_panel<MODEL, NP, CL>), dim3(p->sch_ctiles * p->sch_chunks),
                                             dim3(SCHUR_THREADS), p->sch_lds, p->stream, a, s));
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_schur_reduce, dim3(grid_for((long long)p->n_c * p->n_c, 256, 2048)), dim3(256), 0, p->stream,
                           p->n_c, p->sch_chunks, p->d_S_part, p->d_rhs_part, S, rhs);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const int grid = grid_for(p->n_tiles, 4, 2048);
    SATBA_DISPATCH(p, hipLaunchKernelGGL((k_schur<MODEL, NP>), dim3(grid), dim3(256), schur_lds(p), p->stream, a, p->d_Vinv,
                                          p->d_g + p->n_c, S, rhs));
    HIP_TRY(hipGetLastError());
    if (p->n_split > 0) {
        SATBA_DISPATCH(p, hipLaunchKernelGGL((k_schur_split<MODEL, NP>), dim3(grid_for(p->n_split, 1, 1024)), dim3(64), 0,
                                              p->stream, a, p->n_split, p->d_split_pts, p->d_split_o0, p->d_split_o1,
                                              p->d_Vinv, p->d_g + p->n_c, S, rhs));
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

static int launch_backsub_kernel(satba_problem* p) {
    ObsArgs a = obs_args(p, false);
    const int grid = grid_for(p->n_tiles, 4, 2048);
    // affine: the kernel builds its own 15-double-per-camera table in LDS (k_backsub); otherwise the camera-constant table
    const size_t lds = p->model == AFFINE ? sizeof(double) * (size_t)p->M * BS_ROW : p->camc_bytes;
    if (lds > 158 * 1024) return fail(SATBA_E_ARG, "n_cam = %d exceeds the LDS budget of the back-substitution kernel", p->M);
    SATBA_DISPATCH(p, hipLaunchKernelGGL((k_backsub<MODEL, NP, CL>), dim3(grid), dim3(256), lds, p->stream, a, p->d_dc, p->d_tbuf));
    HIP_TRY(hipGetLastError());
    return 0;
}

// pre: q1 is already in unscaled variables (nv == 1 only)
static int launch_jvp(satba_problem* p, int nv, const double* q1, const double* q2, double* out, bool pre = false) {
    ObsArgs a = obs_args(p, false);
    const int grid = grid_for(p->K, 512, 512);
    if (nv == 1 && pre) {
        SATBA_DISPATCH(p, hipLaunchKernelGGL((k_jvp<MODEL, NP, 1, CL, true>), dim3(grid), dim3(512), p->camc_bytes + sizeof(double) * p->n_c,
                                              p->stream, a, q1, q2, p->d_scale_inv, out));
    } else if (nv == 1) {
        SATBA_DISPATCH(p, hipLaunchKernelGGL((k_jvp<MODEL, NP, 1, CL>), dim3(grid), dim3(512), p->camc_bytes, p->stream, a, q1, q2,
                                              p->d_scale_inv, out));
    } else {
        SATBA_DISPATCH(p, hipLaunchKernelGGL((k_jvp<MODEL, NP, 2, CL>), dim3(grid), dim3(512), p->camc_bytes, p->stream, a, q1, q2,
                                              p->d_scale_inv, out));
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// start of satba_schur_auto: the (already all-reduced) prepare header -> keep[1] = |g|_inf, keep[2..4] = |g_h|^2,
// |J_h g_h|^2, |x_h|^2; trust radius (scipy trf.py:440-442 when Delta <= 0: first iteration) -> keep[6]; damping of the
// Gauss-Newton system from the Cauchy step (scipy trf.py:473-477, common.py:302-322) -> keep[5]
__global__ void k_lambda(const double* __restrict__ hdr, double Delta, double lam_floor, double* __restrict__ keep) {
    const double gh_sq = hdr[1], jg_sq = hdr[2], xs_sq = hdr[3];
    keep[1] = fmax(keep[1], hdr[4]);
    keep[2] = gh_sq; keep[3] = jg_sq; keep[4] = xs_sq;
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
    keep[5] = lam;
    keep[6] = Delta;
}

template <class K>
static int raise_lds_limit(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

extern "C" {

const char* satba_last_error(void) { return g_err.c_str(); }
int satba_version(void) { return 1; }

int satba_problem_create(const satba_problem_desc* d, satba_problem** out) {
    if (!d || !out) return fail(SATBA_E_ARG, "null argument");
    *out = nullptr;
    if (d->cam_model < 0 || d->cam_model > 2) return fail(SATBA_E_ARG, "cam_model must be 0, 1 or 2");
    const int c_p_expected = d->cam_model == SATBA_AFFINE ? 8 : (d->cam_model == SATBA_PERSPECTIVE ? 11 : 9);
    if (d->cam_param_len != c_p_expected) return fail(SATBA_E_ARG, "cam_param_len %d, expected %d", d->cam_param_len, c_p_expected);
    const int np_rt = d->cam_model == SATBA_AFFINE ? 5 : 6;
    if (d->n_params != 3 && d->n_params != np_rt) return fail(SATBA_E_ARG, "n_params %d not in {3, %d}", d->n_params, np_rt);
    if (d->n_cam <= 0 || d->n_pts < 0 || d->n_obs < 0) return fail(SATBA_E_ARG, "negative size");
    if (d->n_obs >= (1ll << 31) - 64) return fail(SATBA_E_ARG, "more than 2^31 observations per shard");
    if (!d->cam_params || (d->n_obs && (!d->cam_ind || !d->pts_ind || !d->pts2d || !d->weights)))
        return fail(SATBA_E_ARG, "null input array");
    if (d->cam_model == SATBA_RPC && !d->rpc_tables) return fail(SATBA_E_ARG, "rpc_tables required for cam_model rpc");
    if (d->n_cam_fix < 0 || d->n_cam_fix > d->n_cam || d->n_pts_fix < 0 || d->n_pts_fix > d->n_pts)
        return fail(SATBA_E_ARG, "n_cam_fix / n_pts_fix out of range");
    if (d->world < 1 || d->rank < 0 || d->rank >= d->world) return fail(SATBA_E_ARG, "bad rank / world");
    // validate indices, build wave tiles (whole points, <= 64 observations)
    const long long K = d->n_obs;
    std::vector<int> tile_start;
    std::vector<unsigned char> tile_split;
    std::vector<int> split_pts, split_o0, split_o1;
    tile_start.push_back(0);
    {
        long long o = 0;
        int fill = 0;  // observations in the open tile
        while (o < K) {
            const int pt = d->pts_ind[o];
            if (pt < 0 || pt >= d->n_pts) return fail(SATBA_E_ARG, "pts_ind[%lld] = %d out of range", o, pt);
            long long e = o;
            while (e < K && d->pts_ind[e] == pt) {
                const int c = d->cam_ind[e];
                if (c < 0 || c >= d->n_cam) return fail(SATBA_E_ARG, "cam_ind[%lld] = %d out of range", e, c);
                ++e;
            }
            if (e < K && d->pts_ind[e] < pt) return fail(SATBA_E_ARG, "pts_ind must be non-decreasing (point-major order)");
            const long long k = e - o;
            if (k > 64) {
                if (fill > 0) { tile_start.push_back((int)o); tile_split.push_back(0); fill = 0; }
                for (long long s = o; s < e; s += 64) {
                    tile_start.push_back((int)std::min(e, s + 64));
                    tile_split.push_back(1);
                }
                split_pts.push_back(pt); split_o0.push_back((int)o); split_o1.push_back((int)e);
            } else {
                if (fill + k > 64) { tile_start.push_back((int)o); tile_split.push_back(0); fill = 0; }
                fill += (int)k;
            }
            o = e;
        }
        if (fill > 0) { tile_start.push_back((int)K); tile_split.push_back(0); }
    }

    // camera-major observation lists and point CSR offsets (Schur panel kernel)
    std::vector<int> cam_ofs(d->n_cam + 1, 0), cam_obs(K), pt_ofs(d->n_pts + 1, 0);
    for (long long o = 0; o < K; ++o) { ++cam_ofs[d->cam_ind[o] + 1]; ++pt_ofs[d->pts_ind[o] + 1]; }
    for (int c = 0; c < d->n_cam; ++c) cam_ofs[c + 1] += cam_ofs[c];
    for (int q = 0; q < d->n_pts; ++q) pt_ofs[q + 1] += pt_ofs[q];
    {
        std::vector<int> fill(cam_ofs.begin(), cam_ofs.end() - 1);
        for (long long o = 0; o < K; ++o) cam_obs[fill[d->cam_ind[o]]++] = (int)o;
    }

    satba_problem* p = new (std::nothrow) satba_problem();
    if (!p) return fail(SATBA_E_ARG, "out of host memory");
    p->model = d->cam_model; p->M = d->n_cam; p->N = d->n_pts; p->NP = d->n_params; p->c_p = d->cam_param_len;
    p->n_cam_fix = d->n_cam_fix; p->n_pts_fix = d->n_pts_fix; p->rank = d->rank; p->world = d->world;
    p->f32 = d->rpc_store_f32; p->device = d->device; p->K = K; p->n_total = d->n_total;
    p->n_c = p->M * p->NP; p->n = p->n_c + 3 * p->N;
    p->hdr = SATBA_HDR_FIXED + p->world + (p->world & 1);
    p->lead = p->rank == 0 ? 1.0 : 0.0;
    p->n_tiles = (int)tile_split.size();
    p->n_split = (int)split_pts.size();

    int rc = [&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        p->camc_lds = (sizeof(double) * (size_t)p->M * CAMC <= 40 * 1024 && !getenv("SATBA_CAMC_GLOBAL")) ? 1 : 0;
        p->camc_bytes = p->camc_lds ? sizeof(double) * (size_t)p->M * CAMC : 0;
        // the fused linearize kernel keeps a per-workgroup camera table in LDS; beyond ~700 cameras the two-pass
        // variant (satba_linearize3.h), which has no such table, takes over
        const bool lin1_fits = lin1_lds(p, false) <= 158 * 1024;
        if (schur_lds(p) > 160 * 1024)
            return fail(SATBA_E_ARG, "n_cam = %d exceeds the LDS budget of the Schur kernels of this build", p->M);
        if (lin1_fits) {
            SATBA_DISPATCH(p, TRY(raise_lds_limit(k_linearize<MODEL, NP, false, CL, true>, lin1_lds(p, false))));
            SATBA_DISPATCH(p, TRY(raise_lds_limit(k_linearize<MODEL, NP, false, CL, true, false, MODEL != RPC>, lin1_lds(p, false))));
            SATBA_DISPATCH(p, TRY(raise_lds_limit(k_linearize<MODEL, NP, false, CL, false, false, MODEL != RPC>, lin1_lds(p, false))));
            SATBA_DISPATCH(p, TRY(raise_lds_limit(k_linearize<MODEL, NP, true, CL, true>, lin1_lds(p, true))));
            SATBA_DISPATCH(p, TRY(raise_lds_limit(k_linearize<MODEL, NP, false, CL, false>, lin1_lds(p, false))));
            SATBA_DISPATCH(p, TRY(raise_lds_limit(k_linearize<MODEL, NP, true, CL, false>, lin1_lds(p, true))));
        }
        SATBA_DISPATCH(p, TRY(raise_lds_limit(k_schur<MODEL, NP>, schur_lds(p))));
        if (p->model == AFFINE) SATBA_DISPATCH(p, TRY(raise_lds_limit(k_backsub<MODEL, NP, CL>, sizeof(double) * (size_t)p->M * BS_ROW)));
        {   // Schur panel configuration: T cameras per panel so that panel (+ camera table) fit the 160 KB LDS
            const size_t budget = 160 * 1024 - 1024;  // static LDS of the kernel and alignment slack
            const size_t col_bytes = sizeof(double) * ((size_t)p->NP * p->n_c + p->NP);  // per camera of the tile
            const size_t camc_bytes = sizeof(double) * (size_t)p->M * CAMC;
            int T = 0;
            const int camc_lds = p->camc_lds;
            const size_t table = camc_lds ? camc_bytes : 0;
            if (table + col_bytes <= budget) T = (int)((budget - table) / col_bytes);
            if (T > SCHUR_MAX_T) T = SCHUR_MAX_T;
            if (T > p->M) T = p->M;
            if (getenv("SATBA_SCHUR_V1")) T = 0;
            p->sch_T = T;
            if (T > 0) {
                p->sch_camc_lds = camc_lds;
                p->sch_ctiles = (p->M + T - 1) / T;
                int chunks = (512 + p->sch_ctiles - 1) / p->sch_ctiles;  // ~2 workgroups' worth of work per CU
                if (chunks > 16) chunks = 16;
                if (chunks < 1) chunks = 1;
                while (chunks > 1 && (size_t)chunks * p->n_c * p->n_c * sizeof(double) > ((size_t)1 << 30)) --chunks;
                p->sch_chunks = chunks;
                p->sch_lds = col_bytes * T + (camc_lds ? camc_bytes : 0);
                SATBA_DISPATCH(p, TRY(raise_lds_limit(k_schur_panel<MODEL, NP, CL>, p->sch_lds)));
                TRY(dev_alloc(p, &p->d_S_part, (size_t)chunks * p->n_c * p->n_c));
                TRY(dev_alloc(p, &p->d_rhs_part, (size_t)chunks * p->n_c));
            }
        }
        TRY(dev_alloc(p, &p->d_cam_ofs, cam_ofs.size())); TRY(dev_alloc(p, &p->d_cam_obs, cam_obs.size()));
        TRY(dev_alloc(p, &p->d_pt_ofs, pt_ofs.size()));
        TRY(dev_alloc(p, &p->d_obs, K)); TRY(dev_alloc(p, &p->d_w, K)); TRY(dev_alloc(p, &p->d_cam, K)); TRY(dev_alloc(p, &p->d_pt, K));
        TRY(dev_alloc(p, &p->d_f, K));
        TRY(dev_alloc(p, &p->d_tile_start, tile_start.size())); TRY(dev_alloc(p, &p->d_tile_split, tile_split.size()));
        TRY(dev_alloc(p, &p->d_split_pts, split_pts.size())); TRY(dev_alloc(p, &p->d_split_o0, split_pts.size()));
        TRY(dev_alloc(p, &p->d_split_o1, split_pts.size()));
        TRY(dev_alloc(p, &p->d_cam_static, (size_t)p->M * p->c_p));
        if (p->model == RPC) TRY(dev_alloc(p, &p->d_rpc, (size_t)p->M * SATBA_RPC_TABLE_LEN));
        const size_t n = p->n;
        TRY(dev_alloc(p, &p->d_x, n)); TRY(dev_alloc(p, &p->d_xnew, n)); TRY(dev_alloc(p, &p->d_scale_inv, n));
        TRY(dev_alloc(p, &p->d_g, n)); TRY(dev_alloc(p, &p->d_gh, n)); TRY(dev_alloc(p, &p->d_gn, n));
        TRY(dev_alloc(p, &p->d_q1, n)); TRY(dev_alloc(p, &p->d_wv, n));
        TRY(dev_alloc(p, &p->d_camc, (size_t)p->M * CAMC)); TRY(dev_alloc(p, &p->d_camc_new, (size_t)p->M * CAMC));
        TRY(dev_alloc(p, &p->d_U, (size_t)p->M * p->NP * p->NP)); TRY(dev_alloc(p, &p->d_gc, p->n_c));
        TRY(dev_alloc(p, &p->d_V, (size_t)6 * p->N)); TRY(dev_alloc(p, &p->d_Vinv, (size_t)6 * p->N));
        TRY(dev_alloc(p, &p->d_tbuf, (size_t)3 * p->N)); TRY(dev_alloc(p, &p->d_dc, p->n_c));
        TRY(dev_alloc(p, &p->d_fail, 1 + CH_MAX_STEPS));  // [0] not-SPD flag, then the panel-step flags
        if (p->n_c > CH_NB * CH_MAX_STEPS) return fail(SATBA_E_ARG, "reduced camera system too large for the dense solver");
        { const char* cs = getenv("SATBA_CHOL"); p->chol_mode = cs ? atoi(cs) : 0; }
        TRY(dev_alloc(p, &p->d_dch, p->n_c));
        if (getenv("SATBA_CHOL_DAG")) {  // experimental dataflow Cholesky (satba_chol_dag.h): correct, but slower
                                         // than the blocked multi-launch version on MI355X
            std::vector<DagTask> tasks = dag_task_list(p->n_c);
            p->dag.NT = (p->n_c + DG_T - 1) / DG_T;
            p->dag.n_tasks = (int)tasks.size();
            p->dag.flag_ints = dag_flag_ints(p->dag.NT);
            TRY(dev_alloc(p, &p->dag.d_tasks, tasks.size()));
            TRY(dev_alloc(p, &p->dag.d_flags, p->dag.flag_ints));
            HIP_TRY(hipMemcpy(p->dag.d_tasks, tasks.data(), sizeof(DagTask) * tasks.size(), hipMemcpyHostToDevice));
            if (getenv("SATBA_DAG_TIMES")) TRY(dev_alloc(p, &p->dag.d_times, 4 * tasks.size()));
        }
        TRY(dev_alloc(p, &p->d_sc, (size_t)std::max<long long>(K, 1)));  // row scales of weighted / robust runs
        if (p->model == RPC && !getenv("SATBA_RPC_RECOMPUTE")) TRY(dev_alloc(p, &p->d_Jpm, (size_t)std::max<long long>(K, 1) * (2 * p->NP + 6)));
        TRY(dev_alloc(p, &p->d_scal, 8));
        TRY(dev_alloc(p, &p->d_keep, SATBA_KEEP_LEN));
        HIP_TRY(hipMemset(p->d_keep, 0, sizeof(double) * SATBA_KEEP_LEN));
        p->lin_grid = grid_for(p->n_tiles, 16, lin1_lds(p, false) <= 78 * 1024 ? 512 : 256);
        TRY(dev_alloc(p, &p->d_part, (size_t)p->lin_grid * p->M * cam_acc_len(p->NP)));
        {   // camera-major copy of the observation data (Schur v3, linearize v3) and chunking of the camera passes
            const char* sel = getenv("SATBA_LIN");
            const int which = (sel ? atoi(sel) : 1) + (lin1_fits ? 0 : 2);
            int chunks = (2048 + p->M - 1) / p->M;
            if (chunks > 64) chunks = 64;
            while (chunks > 1 && K / ((long long)p->M * chunks) < 512) --chunks;  // keep >= ~2 obs per thread
            if (const char* dc = getenv("SATBA_CM_CHUNKS")) chunks = std::max(1, std::min(256, atoi(dc)));  // experiments
            TRY(dev_alloc(p, &p->d_cm_obs, K)); TRY(dev_alloc(p, &p->d_cm_w, K)); TRY(dev_alloc(p, &p->d_cm_pt, K));
            TRY(dev_alloc(p, &p->d_part3, (size_t)p->M * chunks * cam_acc_len(p->NP)));
            std::vector<double> tmp(2 * (size_t)K + 1);
            for (long long i = 0; i < K; ++i) { tmp[2 * i] = d->pts2d[2 * (size_t)cam_obs[i]]; tmp[2 * i + 1] = d->pts2d[2 * (size_t)cam_obs[i] + 1]; }
            HIP_TRY(hipMemcpy(p->d_cm_obs, tmp.data(), sizeof(double) * 2 * K, hipMemcpyHostToDevice));
            for (long long i = 0; i < K; ++i) tmp[i] = d->weights[cam_obs[i]];
            HIP_TRY(hipMemcpy(p->d_cm_w, tmp.data(), sizeof(double) * K, hipMemcpyHostToDevice));
            std::vector<int> tmpi(K + 1);
            for (long long i = 0; i < K; ++i) tmpi[i] = d->pts_ind[cam_obs[i]];
            HIP_TRY(hipMemcpy(p->d_cm_pt, tmpi.data(), sizeof(int) * K, hipMemcpyHostToDevice));
            p->lin3_chunks = chunks;  // chunk count of the camera-major passes (also used by k_schur_diag)
            p->lin3_grid = (which >= 3 && K > 0) ? grid_for(p->N, 256, 256 * 4) : 0;

            // Schur v3: visibility bitmaps and ranks.  Scan cost grows with M^2 N / 64: beyond 512 cameras the
            // panel kernel (v2) is used instead.  SATBA_SCHUR=1|2 forces the older variants.
            const char* ssel = getenv("SATBA_SCHUR");
            const int swhich = ssel ? atoi(ssel) : 3;
            if (swhich >= 3 && p->M <= 512 && K > 0) {
                const int NW = (p->N + 63) / 64;
                p->NW = NW;
                std::vector<unsigned long long> bits((size_t)p->M * NW, 0ull);
                for (long long o = 0; o < K; ++o)
                    bits[(size_t)d->cam_ind[o] * NW + (d->pts_ind[o] >> 6)] |= 1ull << (d->pts_ind[o] & 63);
                std::vector<int> rank((size_t)p->M * NW);
                for (int cc = 0; cc < p->M; ++cc) {
                    int run = 0;
                    for (int w = 0; w < NW; ++w) { rank[(size_t)cc * NW + w] = run; run += __builtin_popcountll(bits[(size_t)cc * NW + w]); }
                }
                TRY(dev_alloc(p, &p->d_bits, bits.size())); TRY(dev_alloc(p, &p->d_rank, rank.size()));
                TRY(dev_alloc(p, &p->d_PV, (size_t)PV_STRIDE * p->N));
                p->unit_weights = 1;
                for (long long o = 0; o < K; ++o) if (d->weights[o] != 1.0) { p->unit_weights = 0; break; }
                HIP_TRY(hipMemcpy(p->d_bits, bits.data(), sizeof(unsigned long long) * bits.size(), hipMemcpyHostToDevice));
                HIP_TRY(hipMemcpy(p->d_rank, rank.data(), sizeof(int) * rank.size(), hipMemcpyHostToDevice));
                const long long n_pairs = (long long)p->M * (p->M - 1) / 2;
                int sc = 1;
                if (n_pairs > 0) {
                    sc = (int)std::min<long long>((8192 + n_pairs - 1) / n_pairs, (NW + 63) / 64);
                    if (sc < 1) sc = 1;
                }
                p->sch3_chunks = sc;
                // shared-point lists per camera pair (static structure; 4 B per pair entry, e.g. 200 MB at 200 x 1M x 10M),
                // cut into point-range chunks sized so that one chunk's packed point records (96 B each) stay in L2
                if (n_pairs > 0 && !getenv("SATBA_SCHUR_BITMAP")) {
                    // measured on 200 x 1M x 10M: 12 MB windows (8 chunks) are the optimum between gather locality and
                    // the fixed cost per (pair, chunk) work item; fewer chunks when the lists are short
                    long long n_hits = 0;
                    for (int q = 0; q < p->N; ++q) { const long long dq = pt_ofs[q + 1] - pt_ofs[q]; n_hits += dq * (dq - 1) / 2; }
                    // Coarse chunks (one wave per (pair, chunk) item kernel): 12 MB windows and >= 256 hits per item are
                    // the measured optimum between gather locality and the fixed cost per item.  Fine chunks (lane-group
                    // kernels, which walk all chunks inside one wave): 3 MB windows stay in the 4 MB L2 of every XCD.
                    // The lists are cut into the fine chunks; the item kernel takes them chunk_mul at a time.
                    const char* st = getenv("SATBA_SCHUR_STREAM");
                    const bool stream = st && atoi(st) == 1;
                    // 32 MB windows of the 128-byte point records (with line-aligned records the optimum moved from 8 chunks to
                    // 3..5 at 200 x 1M: 0.835 -> 0.789 ms for the Schur phase)
                    int Cc = (int)std::max<long long>(1, ((long long)p->N * 8 * PV_STRIDE + (32ll << 20) - 1) / (32ll << 20));
                    Cc = (int)std::max<long long>(1, std::min<long long>(Cc, n_hits / n_pairs / 256));
                    // few cameras: enough (pair, chunk) items to fill the chip (>= 8192 waves), at least 64 hits each
                    Cc = (int)std::max<long long>(Cc, std::min<long long>((8192 + n_pairs - 1) / n_pairs, std::max<long long>(1, n_hits / n_pairs / 64)));
                    if (const char* cs = getenv("SATBA_SCHUR_CHUNKS")) Cc = std::max(1, atoi(cs));
                    Cc = std::min(Cc, S3_MAXC);
                    while (Cc > 1 && n_pairs * (long long)(Cc + 1) > (1ll << 27)) --Cc;
                    const char* mo = getenv("SATBA_SCHUR_MOMENTS");
                    const bool moments = p->model == AFFINE && p->unit_weights && mo && atoi(mo) == 1;  // experiment
                    int mul = 1;
                    if (stream || moments) {
                        long long Cf = std::max<long long>(1, ((long long)p->N * 96 + (3ll << 20) - 1) / (3ll << 20));
                        Cf = std::max<long long>(1, std::min<long long>(Cf, n_hits / n_pairs / 24));
                        if (const char* cs = getenv("SATBA_SCHUR_FINE")) Cf = std::max(1, atoi(cs));
                        mul = (int)std::max<long long>(1, std::min<long long>(Cf, S3_MAXC) / Cc);
                        while (mul > 1 && n_pairs * (long long)(Cc * mul + 1) > (1ll << 27)) --mul;
                    }
                    p->sch3_chunk_mul = mul;
                    const int C = Cc * mul;  // chunks the lists are cut into
                    const long long M_ = p->M;
                    auto pair_index = [M_](long long a, long long b) { return a * M_ - a * (a + 1) / 2 + (b - a - 1); };
                    auto chunk_of = [&](int q) { return (int)((long long)q * C / std::max(p->N, 1)); };
                    // counts per (pair, chunk), then offsets in (pair-major, chunk) order == the order of the lists
                    std::vector<long long> ofs((size_t)n_pairs * (C + 1) + 1, 0);
                    for (int q = 0; q < p->N; ++q) {
                        const int ch = chunk_of(q);
                        for (int x0 = pt_ofs[q]; x0 < pt_ofs[q + 1]; ++x0)
                            for (int x1 = x0 + 1; x1 < pt_ofs[q + 1]; ++x1)
                                ++ofs[pair_index(d->cam_ind[x0], d->cam_ind[x1]) * (C + 1) + ch + 1];
                    }
                    // ofs[pair*(C+1) + ch + 1] holds count(pair, ch); turn into running offsets, entry [pair*(C+1)] = start
                    long long run = 0;
                    for (long long pr = 0; pr < n_pairs; ++pr) {
                        ofs[pr * (C + 1)] = run;
                        for (int ch = 0; ch < C; ++ch) { run += ofs[pr * (C + 1) + ch + 1]; ofs[pr * (C + 1) + ch + 1] = run; }
                    }
                    const long long E = run;
                    if (E > 0 && E < (1ll << 31) && E * 4 < (8ll << 30)) {
                        // per entry: the point and the indices of its two observations (the weighted / robust and the RPC pair
                        // kernels fetch per-observation data there; 12 B per entry, 540 MB at 200 x 1M x 10M)
                        const bool with_pos = !getenv("SATBA_SCHUR_NO_POS") && E * 12 < (24ll << 30);
                        std::vector<int> pts(E), ppi(with_pos ? E : 0), ppj(with_pos ? E : 0);
                        std::vector<long long> fill((size_t)n_pairs);
                        for (long long pr = 0; pr < n_pairs; ++pr) fill[pr] = ofs[pr * (C + 1)];
                        for (int q = 0; q < p->N; ++q)  // ascending q: each pair's list comes out sorted, chunks contiguous
                            for (int x0 = pt_ofs[q]; x0 < pt_ofs[q + 1]; ++x0)
                                for (int x1 = x0 + 1; x1 < pt_ofs[q + 1]; ++x1) {
                                    const long long at = fill[pair_index(d->cam_ind[x0], d->cam_ind[x1])]++;
                                    pts[at] = q;
                                    if (with_pos) { ppi[at] = x0; ppj[at] = x1; }
                                }
                        {   // pair index -> (i, j) table (the kernels used to unrank it with a square root and two loops)
                            std::vector<int2> ij((size_t)n_pairs);
                            size_t at = 0;
                            for (int ci = 0; ci < p->M; ++ci)
                                for (int cj = ci + 1; cj < p->M; ++cj) ij[at++] = make_int2(ci, cj);
                            TRY(dev_alloc(p, &p->d_pair_ij, ij.size()));
                            HIP_TRY(hipMemcpy(p->d_pair_ij, ij.data(), sizeof(int2) * ij.size(), hipMemcpyHostToDevice));
                        }
                        TRY(dev_alloc(p, &p->d_pair_ofs, ofs.size())); TRY(dev_alloc(p, &p->d_pair_pts, pts.size()));
                        HIP_TRY(hipMemcpy(p->d_pair_ofs, ofs.data(), sizeof(long long) * ofs.size(), hipMemcpyHostToDevice));
                        HIP_TRY(hipMemcpy(p->d_pair_pts, pts.data(), sizeof(int) * pts.size(), hipMemcpyHostToDevice));
                        if (with_pos) {
                            TRY(dev_alloc(p, &p->d_pair_pi, ppi.size())); TRY(dev_alloc(p, &p->d_pair_pj, ppj.size()));
                            HIP_TRY(hipMemcpy(p->d_pair_pi, ppi.data(), sizeof(int) * ppi.size(), hipMemcpyHostToDevice));
                            HIP_TRY(hipMemcpy(p->d_pair_pj, ppj.data(), sizeof(int) * ppj.size(), hipMemcpyHostToDevice));
                        }
                        p->sch3_chunks = Cc;
                        if (stream || moments) {  // groups of pairs (i, j0 ..) of one camera i: 8 (10 with six lanes per pair)
                            std::vector<int> groups;
                            const char* gp = getenv("SATBA_SCHUR_GROUP");
                            const int PW = (moments && !stream) || (gp && atoi(gp) == 10 && p->unit_weights) ? 10 : 8;
                            p->sch3_group_pairs = PW;
                            for (int ci = 0; ci + 1 < p->M; ++ci)
                                for (int cj = ci + 1; cj < p->M; cj += PW) {
                                    groups.push_back(ci); groups.push_back(cj); groups.push_back(std::min(PW, p->M - cj));
                                }
                            TRY(dev_alloc(p, &p->d_groups, groups.size()));
                            HIP_TRY(hipMemcpy(p->d_groups, groups.data(), sizeof(int) * groups.size(), hipMemcpyHostToDevice));
                            if (stream) p->sch3_groups = (int)(groups.size() / 3);
                            if (moments && !stream) {
                                p->sch3_moments = true;
                                p->sch3_groups_m = (int)(groups.size() / 3);
                                TRY(dev_alloc(p, &p->d_Tbuf, (size_t)n_pairs * S3_NT));
                            }
                        }
                        if (Cc > 1 && !stream) TRY(dev_alloc(p, &p->d_pair_part, (size_t)Cc * n_pairs * p->NP * p->NP));
                    }
                }
            }
            if (swhich == 1) p->sch_T = 0;
            // with Schur v3 and the fused linearize kernel, U_c's off-diagonal entries are formed in k_schur_diag
            p->u_full = (p->sch3_chunks > 0 && p->lin3_grid == 0 && !getenv("SATBA_FULL_U")) ? 0 : 1;
        }
        p->xb_len = satba_exchange_len(p);
        TRY(dev_alloc(p, &p->d_xb_own, p->xb_len));
        p->d_xb = p->d_xb_own;
        HIP_TRY(hipHostMalloc((void**)&p->h_pin, sizeof(double) * (p->hdr + 64)));
        HIP_TRY(hipMemset(p->d_xb, 0, sizeof(double) * p->xb_len));
        HIP_TRY(hipMemset(p->d_scale_inv, 0, sizeof(double) * n));
        HIP_TRY(hipMemset(p->d_x, 0, sizeof(double) * n));
        HIP_TRY(hipMemset(p->d_xnew, 0, sizeof(double) * n));
        // uploads (synchronous: the caller's arrays may go away after this call)
        HIP_TRY(hipMemcpy(p->d_obs, d->pts2d, sizeof(double) * 2 * K, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_w, d->weights, sizeof(double) * K, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_cam, d->cam_ind, sizeof(int) * K, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_pt, d->pts_ind, sizeof(int) * K, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_cam_ofs, cam_ofs.data(), sizeof(int) * cam_ofs.size(), hipMemcpyHostToDevice));
        if (K) HIP_TRY(hipMemcpy(p->d_cam_obs, cam_obs.data(), sizeof(int) * K, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_pt_ofs, pt_ofs.data(), sizeof(int) * pt_ofs.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_tile_start, tile_start.data(), sizeof(int) * tile_start.size(), hipMemcpyHostToDevice));
        if (!tile_split.empty())
            HIP_TRY(hipMemcpy(p->d_tile_split, tile_split.data(), tile_split.size(), hipMemcpyHostToDevice));
        if (p->n_split) {
            HIP_TRY(hipMemcpy(p->d_split_pts, split_pts.data(), sizeof(int) * p->n_split, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(p->d_split_o0, split_o0.data(), sizeof(int) * p->n_split, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(p->d_split_o1, split_o1.data(), sizeof(int) * p->n_split, hipMemcpyHostToDevice));
        }
        HIP_TRY(hipMemcpy(p->d_cam_static, d->cam_params, sizeof(double) * p->M * p->c_p, hipMemcpyHostToDevice));
        if (p->model == RPC)
            HIP_TRY(hipMemcpy(p->d_rpc, d->rpc_tables, sizeof(double) * p->M * SATBA_RPC_TABLE_LEN, hipMemcpyHostToDevice));
        return 0;
    }();
    if (rc) {
        satba_problem_destroy(p);
        return rc;
    }
    *out = p;
    return 0;
}

void satba_problem_destroy(satba_problem* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    for (void* q : p->allocs) (void)hipFree(q);
    if (p->h_pin) (void)hipHostFree(p->h_pin);
    delete p;
}

int satba_set_stream(satba_problem* p, void* hip_stream) {
    if (!p) return fail(SATBA_E_ARG, "null handle");
    p->stream = static_cast<hipStream_t>(hip_stream);
    return 0;
}

int64_t satba_header_len(const satba_problem* p) { return p ? p->hdr : 0; }

int64_t satba_exchange_len(const satba_problem* p) {
    if (!p) return 0;
    const long long lin = (long long)p->M * p->NP * p->NP + p->n_c;
    const long long sch = (long long)p->n_c * p->n_c + p->n_c;
    return p->hdr + (lin > sch ? lin : sch);
}

int satba_bind_exchange(satba_problem* p, double* device_ptr, int64_t len) {
    if (!p) return fail(SATBA_E_ARG, "null handle");
    if (!device_ptr) { p->d_xb = p->d_xb_own; return 0; }
    if (len < satba_exchange_len(p)) return fail(SATBA_E_ARG, "exchange buffer too small: %lld < %lld", (long long)len, (long long)satba_exchange_len(p));
    p->d_xb = device_ptr;
    return 0;
}

// lower triangle of S (column-major, n x n) <-> packed columns; block j = column j, block n = header and rhs
__global__ __launch_bounds__(256) void k_pack_lower(int n, int hdr, double* __restrict__ xb, double* __restrict__ packed, int unpack) {
    const int j = blockIdx.x;
    const long long tri = (long long)n * (n + 1) / 2;
    if (j == n) {
        for (int i = threadIdx.x; i < hdr + n; i += 256) {
            double* a = i < hdr ? xb + i : xb + hdr + (size_t)n * n + (i - hdr);
            double* b = i < hdr ? packed + i : packed + hdr + tri + (i - hdr);
            if (unpack) *a = *b; else *b = *a;
        }
        return;
    }
    double* col = xb + hdr + (size_t)j * n;
    double* pk = packed + hdr + ((long long)j * n - (long long)j * (j - 1) / 2) - j;  // pk[r] for r >= j
    for (int r = j + threadIdx.x; r < n; r += 256) {
        if (unpack) col[r] = pk[r]; else pk[r] = col[r];
    }
}

int64_t satba_packed_schur_len(const satba_problem* p) {
    return p ? p->hdr + (long long)p->n_c * (p->n_c + 1) / 2 + p->n_c : 0;
}

static int pack_schur_impl(satba_problem* p, double* packed, int unpack) {
    if (!p || !packed) return fail(SATBA_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    hipLaunchKernelGGL(k_pack_lower, dim3(p->n_c + 1), dim3(256), 0, p->stream, p->n_c, (int)p->hdr, p->d_xb, packed, unpack);
    HIP_TRY(hipGetLastError());
    return 0;
}
int satba_pack_schur(satba_problem* p, double* packed) { return pack_schur_impl(p, packed, 0); }
int satba_unpack_schur(satba_problem* p, const double* packed) { return pack_schur_impl(p, const_cast<double*>(packed), 1); }

int satba_configure(satba_problem* p, int32_t loss, double f_scale) {
    if (!p) return fail(SATBA_E_ARG, "null handle");
    if (loss < 0 || loss > 4) return fail(SATBA_E_ARG, "unknown loss %d", loss);
    if (!(f_scale > 0.0)) return fail(SATBA_E_ARG, "f_scale must be positive");
    p->loss = loss; p->f_scale = f_scale;
    return 0;
}

int satba_set_x(satba_problem* p, const double* host_x) {
    if (!p || !host_x) return fail(SATBA_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMemcpyAsync(p->d_x, host_x, sizeof(double) * p->n, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->N > 0 && !p->c0_set) {  // expansion point of the Schur moments: any point of the scene (conditioning only)
        const double* q = host_x + p->n_c + 3 * (size_t)(p->N / 2);
        p->c0[0] = q[0]; p->c0[1] = q[1]; p->c0[2] = q[2];
        p->c0_set = true;
    }
    TRY(launch_cam_consts(p, false));
    p->linearized = false; p->have_step = false;
    return 0;
}

int satba_get_x(satba_problem* p, double* host_x) {
    if (!p || !host_x) return fail(SATBA_E_ARG, "null argument");
    HIP_TRY(hipMemcpyAsync(host_x, p->d_x, sizeof(double) * p->n, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return 0;
}

int satba_residuals(satba_problem* p, double* host_r, double* host_cost) {
    if (!p) return fail(SATBA_E_ARG, "null handle");
    HIP_TRY(hipSetDevice(p->device));
    // the cost goes through a private scalar so the exchange header of a running solve is left alone
    double* slot = p->d_scal;
    HIP_TRY(hipMemsetAsync(slot, 0, sizeof(double), p->stream));
    TRY(launch_residual(p, false, p->d_f, slot));
    HIP_TRY(hipMemcpyAsync(p->h_pin, slot, sizeof(double), hipMemcpyDeviceToHost, p->stream));
    if (host_r) HIP_TRY(hipMemcpyAsync(host_r, p->d_f, sizeof(double) * 2 * p->K, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (host_cost) *host_cost = p->h_pin[0];
    return 0;
}

int satba_linearize(satba_problem* p) {
    if (!p) return fail(SATBA_E_ARG, "null handle");
    HIP_TRY(hipSetDevice(p->device));
    TRY(zero_header(p));
    if (p->n_split) {
        HIP_TRY(hipMemsetAsync(p->d_V, 0, sizeof(double) * 6 * p->N, p->stream));
        HIP_TRY(hipMemsetAsync(p->d_g + p->n_c, 0, sizeof(double) * 3 * p->N, p->stream));
    }
    TRY(launch_linearize_kernel(p));
    if (p->n_split) {
        hipLaunchKernelGGL(k_gpmax, dim3(grid_for(3ll * p->N, 256, 1024)), dim3(256), 0, p->stream, p->N, p->d_g + p->n_c,
                           p->d_xb + SATBA_HDR_FIXED + p->rank);
        HIP_TRY(hipGetLastError());
    }
    double* U = p->payload();
    double* gc = U + (size_t)p->M * p->NP * p->NP;
    const int total = p->M * cam_acc_len(p->NP);
    if (p->lin3_grid > 0) {
        hipLaunchKernelGGL(k_lin3_finish, dim3((total + 255) / 256), dim3(256), 0, p->stream, p->M, p->NP, p->lin3_chunks, p->d_part3, U, gc);
        HIP_TRY(hipGetLastError());
        p->linearized = true; p->have_step = false;
        return 0;
    }
    // unit weights + linear loss + affine R+T: the translation entries of diag(U_c) are (observation count) x constants
    // and were not accumulated by the kernel (lin_const_t in satba_kernels.h)
    const bool const_t = !p->u_full && lin_const_t(p->model, p->NP, p->loss != 0, p->loss == 0 && p->unit_weights);
    hipLaunchKernelGGL(k_lin_finish, dim3((total + 63) / 64), dim3(1024), 0, p->stream, p->M, p->NP, lin_partials(p), p->d_part,
                       nullptr, U, gc, const_t ? p->d_cam_ofs : nullptr, p->d_camc, p->n_cam_fix);
    HIP_TRY(hipGetLastError());
    p->linearized = true; p->have_step = false;
    return 0;
}

int satba_prepare(satba_problem* p, int32_t first) {
    if (!p) return fail(SATBA_E_ARG, "null handle");
    if (!p->linearized) return fail(SATBA_E_STATE, "prepare before linearize");
    HIP_TRY(hipSetDevice(p->device));
    const size_t nU = (size_t)p->M * p->NP * p->NP;
    if (p->hdr > 1024) return fail(SATBA_E_ARG, "header too long");
    hipLaunchKernelGGL(k_prepare_stash, dim3(1), dim3(1024), 0, p->stream, (int)nU, p->n_c, p->world, (int)p->hdr, p->d_xb, p->d_U, p->d_gc,
                       p->d_keep);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_prepare_vec, dim3(grid_for(p->n, 256, 512)), dim3(256), 0, p->stream, p->n, p->n_c, p->NP, first,
                       p->lead, p->d_U, p->d_gc, p->d_V, p->d_x, p->d_g, p->d_scale_inv, p->d_gh, p->d_q1, p->d_xb);
    HIP_TRY(hipGetLastError());
    TRY(launch_jvp(p, 1, p->d_q1, p->d_q1, p->d_xb + 2, true));  // d_q1 is free until the subspace phase
    p->prepared = true;
    return 0;
}

static int schur_impl(satba_problem* p, double lam, const double* lam_dev) {
    const size_t nS = (size_t)p->n_c * p->n_c + p->n_c;
    HIP_TRY(hipMemsetAsync(p->d_xb, 0, sizeof(double) * (p->hdr + nS), p->stream));
    if (p->N > 0) {
        hipLaunchKernelGGL(k_vinv, dim3((p->N + 255) / 256), dim3(256), 0, p->stream, p->N, lam, lam_dev, p->d_V,
                           p->d_scale_inv + p->n_c, p->d_Vinv, p->d_x + p->n_c, p->d_g + p->n_c, p->d_PV);
        HIP_TRY(hipGetLastError());
    }
    double* S = p->payload();
    hipLaunchKernelGGL(k_schur_init, dim3((p->M * p->NP * p->NP + 255) / 256), dim3(256), 0, p->stream, p->M, p->NP, lam, lam_dev,
                       p->lead, p->u_full, p->d_U, p->d_gc, p->d_scale_inv, S, S + (size_t)p->n_c * p->n_c);
    HIP_TRY(hipGetLastError());
    if (p->K > 0) TRY(launch_schur_kernel(p));
    return 0;
}

int satba_schur(satba_problem* p, double lam) {
    if (!p) return fail(SATBA_E_ARG, "null handle");
    if (!p->linearized) return fail(SATBA_E_STATE, "schur before linearize");
    HIP_TRY(hipSetDevice(p->device));
    return schur_impl(p, lam, nullptr);
}

int satba_schur_auto(satba_problem* p, double Delta, double lam_floor) {
    if (!p) return fail(SATBA_E_ARG, "null handle");
    if (!p->linearized || !p->prepared) return fail(SATBA_E_STATE, "schur_auto before prepare");
    HIP_TRY(hipSetDevice(p->device));
    hipLaunchKernelGGL(k_lambda, dim3(1), dim3(1), 0, p->stream, p->d_xb, Delta, lam_floor, p->d_keep);
    HIP_TRY(hipGetLastError());
    p->prepared = false;  // the prepare header is gone after this call
    return schur_impl(p, 0.0, p->d_keep + 5);
}

int satba_solve(satba_problem* p) {
    if (!p) return fail(SATBA_E_ARG, "null handle");
    HIP_TRY(hipSetDevice(p->device));
    double* S = p->payload();
    double* rhs = S + (size_t)p->n_c * p->n_c;
    hipLaunchKernelGGL(k_scale_system, dim3(grid_for((long long)p->n_c * p->n_c, 256, 2048)), dim3(256), 0, p->stream, p->n_c,
                       p->d_scale_inv, S, rhs, p->d_dch);
    HIP_TRY(hipGetLastError());
    TRY(dense_solve(p, S, p->d_dch));  // also clears the not-SPD flag first
    const int nu = std::max(p->n_c, (int)p->hdr);
    hipLaunchKernelGGL(k_unscale, dim3((nu + 255) / 256), dim3(256), 0, p->stream, p->n_c, p->d_scale_inv, p->d_dch, p->d_dc, (int)p->hdr,
                       p->d_xb, p->d_fail, p->lead, p->d_keep);
    HIP_TRY(hipGetLastError());
    if (p->n_split > 0) HIP_TRY(hipMemsetAsync(p->d_tbuf, 0, sizeof(double) * 3 * p->N, p->stream));
    if (p->K > 0) TRY(launch_backsub_kernel(p));
    hipLaunchKernelGGL(k_backsub_finish, dim3(grid_for(p->n_c + p->N, 256, 512)), dim3(256), 0, p->stream, p->n_c, p->N, p->lead,
                       p->d_dch, p->d_Vinv, p->d_g, p->d_tbuf, p->d_scale_inv, p->d_gh, p->d_gn, p->d_xb);
    HIP_TRY(hipGetLastError());
    p->have_step = true;
    return 0;
}

int satba_subspace(satba_problem* p, double alpha, double inv_norm_g) {
    if (!p) return fail(SATBA_E_ARG, "null handle");
    if (!p->have_step) return fail(SATBA_E_STATE, "subspace before solve");
    HIP_TRY(hipSetDevice(p->device));
    TRY(zero_header(p));
    hipLaunchKernelGGL(k_subspace_vec, dim3(grid_for(p->n, 256, 512)), dim3(256), 0, p->stream, p->n, p->n_c, p->lead, alpha,
                       inv_norm_g, p->d_gh, p->d_gn, p->d_q1, p->d_wv, p->d_xb);
    HIP_TRY(hipGetLastError());
    return 0;
}

int satba_subspace_products(satba_problem* p) {
    if (!p) return fail(SATBA_E_ARG, "null handle");
    if (!p->have_step) return fail(SATBA_E_STATE, "subspace_products before solve");
    HIP_TRY(hipSetDevice(p->device));
    TRY(zero_header(p));
    TRY(launch_jvp(p, 2, p->d_q1, p->d_wv, p->d_xb + 3));
    return 0;
}

int satba_trial(satba_problem* p, double p0, double p1) {
    if (!p) return fail(SATBA_E_ARG, "null handle");
    if (!p->have_step) return fail(SATBA_E_STATE, "trial before solve");
    HIP_TRY(hipSetDevice(p->device));
    TRY(zero_header(p));
    hipLaunchKernelGGL(k_trial_vec, dim3(grid_for(p->n, 256, 512)), dim3(256), 0, p->stream, p->n, p->n_c, p->lead, p0, p1,
                       p->d_x, p->d_q1, p->d_wv, p->d_scale_inv, p->d_xnew, p->d_xb);
    HIP_TRY(hipGetLastError());
    TRY(launch_cam_consts(p, true));
    TRY(launch_residual(p, true, nullptr, p->d_xb + 1));
    return 0;
}

int satba_trial_gn(satba_problem* p, double ca, double cb) {
    if (!p) return fail(SATBA_E_ARG, "null handle");
    if (!p->have_step) return fail(SATBA_E_STATE, "trial before solve");
    HIP_TRY(hipSetDevice(p->device));
    TRY(zero_header(p));
    hipLaunchKernelGGL(k_trial_vec, dim3(grid_for(p->n, 256, 512)), dim3(256), 0, p->stream, p->n, p->n_c, p->lead, ca, cb,
                       p->d_x, p->d_gh, p->d_gn, p->d_scale_inv, p->d_xnew, p->d_xb);
    HIP_TRY(hipGetLastError());
    TRY(launch_cam_consts(p, true));
    TRY(launch_residual(p, true, nullptr, p->d_xb + 1));
    return 0;
}

int satba_accept(satba_problem* p) {
    if (!p) return fail(SATBA_E_ARG, "null handle");
    std::swap(p->d_x, p->d_xnew);
    std::swap(p->d_camc, p->d_camc_new);
    p->linearized = false; p->have_step = false; p->prepared = false;
    return 0;
}

int satba_read_header(satba_problem* p, double* host_hdr) {
    if (!p || !host_hdr) return fail(SATBA_E_ARG, "null argument");
    HIP_TRY(hipMemcpyAsync(p->h_pin, p->d_xb, sizeof(double) * p->hdr, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    memcpy(host_hdr, p->h_pin, sizeof(double) * p->hdr);
    return 0;
}

int satba_get_blocks(satba_problem* p, double* U, double* gc, double* V, double* gp) {
    if (!p) return fail(SATBA_E_ARG, "null handle");
    if (!p->linearized) return fail(SATBA_E_STATE, "get_blocks before linearize");
    HIP_TRY(hipStreamSynchronize(p->stream));
    const size_t nU = (size_t)p->M * p->NP * p->NP;
    if (U && !p->u_full) {
        // the linearize kernel only kept diag(U_c): form the full blocks with the camera-major pass (inspection only)
        double *dU = nullptr, *dg = nullptr;
        HIP_TRY(hipMalloc((void**)&dU, sizeof(double) * nU));
        HIP_TRY(hipMalloc((void**)&dg, sizeof(double) * p->n_c));
        ObsArgs a = obs_args(p, false);
        CamMajor cm;
        cm.cam_ofs = p->d_cam_ofs; cm.obs = p->d_cm_obs; cm.w = p->d_cm_w; cm.pt = p->d_cm_pt; cm.oidx = p->d_cam_obs;
        int rc = [&]() -> int {
            SATBA_DISPATCH(p, hipLaunchKernelGGL((k_lin_cameras<MODEL, NP, true>), dim3(p->lin3_chunks, p->M), dim3(LINC_THREADS), 0,
                                                 p->stream, a, cm, p->d_part3));
            const int total = p->M * cam_acc_len(p->NP);
            hipLaunchKernelGGL(k_lin3_finish, dim3((total + 255) / 256), dim3(256), 0, p->stream, p->M, p->NP, p->lin3_chunks, p->d_part3, dU, dg);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(p->stream));
            HIP_TRY(hipMemcpy(U, dU, sizeof(double) * nU, hipMemcpyDeviceToHost));
            return 0;
        }();
        (void)hipFree(dU);
        (void)hipFree(dg);
        if (rc) return rc;
    } else if (U) {
        HIP_TRY(hipMemcpy(U, p->payload(), sizeof(double) * nU, hipMemcpyDeviceToHost));
    }
    if (gc) HIP_TRY(hipMemcpy(gc, p->payload() + nU, sizeof(double) * p->n_c, hipMemcpyDeviceToHost));
    if (V) HIP_TRY(hipMemcpy(V, p->d_V, sizeof(double) * 6 * p->N, hipMemcpyDeviceToHost));
    if (gp) HIP_TRY(hipMemcpy(gp, p->d_g + p->n_c, sizeof(double) * 3 * p->N, hipMemcpyDeviceToHost));
    return 0;
}

int satba_get_jacobian(satba_problem* p, double* Jc, double* Jp) {
    if (!p || !Jc || !Jp) return fail(SATBA_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    double *dJc = nullptr, *dJp = nullptr;
    HIP_TRY(hipMalloc((void**)&dJc, sizeof(double) * (2 * p->K * p->NP + 1)));
    HIP_TRY(hipMalloc((void**)&dJp, sizeof(double) * (6 * p->K + 1)));
    ObsArgs a = obs_args(p, false);
    int rc = [&]() -> int {
        SATBA_DISPATCH(p, hipLaunchKernelGGL((k_jacobian<MODEL, NP>), dim3(grid_for(p->K, 256, 1024)), dim3(256), 0, p->stream, a, dJc, dJp));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(p->stream));
        HIP_TRY(hipMemcpy(Jc, dJc, sizeof(double) * 2 * p->K * p->NP, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(Jp, dJp, sizeof(double) * 6 * p->K, hipMemcpyDeviceToHost));
        return 0;
    }();
    (void)hipFree(dJc);
    (void)hipFree(dJp);
    return rc;
}

int satba_get_exchange(satba_problem* p, int64_t offset, int64_t n, double* host_out) {
    if (!p || !host_out) return fail(SATBA_E_ARG, "null argument");
    if (offset < 0 || n < 0 || offset + n > p->xb_len) return fail(SATBA_E_ARG, "exchange range out of bounds");
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy(host_out, p->d_xb + offset, sizeof(double) * n, hipMemcpyDeviceToHost));
    return 0;
}

int satba_set_exchange(satba_problem* p, int64_t offset, int64_t n, const double* host_in) {
    if (!p || !host_in) return fail(SATBA_E_ARG, "null argument");
    if (offset < 0 || n < 0 || offset + n > p->xb_len) return fail(SATBA_E_ARG, "exchange range out of bounds");
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy(p->d_xb + offset, host_in, sizeof(double) * n, hipMemcpyHostToDevice));
    return 0;
}

// tools only (not declared in satba.h): per-task timestamps of the last dataflow Cholesky
int satba_debug_dag_times(satba_problem* p, long long* host_out, int32_t* n_tasks) {
    if (!p || !n_tasks) return fail(SATBA_E_ARG, "null argument");
    *n_tasks = p->dag.d_times ? p->dag.n_tasks : 0;
    if (host_out && p->dag.d_times) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        HIP_TRY(hipMemcpy(host_out, p->dag.d_times, sizeof(long long) * 4 * p->dag.n_tasks, hipMemcpyDeviceToHost));
    }
    return 0;
}

// tools only (tools/chol_times.py): wall-clock stamps (100 MHz) of every panel step of one factorisation of the
// current reduced system; host_out holds 8 * CH_MAX_STEPS values.  Destroys S.
int satba_debug_chol_times(satba_problem* p, long long* host_out, int32_t* n_steps) {
    if (!p || !host_out || !n_steps) return fail(SATBA_E_ARG, "null argument");
    long long* d_ts = nullptr;
    HIP_TRY(hipMalloc((void**)&d_ts, sizeof(long long) * 8 * CH_MAX_STEPS));
    HIP_TRY(hipMemset(d_ts, 0, sizeof(long long) * 8 * CH_MAX_STEPS));
    cholesky_solve(p->payload(), p->n_c, p->d_dch, p->d_fail, p->d_fail + 1, 2, p->stream, d_ts);
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy(host_out, d_ts, sizeof(long long) * 8 * CH_MAX_STEPS, hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(d_ts));
    *n_steps = (p->n_c + CH_NB - 1) / CH_NB;
    return 0;
}

int satba_get_vector(satba_problem* p, int32_t which, double* host_out) {
    if (!p || !host_out) return fail(SATBA_E_ARG, "null argument");
    const double* src[] = {p->d_g, p->d_scale_inv, p->d_gn, p->d_q1, p->d_wv, p->d_xnew, p->d_gh};
    if (which < 0 || which > 6) return fail(SATBA_E_ARG, "unknown vector id %d", which);
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy(host_out, src[which], sizeof(double) * p->n, hipMemcpyDeviceToHost));
    return 0;
}

int satba_time_kernel(satba_problem* p, int32_t phase, int32_t reps, float* ms_avg) {
    if (!p || !ms_avg || reps <= 0) return fail(SATBA_E_ARG, "bad argument");
    if (phase < 0 || phase > 5) return fail(SATBA_E_ARG, "unknown phase %d", phase);
    if (phase >= 2 && !p->linearized) return fail(SATBA_E_STATE, "time_kernel: linearize first");
    HIP_TRY(hipSetDevice(p->device));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    // scratch outputs so a measurement never disturbs solver state that a later phase reads
    int rc = 0;
    auto once = [&]() -> int {
        switch (phase) {
            case 0: return launch_residual(p, false, p->d_f, p->d_scal);
            case 1: return launch_linearize_kernel(p);
            case 2: return launch_schur_kernel(p);
            case 3: {
                // factorising an already factorised matrix is meaningless numerically but identical in work
                return dense_solve(p, p->payload(), p->d_dch);
            }
            case 4: return launch_backsub_kernel(p);
            default: return launch_jvp(p, 1, p->d_q1, p->d_q1, p->d_scal, true);  // the pass of the prepare phase
        }
    };
    rc = once();  // warm-up
    if (!rc) {
        (void)hipEventRecord(e0, p->stream);
        for (int i = 0; i < reps && !rc; ++i) rc = once();
        (void)hipEventRecord(e1, p->stream);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *ms_avg = ms / reps;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    p->linearized = false; p->have_step = false;  // blocks / exchange payload were overwritten
    return rc;
}

}  // extern "C"
