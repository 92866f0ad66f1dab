/*
 * satba.h -- C ABI of libsatba_hip.so: the bundle-adjustment least-squares hot path on MI355X (gfx950).
 *
 * The reference (centreborelli/sat-bundleadjust) has no FFI for this path: the boundary is the Python call
 * surface of bundle_adjust/ba_core.py and bundle_adjust/ba_params.py.  This header is what a native binding
 * of that surface needs; each entry point names the reference code it replaces (paths relative to the
 * reference tree, scipy paths relative to scipy 1.15.3).  The only caller in this repository is the ctypes
 * shim sat-bundleadjust_amd/satba/engine_hip.py; INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative SATBA_E_* code and never throws;
 *     satba_last_error() returns a thread-local description of the last failure.
 *   - "host" pointers are caller-owned host memory, copied before the call returns (after a stream sync for
 *     outputs).  "device" pointers are HIP device memory on the problem's device.
 *   - one solve per handle at a time (the reference is single-threaded and non-reentrant as well).
 *   - all arithmetic is IEEE float64; the only float32 is the optional rounding of RPC projections that
 *     mirrors ba_core.py:150.
 *
 * Variable vector (ba_params.py:152-172):  x = [cam 0 (n_params) | ... | cam M-1 | pt 0 (3) | ... | pt N-1 (3)]
 * Residual vector (ba_core.py:180-181):   r = [x0, y0, x1, y1, ...] = w_k * (projection_k - observation_k)
 *
 * Multi-GPU: one process and one handle per GPU.  Each handle holds ALL cameras and a contiguous shard of the
 * points with their observations.  Camera-side sums leave the device through the exchange buffer
 * (satba_bind_exchange), which the host all-reduces (RCCL through torch.distributed) between phases; `rank`
 * and `world` only decide who contributes the camera-only terms (rank 0) and which header slot a rank owns.
 */
#ifndef SATBA_H
#define SATBA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SATBA_AFFINE = 0, SATBA_PERSPECTIVE = 1, SATBA_RPC = 2 };
/* losses of scipy.optimize.least_squares (scipy:optimize/_lsq/least_squares.py:172-207) */
enum { SATBA_LOSS_LINEAR = 0, SATBA_LOSS_SOFT_L1 = 1, SATBA_LOSS_HUBER = 2, SATBA_LOSS_CAUCHY = 3, SATBA_LOSS_ARCTAN = 4 };

enum {
    SATBA_OK = 0,
    SATBA_E_ARG = -1,      /* invalid argument (the Python shim raises ValueError)        */
    SATBA_E_HIP = -2,      /* HIP runtime error (RuntimeError)                             */
    SATBA_E_STATE = -3,    /* phase called out of order                                    */
    SATBA_E_NONFINITE = -4 /* non-finite residuals where the reference raises ValueError   */
};

/* per-camera RPC record: col_num[20] col_den[20] row_num[20] row_den[20]
 * lon_off lon_scale lat_off lat_scale alt_off alt_scale col_off col_scale row_off row_scale
 * (attribute names of rpcm.RPCModel; polynomial term order of ba_rpcfit.py:17-44 == c/rpc.c:285-289) */
#define SATBA_RPC_TABLE_LEN 90

/* exchange-buffer header: SATBA_HDR_FIXED scalars + one slot per rank, rounded up to an even count */
#define SATBA_HDR_FIXED 16
/* after satba_solve, header slots SATBA_HDR_KEEP .. +SATBA_KEEP_LEN-1 repeat the scalars of the earlier phases of the
 * same iteration: cost, |g|_inf, |g_h|^2, |J_h g_h|^2, |x_h|^2, damping used, trust radius used (0 unless
 * satba_schur_auto computed them), and SATBA_HDR_FX_BAD: non-zero when the fixed-point camera sums of the linearisation left
 * their range (the iteration is void: call satba_camera_sums_fallback and repeat it; satba_solve_lm / satba_lm_step and
 * satba/trf.py do) */
#define SATBA_HDR_KEEP 8
#define SATBA_KEEP_LEN 8
#define SATBA_HDR_FX_BAD 15

typedef struct satba_problem satba_problem;

/* satba_problem_desc.flags.  Every run is bitwise repeatable: all reductions of the library have a fixed order except the
 * per-camera sums of the linearisation, and those are accumulated in 64-bit fixed point (integer addition does not depend on the
 * order; csrc/satba_kernels.h, k_linearize).  CAMERA_MAJOR_SUMS (the former DETERMINISTIC flag, same value; environment
 * variable SATBA_DETERMINISTIC) selects the other route to the same property from the start: a camera-major pass in float64 with
 * a fixed order (k_cam_sums; one extra pass over the observations per linearisation, one lane per point) -- the route a solve
 * falls back to when a term leaves the fixed-point range (satba_camera_sums_fallback). */
#define SATBA_FLAG_DETERMINISTIC 1
#define SATBA_FLAG_CAMERA_MAJOR_SUMS 1

typedef struct satba_problem_desc {
    int32_t cam_model;      /* SATBA_AFFINE | SATBA_PERSPECTIVE | SATBA_RPC  (BundleAdjustmentParameters.cam_model) */
    int32_t n_cam;          /* M                                                                    */
    int32_t n_pts;          /* N: points held by this handle                                        */
    int32_t n_params;       /* optimised parameters per camera: 3 (R) | 5 (affine R+T) | 6 (R+T)     */
    int32_t cam_param_len;  /* columns of cam_params: 8 affine, 11 perspective, 9 rpc (ba_params.py:19-44) */
    int32_t n_cam_fix;      /* first n_cam_fix cameras are frozen (ba_params.py:246-249)             */
    int32_t n_pts_fix;      /* first n_pts_fix LOCAL points are frozen (ba_params.py:240-243)        */
    int32_t rank, world;    /* position of this shard; 0, 1 for a single GPU                        */
    int32_t rpc_store_f32;  /* non-zero: round RPC projections to float32 like ba_core.py:150        */
    int32_t device;         /* HIP device ordinal                                                    */
    int32_t flags;          /* SATBA_FLAG_* bits                                                     */
    int64_t n_obs;          /* K: observations held by this handle                                   */
    int64_t n_total;        /* size of the global variable vector (only used for default max_nfev)   */
    const double *cam_params;  /* host, M x cam_param_len, row-major (BundleAdjustmentParameters.cam_params) */
    const double *rpc_tables;  /* host, M x SATBA_RPC_TABLE_LEN, or NULL unless cam_model == SATBA_RPC */
    const int32_t *cam_ind;    /* host, K: camera of each observation                                */
    const int32_t *pts_ind;    /* host, K: LOCAL point of each observation, non-decreasing (point-major,
                                  the order ba_params.py:142-147 produces)                           */
    const double *pts2d;       /* host, K x 2 observed (col, row)                                    */
    const double *weights;     /* host, K (BundleAdjustmentParameters.pts2d_w)                       */
} satba_problem_desc;

const char *satba_last_error(void);
int satba_version(void);

/* Upload a problem (replaces the per-call numpy gathers of ba_core.py:72-81, 97-107, 147-153) and build its index structures
 * on the device: points sorted by track length into 64-point slices (sliced-ELL observation order), camera-major lists,
 * per-camera-pair lists of shared points (csrc/satba_layout.h).  pts_ind must be non-decreasing and the cameras of a point
 * strictly ascending -- the order ba_params.py:142-147 emits. */
int satba_problem_create(const satba_problem_desc *desc, satba_problem **out);
void satba_problem_destroy(satba_problem *p);

/* A handle launches everything on a stream of its own (created non-blocking with the handle).  satba_set_stream moves it to
 * `hip_stream` (a hipStream_t; NULL = the legacy default stream) -- e.g. the stream torch.distributed queues its collectives on --
 * or, with use_own != 0, back to its own. */
int satba_set_stream(satba_problem *p, void *hip_stream, int32_t use_own);

/* Length in doubles of the exchange buffer: header + max(M n_p^2 + M n_p, (M n_p)^2 + M n_p). */
int64_t satba_exchange_len(const satba_problem *p);
int64_t satba_header_len(const satba_problem *p);
/* Use caller-provided device memory (e.g. a torch tensor that torch.distributed will all-reduce) as the
 * exchange buffer; NULL restores the internally allocated one. */
int satba_bind_exchange(satba_problem *p, double *device_ptr, int64_t len);

/* The reduced camera system S is symmetric and only its lower triangle is formed and used: for the all-reduce between
 * ranks (no counterpart in the reference) the Schur payload [header | S (n_c^2) | rhs (n_c)] of the
 * exchange buffer is packed to [header | rhs | lower triangle, column-major, n_c (n_c + 1) / 2] in caller-provided device
 * memory (again typically a torch tensor), all-reduced there, and unpacked -- half the bytes over xGMI.
 * satba_packed_schur_len: length of that packed buffer in doubles. */
int64_t satba_packed_schur_len(const satba_problem *p);
int satba_pack_schur(satba_problem *p, double *packed_device_ptr);
int satba_unpack_schur(satba_problem *p, const double *packed_device_ptr);

/* The same exchange in MESSAGES, the dense factorisation beside it (round 6; replaces, for several ranks, the one LSMR call of
 * scipy:optimize/_lsq/trf.py:479-480 on the all-reduced system): the packed payload is cut at camera boundaries into
 * satba_solve_messages() pieces of about equal size -- bounds[m] .. bounds[m + 1] is the index range of message m, the first one
 * carries header and right-hand side; returns 0 when this handle solves its system in one piece (fewer than 129 or more than 1 024
 * unknowns: pack, all-reduce, unpack, satba_solve), -1 on a bad argument.  satba_solve_messages_begin packs the payload (unless
 * packed_already: satba_pack_schur has been called and the messages have ALL been reduced -- for collectives that block the host or
 * synchronise the device, e.g. gloo on device tensors, which would wait for the waiting factorisation) and launches
 * the factorisation on a stream of the handle's own, where it waits for tile columns; after the all-reduce of message m (a library
 * that queues it on the handle's stream: RCCL) satba_solve_messages_arrived unpacks its columns and releases the tiles they complete;
 * satba_solve_messages_end queues the rest of the solve phase (what satba_solve leaves behind: step and header).  Same arithmetic in
 * the same order as satba_solve on the unpacked system: the same bits.  _bind names the payload for the device-resident loop's parts
 * 10 (begin), 11 (arrived; the message index travels in lam_floor) and 12 (end) of satba_lm_part. */
int32_t satba_solve_messages(satba_problem *p, int64_t *bounds, int32_t cap);
int satba_solve_messages_bind(satba_problem *p, double *packed_device_ptr);
int satba_solve_messages_begin(satba_problem *p, double *packed_device_ptr, int32_t packed_already);
int satba_solve_messages_arrived(satba_problem *p, const double *packed_device_ptr, int32_t m);
int satba_solve_messages_end(satba_problem *p);

/* loss and f_scale of least_squares (ba_core.py:292-293). */
int satba_configure(satba_problem *p, int32_t loss, double f_scale);

int satba_set_x(satba_problem *p, const double *host_x);  /* n_cam*n_params + 3*n_pts doubles */
int satba_get_x(satba_problem *p, double *host_x);

/* ba_core.fun (ba_core.py:157-183) at the current x: residuals (2K doubles, may be NULL) and the cost
 * 0.5 * sum rho(f^2) (scipy:optimize/_lsq/trf.py:413-418). */
int satba_residuals(satba_problem *p, double *host_r, double *host_cost);

/* ---- phases of one trust-region iteration (satba/trf.py; scipy:optimize/_lsq/trf.py:450-551).
 * Each phase zeroes the exchange header, then writes the slots documented in satba/trf.py.               */

/* residuals + analytic Jacobian -> normal-equation blocks at x.  Replaces scipy's finite-difference
 * Jacobian (scipy:optimize/_numdiff.py:628-705), compute_grad (common.py:590-595) and the robust
 * rescaling (common.py:720-731).  Exchange payload: U (M x n_p x n_p) | g_c (M x n_p); only diag(U_c) is
 * guaranteed (the off-diagonal entries are formed in the Schur phase by its camera-major pass).           */
int satba_linearize(satba_problem *p);
/* after the all-reduce: x_scale="jac" update (common.py:598-610), g_h, |J_h g_h|^2 for the Cauchy step
 * (trf.py:473-477).                                                                                       */
int satba_prepare(satba_problem *p, int32_t first);
/* local part of the reduced camera system  S = U + lam Dc^2 - sum_p W (V + lam Dp^2)^-1 W^T  and its
 * right-hand side; replaces LSMR (trf.py:479-480).  Exchange payload: S (n_c x n_c, column-major lower) | rhs. */
int satba_schur(satba_problem *p, double lam);
/* The same with the damping computed on the device from the (all-reduced) header of satba_prepare, exactly as the
 * host does it for satba_schur: scipy's Cauchy-step regulariser (scipy:optimize/_lsq/trf.py:473-477,
 * common.py:302-322) for the trust radius Delta, floored at lam_floor.  Delta <= 0 selects scipy's initial radius
 * |x_h| (trf.py:440-442).  linearize -> prepare -> schur_auto -> solve can then be queued without a host
 * round trip; satba_solve repeats the scalars the host would have read in between in header slots SATBA_HDR_KEEP.. */
int satba_schur_auto(satba_problem *p, double Delta, double lam_floor);
/* after the all-reduce: dense Cholesky solve, point back-substitution, Gram matrix of (g_h, gn_h).        */
int satba_solve(satba_problem *p);
/* basis of span{g_h, gn_h} (trf.py:481-482): q1 = inv_norm_g * g_h, w = gn_h - alpha * g_h, and their dots.
 * The quadratic model on that basis (trf.py:483-485) follows on the host from the normal equations,
 * J_h^T J_h gn_h = g_h - reg gn_h, without another pass over the observations (satba/trf.py);
 * satba_subspace_products computes the three products |J_h q1|^2, (J_h q1).(J_h w), |J_h w|^2 explicitly
 * (header slots 3..5) for the ill-conditioned case and for the tests.                                      */
int satba_subspace(satba_problem *p, double alpha, double inv_norm_g);
int satba_subspace_products(satba_problem *p);
/* x_new = x + scale * (p0 q1 + p1 w); cost at x_new (trf.py:497-512).                                     */
int satba_trial(satba_problem *p, double p0, double p1);
/* the same step written on (g_h, gn_h): x_new = x + scale * (ca g_h + cb gn_h); needs no satba_subspace.           */
int satba_trial_gn(satba_problem *p, double ca, double cb);
int satba_accept(satba_problem *p);
/* Form the per-camera sums of the following linearisations with the camera-major float64 pass instead of the fixed-point LDS
 * table.  For callers that drive the phases themselves: call it when header slot SATBA_HDR_FX_BAD is non-zero after satba_solve
 * and repeat the iteration from satba_linearize.  (No counterpart in the reference.) */
int satba_camera_sums_fallback(satba_problem *p);
/* synchronise the stream and copy the exchange header to the host. */
int satba_read_header(satba_problem *p, double *host_hdr);

/* One fixed-work Levenberg-Marquardt iteration of a single-rank handle with the host side in C++: linearize, prepare, damped
 * Gauss-Newton step, 2-D trust-region subproblem, trial point, accept if the cost went down -- the body of satba_solve_lm's
 * loop without its termination tests (ref: scipy optimize/_lsq/trf.py:trf_no_bounds, one pass).  bench.py times this.
 * first != 0: first iteration (Jacobian scaling and trust radius are initialised; Delta is ignored).
 * out[8]: cost at x, cost at the trial point, new trust radius, accepted (0/1), interior Newton step (0/1), predicted reduction,
 * actual reduction, damping. */
int satba_lm_step(satba_problem* p, int32_t first, double Delta, double lam_floor, double* out);

/* The same iterations WITHOUT a host round trip (round 3; csrc/satba_lmdev.h): the loop's decisions -- scipy's top-of-loop tests,
 * damping escalation after a failed factorisation, the 2-D trust-region subproblem, radius update, accept / reject -- are taken by
 * one-thread kernels on the device, every kernel of an iteration reads a gate word of the loop's state in device memory, accepted
 * points are copied on the device, and the host only queues launches (a few iterations ahead of the device's progress reports in
 * pinned memory).  satba_lm_run performs n_iterations fixed-work iterations (no termination tests: satba_lm_step's semantics) from
 * the current x; cycle_len > 0: every cycle_len iterations the solve starts again from the point kept by satba_snapshot_x (what
 * bench.py does with the solve that the shipped tolerances end after that many iterations).  It returns when the device is done.
 * out (may be NULL; n_out >= 16), also returned by satba_lm_state: [0..7] as satba_lm_step for the last iteration ([3], [4]: totals
 * of accepted and interior steps), [8] phase (0 running, 1 finished, 2 stopped for the host), [9] scipy status, [10] nfev, [11] njev,
 * [12] iterations, [13] launch patterns ("ticks") executed, [14] reason of phase 2 (1 fixed-point overflow of the camera sums,
 * 2 factorisation failed ten times, 3 non-finite residuals at the start), [15] |g|_inf.  satba_solve_lm runs on the same
 * machinery with scipy's termination tests switched on.  Single-rank handles only. */
int satba_lm_run(satba_problem* p, int64_t n_iterations, int32_t cycle_len, double lam_floor, double* out, int32_t n_out);
int satba_lm_state(satba_problem* p, double* out, int32_t n);

/* ---- one-shot solve: the whole trust-region loop (scipy:optimize/_lsq/trf.py:401-560 as ba_core.py:284-297 configures it,
 * with the exact damped step of this library) below the ABI, for callers that do not want to drive the phases themselves.
 * Starts from the current x (satba_set_x), leaves the solution in the handle (satba_get_x, satba_residuals).  Single-rank
 * handles only (world == 1): with several ranks the exchange buffer must be all-reduced between the phases by the caller.
 * status, nfev: scipy's (0 max_nfev reached, 1 gtol, 2 ftol, 3 xtol, 4 ftol and xtol).  max_nfev <= 0: 100 * n_total. */
typedef struct satba_lm_opts {
    double ftol, xtol, gtol, f_scale;
    int64_t max_nfev;
    int32_t loss;     /* SATBA_LOSS_* */
    int32_t verbose;  /* 0 silent, 1 final message, 2 scipy's iteration table */
} satba_lm_opts;
typedef struct satba_lm_stats {
    double cost, initial_cost, optimality;
    int64_t nfev, njev, iterations;
    int32_t status, reserved;
} satba_lm_stats;
int satba_solve_lm(satba_problem *p, const satba_lm_opts *opts, satba_lm_stats *stats);

/* The device-resident loop for several ranks (replaces the host side of scipy:optimize/_lsq/trf.py:450-551 when the points are sharded,
 * north_star: "LM outer loop on device ... points shard across the 8 GPUs").  A tick is queued in parts with the caller's all-reduces
 * of the exchange buffer between them (torch.distributed on the handle's stream; satba/trf.py: trf_solve_sharded):
 *     0 | linearize payload | 1 | header | 2 | reduced system | 3 | header | 4 | header | 5
 * and the degenerate-subspace pattern  6 | header | 7 | header | 8 | header | 9.  Nothing waits for the device; the decisions are taken
 * by one-thread kernels on all-reduced scalars, identically on every rank.  satba_lm_begin: reset the loop (o: tolerances, loss;
 * never_stop / max_iterations / cycle_len as satba_lm_run).  satba_lm_poll (no wait): out[0] patterns executed, [1] phase (0 running,
 * 1 done, 2 needs the host, 3 paused for the subspace pattern), [2] pauses so far, [3] tick of the latest pause, [4] tick at which the
 * loop left the running phase (0: not yet).  Every rank must queue exactly out[4] + 3 patterns.  satba_lm_state returns the scalars. */
int satba_lm_begin(satba_problem* p, const satba_lm_opts* o, int32_t never_stop, int64_t max_iterations, int32_t cycle_len);
int satba_lm_part(satba_problem* p, int32_t part, double lam_floor);
int satba_lm_poll(satba_problem* p, int64_t* out, int32_t n);

/* ---- outlier rejection between the two solves of the pipeline (ba_outliers.py:14-58, 112-155): per-camera elbow threshold
 * on the reprojection errors of the current x and the observations above it.
 * err (host, K, caller's observation order, may be NULL: computed from the residuals at the current x as
 * ba_core.compute_reprojection_error does); predef_thr < 0: automatic (elbow) thresholds; outputs: cam_thr (host, M),
 * remove (host, K bytes: 1 = observation is an outlier), n_removed. */
int satba_outliers(satba_problem *p, const double *err, double predef_thr, double min_thr, double *cam_thr, uint8_t *remove,
                   int64_t *n_removed);

/* ba_core.compute_reprojection_error(ba_core.fun(x, p), p.pts2d_w) (ref:bundle_adjust/ba_core.py:335-349 on :157-183) at the current
 * x without the residual vector crossing the bus: host_err (K doubles, caller's observation order) = || f_k / w_k ||_2, the
 * reference's operations with their roundings (bit-identical to the host formula); host_cost (optional): 0.5 |f|^2.  What
 * ref:bundle_adjust/ba_core.py:277,303-304 computes before and after the solve.                                                  */
int satba_reprojection_errors(satba_problem *p, double *host_err, double *host_cost);
/* The same in two steps (round 6): _begin queues the error kernels at the current x into a device buffer of the handle and returns
 * without waiting; _fetch moves the K errors to host_err and may be called from ANOTHER host thread while this handle runs the solve
 * that follows -- the download of ref:bundle_adjust/ba_core.py:277's initial errors then overlaps the region :283-299 times.  One
 * _fetch per _begin; the values are those of the x at _begin whatever the handle has done since.                                  */
int satba_reprojection_errors_begin(satba_problem *p);
int satba_reprojection_errors_fetch(satba_problem *p, double *host_err);

/* device-side copy of the current point (no host transfer): restore == 0 keeps x, restore != 0 returns to the kept x, as
 * satba_set_x with the same vector would.  bench.py restarts its solve with it; a caller can use it to retry a solve. */
int satba_snapshot_x(satba_problem *p, int32_t restore);

/* ---- initial triangulation of the feature tracks, the step before the path.  Stand-alone: no problem handle.
 * cameras: n_cam x 12 (3 x 4 projection matrices, row-major; affine and perspective) or n_cam x SATBA_RPC_TABLE_LEN (rpc).
 *
 * satba_triangulate_pairwise replaces ft_triangulate.linear_triangulation_multiple_pts (ft_triangulate.py:18-34, the
 * cv2.triangulatePoints call) and ft_triangulate.rpc_triangulation (ft_triangulate.py:37-54 -> s2p/triangulation.py:82-135 ->
 * c/disp_to_h.c:40-64 `stereo_corresp_to_lonlatalt`): n correspondences (pts_i, pts_j: host, n x 2, (col, row)) between two
 * cameras -> pts3d (host, n x 3 float64, ECEF for rpc) and, for rpc, err (host, n float32, may be NULL: distance to the epipolar
 * curve in pixels).  kernel_ms (may be NULL): duration of the kernel from HIP events.
 *
 * satba_init_pts3d replaces ft_triangulate.init_pts3d (ft_triangulate.py:57-127): every track is triangulated from every listed
 * pair (pairs: host, n_pairs x 2, the reference's pairs_to_triangulate, processed in list order; pairs naming a camera >= n_cam
 * are skipped as at :99) whose two cameras observe it, and the results are folded into a float32 running mean with the
 * reference's sequence of float32 operations.  The tracks come as the observation lists of ba_params.py:142-147 instead of the
 * dense NaN-sparse C matrix: pt_ofs (host, n_pts + 1, pt_ofs[0] = 0), cam_ind (host, K), obs (host, K x 2).  Outputs: pts3d
 * (host, n_pts x 3 float32; zero where no pair applies, like the reference), n_tri (host, n_pts, may be NULL: triangulations
 * per track).  reps >= 1: the kernel is launched reps times (measurement), kernel_ms (may be NULL) = average duration. */
int satba_triangulate_pairwise(int32_t cam_model, const double *cam_i, const double *cam_j, int64_t n, const double *pts_i,
                               const double *pts_j, double *pts3d, float *err, int32_t device, float *kernel_ms);
int satba_init_pts3d(int32_t cam_model, int32_t n_cam, int64_t n_pts, const int64_t *pt_ofs, const int32_t *cam_ind,
                     const double *obs, const double *cameras, int32_t n_pairs, const int32_t *pairs, float *pts3d,
                     int32_t *n_tri, int32_t device, int32_t reps, float *kernel_ms);
/* The same on a problem handle's RESIDENT observations -- what ba_outliers.py:89-93 does right after the outlier rejection: the
 * tracks are not uploaded again.  remove (host, n_obs bytes in the caller's observation order, e.g. satba_outliers' mask; may be
 * NULL): observations to treat as absent.  cameras: host, n_cam x (12 | SATBA_RPC_TABLE_LEN) -- the caller's choice (the
 * reference passes p.cameras).  pts3d (host, n_pts x 3 float32) and n_tri (host, n_pts, may be NULL) in the caller's point order;
 * a track with n_tri = 0 is one ft_utils.filter_C_using_pairs_to_triangulate would drop.  Single-rank handles only. */
int satba_init_pts3d_resident(satba_problem *p, const uint8_t *remove, const double *cameras, int32_t n_pairs,
                              const int32_t *pairs, float *pts3d, int32_t *n_tri, float *kernel_ms);

/* ---- RPC re-fit after the solve, the step behind the path.  Stand-alone: no problem handle.
 * satba_rpc_fit replaces ba_rpcfit.weighted_lsq (ba_rpcfit.py:88-153, with initialize_rpc / scaling_params :156-198), batched over
 * cameras: target (host, n_cam x n_samples x 2: col, row), locs (host, n_cam x n_samples x 3: lon, lat, alt) -> tables (host,
 * n_cam x SATBA_RPC_TABLE_LEN records of the fitted models), rmse (host, n_cam, may be NULL: the loop's last RMSE in pixels),
 * iters (host, n_cam, may be NULL: re-weighted passes that ran).  h, tol, max_iter: the reference's defaults are 1e-3, 1e-2, 20.
 * satba_rpc_localization replaces rpcm.RPCModel.localization as ba_rpcfit.py:245,323 calls it: image points at given altitudes
 * -> lon, lat by inverting the projection of one camera (table: one record). */
int satba_rpc_fit(int32_t n_cam, int32_t n_samples, const double *target, const double *locs, double h, double tol,
                  int32_t max_iter, double *tables, double *rmse, int32_t *iters, int32_t device);
int satba_rpc_localization(const double *table, int64_t n, const double *col, const double *row, const double *alt, double *lon,
                           double *lat, int32_t device);

/* satba_rpc_refit replaces the loop of ba_pipeline.save_corrected_rpcs over ba_rpcfit.fit_Rt_corrected_rpc (ba_rpcfit.py:270-345,
 * ba_pipeline.py:406-423) for a batch of cameras, device resident: per camera the n_samples^3 mesh over its crop (+ margin) is built,
 * localised through the original RPC (tables_in: n_cam records), moved by global_transform (3 doubles or NULL), pushed through the
 * corrected projection x = P(R (X - T - C) + C) (rt: n_cam x 9 = Euler angles, T, C; the vector ba_params.reconstruct_vars keeps per
 * camera), fitted (satba_rpc_fit's loop) and checked: the margin (10 px at first) doubles until the convex hull of the mesh
 * re-projected through the FITTED model covers the crop (crops: n_cam x 4 = col0, row0, width, height) or exceeds 1000.
 * alt_ranges: n_cam x 2 (lowest, highest altitude of the mesh).  Outputs: tables_out (n_cam records), margins (n_cam: the last
 * margin), err (n_cam x n_samples^3 reprojection errors of the fitted model, check_errors; may be NULL), locs_out / target_out
 * (the last mesh: lon, lat, alt / col, row; may be NULL).  4 <= n_samples <= 16. */
int satba_rpc_refit(int32_t n_cam, const double *tables_in, const double *rt, const double *crops, const double *alt_ranges,
                    const double *global_transform, int32_t n_samples, double h, double tol, int32_t max_iter, double *tables_out, double *err,
                    double *margins, double *locs_out, double *target_out, int32_t device);

/* ---- inspection entry points (parity tests; not used by the solver loop) */
/* index structures built by satba_problem_create, as int32 arrays (SATBA_LAY_PAIR_OFS: int64): n must equal satba_layout_len */
enum { SATBA_LAY_PERM = 0, SATBA_LAY_RANK, SATBA_LAY_PT_CNT, SATBA_LAY_SLICE_BASE, SATBA_LAY_E_CAM, SATBA_LAY_OBS_POS, SATBA_LAY_CAM_OFS,
       SATBA_LAY_CM_PT, SATBA_LAY_CM_POS, SATBA_LAY_PAIR_OFS, SATBA_LAY_PAIR_PTS, SATBA_LAY_PAIR_PI, SATBA_LAY_PAIR_PJ, SATBA_LAY_PAIR_IJ, SATBA_LAY_CM_IO, SATBA_LAY_IPT_OFS,
       /* the merged records of the weighted / robust runs (affine, perspective; built by the first such linearisation, length -1 before):
        * piece offsets of the records' fixed parts (n_pts + 1: the last one is the all-zero record), of every point's first row scale,
        * the pair lists as (record, distances of the two scales in front of it), the camera-major lists as (record, scale piece), the
        * first camera-major entry of every diagonal item (+ 1 sentinel) */
       SATBA_LAY_W_FIX, SATBA_LAY_SC_OFS, SATBA_LAY_PAIR_REC, SATBA_LAY_PAIR_KK, SATBA_LAY_CM_REC, SATBA_LAY_CM_SC, SATBA_LAY_DG_OFS };
int64_t satba_layout_len(const satba_problem *p, int32_t which);
int satba_get_layout(satba_problem *p, int32_t which, int64_t n, void *host_out);
/* n >= 16 doubles: [0..4] milliseconds since the start of satba_problem_create when the uploads were queued, the layout sizes were
 * known, the ELL + camera-major lists were queued, the pair lists were finished, the handle was complete; [5] padded ELL length,
 * [6] pair-list entries, [7] pair-list chunks, [8] unit weights, [9] camera constants in LDS, [10] RPC tables in LDS,
 * [11] camera sums by (fixed-point) LDS atomics, [12] camera-major sums requested, [13] chunks of the camera-major passes,
 * [14] workgroups of k_linearize, [15] fall-backs from the fixed-point sums so far, [16] satba_solve_lm runs on the
 * device-resident loop (n >= 17), [17] the last Schur + solve front had the factorisation running beside the pair
 * kernel (1), not (0), or that mode is switched off on this handle after a wait timed out (-1; it is tried again
 * after 128, 256, ... sequential fronts) (n >= 18), [18] such time-outs so far (n >= 19), [19] diagonal items per (camera,
 * chunk) of the weighted / robust pair kernel, 0 before the merged records exist (n >= 20)                               */
int satba_get_info(const satba_problem *p, double *out, int32_t n);
/* normal-equation blocks of the last linearize: U (M n_p n_p), g_c (M n_p) as written to the exchange
 * payload, V (N x 6: xx xy xz yy yz zz), g_p (N x 3), points in the caller's order. Any pointer may be NULL.
 * U is formed by a camera-major pass (the solver itself only needs diag U_c before the Schur phase).       */
int satba_get_blocks(satba_problem *p, double *U, double *gc, double *V, double *gp);
/* materialised, weighted, row-scaled Jacobian blocks at x: Jc (K x 2 x n_p), Jp (K x 2 x 3).            */
int satba_get_jacobian(satba_problem *p, double *Jc, double *Jp);
/* copy n doubles of the exchange buffer starting at `offset`.                                            */
int satba_get_exchange(satba_problem *p, int64_t offset, int64_t n, double *host_out);
int satba_set_exchange(satba_problem *p, int64_t offset, int64_t n, const double *host_in);
/* state vectors, n_cam*n_params + 3*n_pts doubles: 0 g, 1 scale_inv, 2 gn_h, 3 q1, 4 w, 5 x_new, 6 g_h  */
int satba_get_vector(satba_problem *p, int32_t which, double *host_out);

/* ---- measurement: average duration in milliseconds of `reps` back-to-back launches of one phase's
 * dominant kernel, bracketed by HIP events on the handle's stream.
 * phase: 0 residual kernel, 1 linearize kernel (residual + Jacobian -> normal blocks), 2 Schur kernels,
 *        3 dense Cholesky solve, 4 back-substitution, 5 the Jacobian-vector product of the prepare phase */
int satba_time_kernel(satba_problem *p, int32_t phase, int32_t reps, float *ms_avg);

/* ---- measurement inside a running solve: with on != 0 every satba_linearize brackets its k_linearize launch with two HIP
 * events on the handle's stream (up to 4096 launches are kept); satba_profile_read waits for the stream, returns the number of
 * bracketed launches and the sum of their durations in milliseconds since the last read, and clears the record.  bench.py's
 * roofline figure is this average over the timed LM iterations. */
int satba_profile_linearize(satba_problem *p, int32_t on);
int satba_profile_read(satba_problem *p, int64_t *n_launches, double *ms_total);

#ifdef __cplusplus
}
#endif
#endif /* SATBA_H */
