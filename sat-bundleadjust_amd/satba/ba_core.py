"""
Bundle-adjustment core on MI355X: drop-in for the names of ref:bundle_adjust/ba_core.py.

`fun` and `run_ba_optimization` keep the reference's signatures and return values but run on the GPU:
residuals, analytic Jacobian blocks, Schur complement and the trust-region loop are HIP kernels behind
libsatba_hip.so (include/satba.h), driven by satba/trf.py.  Nothing here falls back to the CPU: without the
library or a HIP device these two functions raise.

The projection helpers (`rotate_euler`, `project_*`, `adjust_pts3d`) are kept as small numpy functions because
code outside the hot path calls them on a handful of points (e.g. ref:bundle_adjust/ba_rpcfit.py:331); the
device path does not use them.

Differences from the reference, all deliberate:
  * the Jacobian is analytic; `build_jacobian_sparsity` is kept for API compatibility only;
  * `ls_params` accepts three extra optional keys -- "gtol" (default 1e-8, scipy's default), "rpc_store_f32"
    (default True: round RPC projections to float32 like ref:bundle_adjust/ba_core.py:150) and "return_result"
    -- without changing any default of ref:bundle_adjust/ba_core.py:233-234;
  * when torch.distributed is initialised with world_size > 1, `run_ba_optimization` shards the points over
    the ranks (satba/sharding.py) and every rank returns the full vectors;
  * the figure helpers (`save_histogram_of_errors`, `save_heatmap_of_reprojection_error`, `idw_interpolation`,
    ref:bundle_adjust/ba_core.py:373-567) keep their signatures but are self-contained: UTM coordinates come from
    satba/geo_utils.py and footprint masks from matplotlib paths instead of pyproj / utm / shapely; `plots=True` of
    `run_ba_optimization` is accepted and ignored (the reference opens an interactive window there).
"""
import os
import time

import numpy as np

from . import sharding, trf
from .loader import display_dict, flush_print

_ENGINE_ATTR = "_satba_engines"


# ----------------------------------------------------------------------------- host projection helpers

def rotate_euler(pts, euler_angles):
    """Rotate each row of pts by R = Rz Ry Rx of the matching row of euler_angles (ref:bundle_adjust/ba_core.py:36-56)."""
    c, s = np.cos(euler_angles), np.sin(euler_angles)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    y, z = c[:, 0] * y - s[:, 0] * z, s[:, 0] * y + c[:, 0] * z
    x, z = c[:, 1] * x + s[:, 1] * z, -s[:, 1] * x + c[:, 1] * z
    x, y = c[:, 2] * x - s[:, 2] * y, s[:, 2] * x + c[:, 2] * y
    return np.column_stack((x, y, z))


def project_affine(pts3d, cam_params, pts_ind, cam_ind):
    """(col, row) of each observation under affine cameras [a, b, g, t0, t1, fx, fy, skew] (ref:bundle_adjust/ba_core.py:59-81)."""
    cp = cam_params[cam_ind]
    q = rotate_euler(pts3d[pts_ind], cp[:, :3])[:, :2] + cp[:, 3:5]
    return np.column_stack((cp[:, 5] * q[:, 0] + cp[:, 7] * q[:, 1], cp[:, 6] * q[:, 1]))


def project_perspective(pts3d, cam_params, pts_ind, cam_ind):
    """Perspective cameras [a, b, g, t0, t1, t2, fx, fy, skew, cx, cy] (ref:bundle_adjust/ba_core.py:84-107)."""
    cp = cam_params[cam_ind]
    q = rotate_euler(pts3d[pts_ind], cp[:, :3]) + cp[:, 3:6]
    u = cp[:, 6] * q[:, 0] + cp[:, 8] * q[:, 1] + cp[:, 9] * q[:, 2]
    v = cp[:, 7] * q[:, 1] + cp[:, 10] * q[:, 2]
    return np.column_stack((u, v)) / q[:, 2:3]


def adjust_pts3d(pts3d, Rt_vec):
    """X' = R (X - T - C) + C with rows [angles, T, C] (ref:bundle_adjust/ba_core.py:110-130)."""
    C = Rt_vec[:, 6:9]
    return rotate_euler(pts3d - Rt_vec[:, 3:6] - C, Rt_vec[:, :3]) + C


def project_rpc(pts3d, rpcs, cam_params, pts_ind, cam_ind):
    """Corrected points through each camera's RPC, float32 like the reference (ref:bundle_adjust/ba_core.py:133-154)."""
    from .cam_utils import apply_rpc_projection

    X = adjust_pts3d(pts3d[pts_ind], cam_params[cam_ind])
    out = np.zeros((pts_ind.shape[0], 2), dtype=np.float32)
    order = np.argsort(cam_ind, kind="stable")
    bounds = np.searchsorted(cam_ind[order], np.arange(len(rpcs) + 1))
    for c in range(len(rpcs)):
        sel = order[bounds[c]: bounds[c + 1]]
        if sel.size:
            out[sel] = apply_rpc_projection(rpcs[c], X[sel])
    return out


# ----------------------------------------------------------------------------- device plumbing

_SUM_POOL = None
_SUM_CHUNK = 1 << 20  # elements


def _pool():
    global _SUM_POOL
    if _SUM_POOL is None:
        from concurrent.futures import ThreadPoolExecutor

        _SUM_POOL = ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1), thread_name_prefix="satba-host")
    return _SUM_POOL


def _content_sum(a):
    """Sum of a contiguous array's entries, over fixed 8 MB chunks on a few threads (numpy releases the GIL inside a sum): the
    content part of the cache key below.  At 200 x 1M x 10M the three sums over 240 MB were 9 ms of a 45 ms call (round 5); the
    chunking is fixed, so the value is a function of the content alone."""
    a = np.asarray(a)
    if a.size < 4 * _SUM_CHUNK or not a.flags.c_contiguous:
        return float(a.sum())
    flat = a.reshape(-1)
    parts = list(_pool().map(lambda i: float(flat[i:i + _SUM_CHUNK].sum()), range(0, flat.size, _SUM_CHUNK)))
    return float(np.sum(parts))


def _big_copy(a):
    """a.copy() for the large vectors of the drop-in call, in chunks on the threads of _content_sum (a 24 MB copy is 2.5 ms on one)."""
    a = np.asarray(a)
    if a.size < 2 * _SUM_CHUNK or not a.flags.c_contiguous or a.ndim != 1:
        return a.copy()
    out = np.empty_like(a)
    step = _SUM_CHUNK // 2
    list(_pool().map(lambda i: np.copyto(out[i:i + step], a[i:i + step]), range(0, a.size, step)))
    return out


def _fingerprint(p):
    return (p.cam_model, p.n_cam, p.n_pts, p.n_obs, p.n_params, int(p.n_cam_fix), int(p.n_pts_fix),
            id(p.pts_ind), id(p.cam_ind), id(p.pts2d), id(p.pts2d_w), id(p.cam_params),
            _content_sum(p.pts2d_w), _content_sum(p.pts2d), float(p.cam_params.sum()), float(np.sum(p.pts3d[: int(p.n_pts_fix)])))


def _distributed():
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return trf.TorchComm()
    except ImportError:
        pass
    return trf.SingleComm()


def get_engine(p, comm=None, rpc_f32=True):
    """The (cached) device engine of a BundleAdjustmentParameters object for this rank's shard."""
    from .engine_hip import HipEngine

    comm = comm or trf.SingleComm()
    cache = p.__dict__.setdefault(_ENGINE_ATTR, {})
    fp = _fingerprint(p)
    for k in [k for k in cache if k[1] != fp]:  # the object was modified: drop engines built from the old state
        cache.pop(k).close()
    key = ((comm.rank, comm.world, bool(rpc_f32)), fp)
    if key not in cache:
        cache[key] = HipEngine(p, sharding.make_shard(p, comm.rank, comm.world), rpc_f32=rpc_f32)
    return cache[key]


def cached_engine(p, world=1, rpc_f32=None):
    """The engine get_engine has already built for p's current state with `world` ranks (rank 0), or None -- for callers that can
    use a resident whole-problem handle but should not build (and upload) one just for themselves."""
    cache = p.__dict__.get(_ENGINE_ATTR, {})
    fp = _fingerprint(p)
    for (rank, w, f32), key_fp in cache:
        if key_fp == fp and rank == 0 and w == world and (rpc_f32 is None or f32 == bool(rpc_f32)):
            return cache[((rank, w, f32), key_fp)]
    return None


def _frozen_vars(v, p):
    """
    Variable vector with frozen cameras / points restored to their initial values, as
    BundleAdjustmentParameters.get_vars_ready_for_fun does (ref:bundle_adjust/ba_params.py:240-249) -- including
    the reference's side effect of writing the frozen camera rows into the caller's v.
    """
    n_c = p.n_cam * p.n_params
    if p.n_cam_fix > 0:
        v[: p.n_cam_fix * p.n_params] = p.cam_params[: p.n_cam_fix, : p.n_params].ravel()
    if p.n_pts_fix > 0:
        v = v.copy()
        v[n_c: n_c + 3 * int(p.n_pts_fix)] = np.asarray(p.pts3d[: int(p.n_pts_fix)], dtype=np.float64).ravel()
    return v


def fun(v, p):
    """
    Bundle-adjustment residuals [x0, y0, x1, y1, ...] = w * (projection - observation), float64, on the GPU
    (ref:bundle_adjust/ba_core.py:157-183).  Always evaluates the whole problem on this process's device.
    """
    v = np.asarray(v)
    if v.dtype != np.float64 or not v.flags.c_contiguous:
        v = np.ascontiguousarray(v, dtype=np.float64)
    eng = get_engine(p)
    eng.configure("linear", 1.0)
    eng.set_x(_frozen_vars(v, p))
    return eng.residuals()


def build_jacobian_sparsity(p):
    """
    Sparsity structure of the Jacobian (ref:bundle_adjust/ba_core.py:186-219): rows 2k, 2k+1 have ones in the
    n_params columns of observation k's camera and the 3 columns of its point.  The device solver does not need
    it (the Jacobian blocks are analytic and never assembled); kept for callers that expect the function.
    Returned as CSR rather than LIL.
    """
    from scipy.sparse import csr_matrix

    K, n_p = p.pts_ind.size, p.n_params
    cols = np.hstack((p.cam_ind[:, None] * n_p + np.arange(n_p), p.n_cam * n_p + p.pts_ind[:, None] * 3 + np.arange(3)))
    cols = np.repeat(cols, 2, axis=0).ravel()
    indptr = np.arange(0, 2 * K * (n_p + 3) + 1, n_p + 3)
    return csr_matrix((np.ones(cols.size, dtype=int), cols, indptr), shape=(2 * K, p.n_cam * n_p + p.n_pts * 3))


def init_optimization_config(config=None):
    """Solver options with the reference's defaults (ref:bundle_adjust/ba_core.py:222-241)."""
    out = {"loss": "linear", "ftol": 1e-4, "xtol": 1e-10, "f_scale": 1.0, "max_iter": 300, "verbose": 1}
    if config is not None:
        out.update({k: config[k] for k in out if k in config})
    return out


def run_ba_optimization(p, ls_params=None, verbose=False, plots=True):
    """
    Solve the bundle adjustment problem on the GPU (ref:bundle_adjust/ba_core.py:244-332).

    Returns (vars_init, vars_ba, err_init, err_ba, iterations) where, as in the reference, `iterations` is the
    number of residual evaluations (`nfev`).  `plots` (the reference's default: True) draws the reference's three panels
    (plot_residuals_and_errors) and calls plt.show(), which returns at once on a non-interactive backend.
    """
    extra = ls_params or {}
    cfg = init_optimization_config(ls_params)
    if verbose:
        print("\nRunning bundle adjustment...")
        display_dict(cfg)

    # ls_params["timings"] (a dict, optional): filled with the wall-clock seconds of the parts of this call -- what bench.py reports
    # as `e2e`: the reference times the same region at ref:bundle_adjust/ba_core.py:283-299
    tm = extra.get("timings")
    clock = time.perf_counter
    t_call = clock()
    comm = _distributed()
    eng = get_engine(p, comm, rpc_f32=extra.get("rpc_store_f32", True))
    t_eng = clock()
    vars_init = _big_copy(p.params_opt)
    # the frozen camera rows go into the vector that is uploaded, as the reference writes them into its caller's v; here that vector IS
    # vars_init (no second 24 MB copy at 1 M points) and the few entries are put back behind the upload
    n_fix = int(p.n_cam_fix) * int(p.n_params)
    own = vars_init.dtype == np.float64 and vars_init.flags.c_contiguous and int(p.n_pts_fix) == 0
    x0 = _frozen_vars(vars_init if own else np.array(vars_init, dtype=np.float64), p)
    eng.configure("linear", 1.0)
    eng.set_x(eng.shard.local_x(p, x0))
    if own and n_fix:
        vars_init[:n_fix] = np.asarray(p.params_opt[:n_fix])
    # The residual VECTORS are only needed for the figure and for return_result; the five values the reference returns need the
    # per-observation errors, which the device forms from its own residuals with the reference's operations and roundings
    # (satba_reprojection_errors): half the bytes over the bus and no numpy passes over 2 K doubles (C4: 0.2 -> 0.05 s per call)
    need_r = bool(plots) or bool(extra.get("return_result", False)) or not hasattr(eng, "reprojection_errors")
    residuals_init = sharding.assemble_residuals(p, eng.shard, eng.residuals(), comm) if need_r else None
    # Nothing on the device needs the initial errors again: from 8 MB on (1 M observations) their download runs in a second thread
    # beside the solve (engine_hip.reprojection_errors_begin / _fetch; at 200 x 1M x 10M 80 MB over the bus under a 7 ms solve).
    # SATBA_ERR_OVERLAP=0 keeps the one-step call.
    err_pending = None
    if (not need_r and hasattr(eng, "reprojection_errors_fetch") and eng.n_obs >= (1 << 20)
            and os.environ.get("SATBA_ERR_OVERLAP", "1") != "0"):
        import threading

        eng.reprojection_errors_begin()
        box = {}

        def _fetch():
            try:
                box["err"] = eng.reprojection_errors_fetch()
            except BaseException as e:  # noqa: B902  (re-raised in the calling thread below)
                box["exc"] = e
        err_pending = threading.Thread(target=_fetch, name="satba-err-init")
        err_pending.start()
        err_init = None
    else:
        err_init = None if need_r else sharding.assemble_residuals(p, eng.shard, eng.reprojection_errors(), comm)
    t_init = clock()
    if verbose:
        flush_print("Shape of Jacobian sparsity: {}x{}".format(2 * p.n_obs, p.n_cam * p.n_params + 3 * p.n_pts))

    t0 = time.time()
    try:
        res = trf.trf_solve(eng, comm, ftol=cfg["ftol"], xtol=cfg["xtol"], gtol=extra.get("gtol", 1e-8),
                            max_nfev=cfg["max_iter"], loss=cfg["loss"], f_scale=cfg["f_scale"],
                            verbose=cfg["verbose"] if comm.rank == 0 else 0)
    finally:
        if err_pending is not None:
            err_pending.join()  # (before the next transfer of this handle: the copy lanes are shared)
    t_solve = clock()
    if err_pending is not None:
        if "exc" in box:
            raise box["exc"]
        err_init = sharding.assemble_residuals(p, eng.shard, box["err"], comm)
    vars_ba = sharding.assemble_x(p, eng.shard, eng.get_x(), comm)
    residuals_ba = sharding.assemble_residuals(p, eng.shard, eng.residuals(), comm) if need_r else None
    err_ba = None if need_r else sharding.assemble_residuals(p, eng.shard, eng.reprojection_errors(), comm)
    t_back = clock()
    if verbose:
        flush_print("Optimization took {:.2f} seconds\n".format(time.time() - t0))

    iterations = res.nfev
    if need_r:
        err_init = compute_reprojection_error(residuals_init, p.pts2d_w)
        err_ba = compute_reprojection_error(residuals_ba, p.pts2d_w)
    if tm is not None:
        tm.update(engine_s=t_eng - t_call, initial_residuals_s=t_init - t_eng, solve_s=t_solve - t_init, read_back_s=t_back - t_solve,
                  host_errors_s=clock() - t_back, nfev=int(res.nfev))
    if verbose:
        flush_print("Reprojection error before BA (mean / median): {:.2f} / {:.2f}".format(
            np.mean(err_init), np.median(err_init)))
        flush_print("Reprojection error after  BA (mean / median): {:.2f} / {:.2f}\n".format(
            np.mean(err_ba), np.median(err_ba)))
        n_per_cam = np.bincount(p.cam_ind, minlength=p.n_cam)
        mean_init = np.bincount(p.cam_ind, weights=err_init, minlength=p.n_cam) / np.maximum(n_per_cam, 1)
        mean_ba = np.bincount(p.cam_ind, weights=err_ba, minlength=p.n_cam) / np.maximum(n_per_cam, 1)
        for c in range(p.n_cam):
            flush_print("    - cam {:3} - {:5} obs - (mean before / mean after): {:.2f} / {:.2f}".format(
                c, n_per_cam[c], mean_init[c], mean_ba[c]))
        print("\n")

    if plots and comm.rank == 0:
        plot_residuals_and_errors(residuals_init, residuals_ba, err_init, err_ba)
    if tm is not None:
        tm["total_s"] = clock() - t_call

    if extra.get("return_result", False):
        res["x"], res["fun"] = vars_ba, residuals_ba
        return vars_init, vars_ba, err_init, err_ba, iterations, res
    return vars_init, vars_ba, err_init, err_ba, iterations


def compute_reprojection_error(residuals, pts2d_w=None):
    """Per-observation reprojection error: L2 norm of the unweighted residual pair (ref:bundle_adjust/ba_core.py:335-349)."""
    r = np.asarray(residuals).reshape(-1, 2)
    if pts2d_w is not None:
        r = r / np.asarray(pts2d_w)[:, None]
    return np.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1])  # the reference's operations, bit for bit (not np.hypot)


def compute_mean_reprojection_error_per_track(err, pts_ind, cam_ind):
    """Mean reprojection error of each track, float32 (ref:bundle_adjust/ba_core.py:352-370); tracks without observations give NaN."""
    n_pts = pts_ind.max() + 1
    count = np.bincount(pts_ind, minlength=n_pts)
    total = np.bincount(pts_ind, weights=err, minlength=n_pts)
    with np.errstate(invalid="ignore", divide="ignore"):
        return (total / count).astype(np.float32)


# ----------------------------------------------------------------------------- figures (host only; never on the solver path)

def _pyplot():
    import matplotlib

    if not matplotlib.get_backend():  # pragma: no cover
        matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    return plt


def plot_residuals_and_errors(residuals_init, residuals_ba, err_init, err_ba, show=True, max_points=2000000):
    """
    The figure at the end of the reference's run_ba_optimization (ref:bundle_adjust/ba_core.py:321-330): residual vectors before and
    after, and the two histograms of the reprojection error (40 bins; the second on the range of the first).  Beyond max_points
    residuals the first panel draws every n-th one (the reference draws all of them -- 20 million line segments at the headline
    size); the histograms always see every observation.  Returns the figure.
    """
    plt = _pyplot()
    residuals_init, residuals_ba = np.asarray(residuals_init), np.asarray(residuals_ba)
    step = max(1, int(np.ceil(residuals_init.size / max_points)))
    fig, f = plt.subplots(1, 3, figsize=(15, 3))
    f[0].plot(residuals_init[::step])
    f[0].plot(residuals_ba[::step])
    f[0].title.set_text("Residuals before and after BA")
    f[1].hist(err_init, bins=40)
    f[1].title.set_text("Reprojection error before BA")
    f[2].hist(err_ba, bins=40, range=(err_init.min(), err_init.max()) if np.size(err_init) else None)
    f[2].title.set_text("Reprojection error after BA")
    if show:
        plt.show()
        plt.close(fig)  # (a pipeline calls this once per solve)
    return fig


def save_histogram_of_errors(img_path, err_init, err_ba, plot=False):
    """
    Histograms of the reprojection error of every observation before and after the adjustment, on a common range
    (ref:bundle_adjust/ba_core.py:376-401).  Writes img_path unless plot is set.
    """
    import os

    plt = _pyplot()
    err_init, err_ba = np.asarray(err_init, dtype=np.float64), np.asarray(err_ba, dtype=np.float64)
    lim = (float(err_init.min()), float(err_init.max())) if err_init.size else (0.0, 1.0)
    fig, axes = plt.subplots(1, 2, figsize=(12, 3))
    for ax, e, title, rng in ((axes[0], err_init, "Before BA", None), (axes[1], err_ba, "After BA", lim)):
        ax.hist(e, bins=40, range=rng)
        ax.set_title(title)
        ax.set_xlabel("Reprojection error (pixel units)")
        ax.set_ylabel("Number of tie point observations")
    if plot:
        plt.show()
    else:
        if os.path.dirname(img_path):
            os.makedirs(os.path.dirname(img_path), exist_ok=True)
        fig.savefig(img_path, bbox_inches="tight")
    plt.close(fig)


def idw_interpolation(pts2d, z, pts2d_query, N=8):
    """
    Inverse-distance-weighted average of the N nearest known points for every query point
    (ref:bundle_adjust/ba_core.py:525-567): z(q) = sum(z_i / d_i) / sum(1 / d_i); a query that coincides with a known
    point (d < 1e-10) takes that point's value; N = 1 is nearest-neighbour interpolation.
    """
    from scipy.spatial import cKDTree

    pts2d, z = np.asarray(pts2d, dtype=np.float64), np.asarray(z)
    N = int(min(N, pts2d.shape[0]))
    dist, idx = cKDTree(pts2d).query(np.asarray(pts2d_query, dtype=np.float64), k=N)
    if N == 1:
        return z[idx]
    with np.errstate(divide="ignore", invalid="ignore"):
        w = 1.0 / dist
        out = np.sum(w * z[idx], axis=1) / np.sum(w, axis=1)
    hit = dist[:, 0] < 1e-10
    out[hit] = z[idx[hit, 0]]
    return out


def _ring(geojson):
    """Outer ring (n, 2) of a geojson polygon."""
    return np.asarray(geojson["coordinates"][0], dtype=np.float64)


def save_heatmap_of_reprojection_error(img_path, p, err, input_ims_footprints_lonlat, aoi_lonlat_roi=None, plot=False,
                                       smooth=20, global_transform=None):
    """
    Mean reprojection error per track, interpolated (IDW, then a Gaussian of sigma `smooth` pixels) over the union of the
    image footprints on a UTM grid of at most 1000 pixels per side, with the tie points drawn on top
    (ref:bundle_adjust/ba_core.py:404-522).  `p` needs pts_ind, cam_ind and pts3d_ba (set by reconstruct_vars; pts3d is
    used before that).  A .png path gets the figure; a .tif path gets the georeferenced raster (needs rasterio).
    """
    import os

    from scipy.ndimage import gaussian_filter

    from . import geo_utils

    max_size = 1000
    rings = [_ring(g) for g in input_ims_footprints_lonlat]
    all_ll = np.vstack(rings)
    zone = geo_utils.utm_zone_from_lonlat(all_ll[0, 0], all_ll[0, 1])
    south = all_ll[:, 1].mean() < 0

    def to_utm(ll):
        e, n = geo_utils.utm_from_lonlat(ll[:, 0], ll[:, 1], zone)
        return np.column_stack((e, np.where(n < 0, n + 10e6, n)))  # as ref:bundle_adjust/geo_utils.py:72

    rings_utm = [to_utm(r) for r in rings]
    allu = np.vstack(rings_utm)
    xmin, xmax, ymin, ymax = allu[:, 0].min(), allu[:, 0].max(), allu[:, 1].min(), allu[:, 1].max()
    resolution = max(float(max(ymax - ymin, xmax - xmin)) / max_size, 1e-9)
    height, width = int(np.floor((ymax - ymin) / resolution) + 1), int(np.floor((xmax - xmin) / resolution) + 1)

    def to_grid(u):  # column to the east, row to the south
        return np.column_stack(((u[:, 0] - xmin) / resolution, (ymax - u[:, 1]) / resolution))

    track_err = compute_mean_reprojection_error_per_track(err, p.pts_ind, p.cam_ind)
    pts3d = np.array(getattr(p, "pts3d_ba", p.pts3d), dtype=np.float64)[: track_err.size]
    if global_transform is not None:
        pts3d = pts3d - np.asarray(global_transform)
    lat, lon, _ = geo_utils.ecef_to_latlon_custom(pts3d[:, 0], pts3d[:, 1], pts3d[:, 2])
    g = to_grid(to_utm(np.column_stack((lon, lat))))
    ok = (g[:, 0] >= 0) & (g[:, 0] < width) & (g[:, 1] >= 0) & (g[:, 1] < height) & np.isfinite(track_err)
    g, te = g[ok], track_err[ok]

    cols, rows = np.meshgrid(np.arange(width), np.arange(height))
    query = np.column_stack((cols.ravel(), rows.ravel()))
    if te.size:
        raster = idw_interpolation(g, te, query).reshape(height, width)
        if smooth:
            raster = gaussian_filter(raster, sigma=smooth)
    else:
        raster = np.full((height, width), np.nan)
    from matplotlib.path import Path

    inside = np.zeros(height * width, dtype=bool)
    for r in rings_utm:
        inside |= Path(to_grid(r)).contains_points(query)
    raster[~inside.reshape(height, width)] = np.nan

    if os.path.dirname(img_path):
        os.makedirs(os.path.dirname(img_path), exist_ok=True)
    if not plot and os.path.splitext(img_path)[1].lower() == ".tif":
        import rasterio  # not a dependency of the solver: only for this output format
        from rasterio.transform import from_origin

        epsg = geo_utils.epsg_code_from_utm_zone(zone, north=not south)
        with rasterio.open(img_path, "w", driver="GTiff", height=height, width=width, count=1, dtype="float32",
                           crs="EPSG:{}".format(epsg), transform=from_origin(xmin, ymax, resolution, resolution)) as dst:
            dst.write(raster.astype(np.float32), 1)
        return

    plt = _pyplot()
    vmin, vmax = 0.0, 2.0
    fig, ax = plt.subplots(figsize=(10, 10))
    im = ax.imshow(raster, vmin=vmin, vmax=vmax)
    for r in rings_utm:
        ax.plot(*to_grid(r).T, color="black")
    ax.scatter(g[:, 0], g[:, 1], 30, te, edgecolors="k", vmin=vmin, vmax=vmax)
    if aoi_lonlat_roi is not None:
        ax.plot(*to_grid(to_utm(_ring(aoi_lonlat_roi))).T, color="red", linewidth=3.0)
    ax.axis("equal")
    ax.axis("off")
    cbar = fig.colorbar(im, ax=ax, fraction=0.046, pad=0.04)
    ticks = np.linspace(vmin, vmax, 9)
    cbar.set_ticks(ticks)
    cbar.set_ticklabels(["{:.2f}".format(t) for t in ticks[:-1]] + [">={:.2f}".format(vmax)])
    cbar.set_label("Reprojection error across AOI (pixel units)", rotation=270, labelpad=25)
    if plot:
        plt.show()
    else:
        fig.savefig(img_path, bbox_inches="tight")
    plt.close(fig)
