"""
Outlier rejection between the soft-L1 and the L2 solve (the names of ref:bundle_adjust/ba_outliers.py).

`get_elbow_value` and `compute_obs_to_remove` keep the reference's signatures and return values.  The per-camera work of
`compute_obs_to_remove` -- reprojection errors grouped by camera, sort, elbow of the sorted curve, comparison -- runs on the
device (satba_outliers, csrc/satba_outliers.h) and is index-exact against the reference (tests/golden/outliers_*.npz);
`compute_obs_mask` is the same without the dense correspondence matrix (problems built by
BundleAdjustmentParameters.from_observations have none), and `rm_outliers` rebuilds the parameters from the surviving
observations the way ref:bundle_adjust/ba_outliers.py:61-109 does: the surviving tracks are re-triangulated from the observations
left (satba.ft_triangulate, on the device) unless the caller hands over coordinates to carry on with (`pts3d=`).
"""
import numpy as np


def get_elbow_value(err, max_outliers_percent=20, verbose=False):
    """
    Elbow of the sorted values: the one farthest from the chord between the smallest and the largest
    (ref:bundle_adjust/ba_outliers.py:14-58).  Returns (elbow_value, success); success is False when the elbow lies below the
    (100 - max_outliers_percent)-th percentile, i.e. the curve is not L-shaped.  Host version (one vector); the per-camera
    batch runs on the device.
    """
    v = np.sort(np.asarray(err, dtype=np.float64))
    n = v.size
    line = np.array([n - 1.0, v[-1] - v[0]])
    with np.errstate(invalid="ignore", divide="ignore"):
        u = line / np.sqrt(np.sum(line ** 2))
        px, py = np.arange(n, dtype=np.float64), v - v[0]
        sp = px * u[0] + py * u[1]
        d = np.sqrt((px - sp * u[0]) ** 2 + (py - sp * u[1]) ** 2)
    elbow = v[int(np.argmax(d))]
    return elbow, not (elbow < np.percentile(v, 100 - max_outliers_percent))


def compute_obs_mask(err, p, predef_thr=None, min_thr=1.0):
    """
    Per-camera thresholds and the mask of the observations to remove, on the device.
    Returns (remove (K,) bool in the order of p.pts_ind / p.cam_ind, cam_thr list of M floats, n_detected_outliers).
    """
    from . import ba_core

    eng = ba_core.get_engine(p)
    thr, remove, n = eng.outliers(np.asarray(err, dtype=np.float64), predef_thr=predef_thr, min_thr=min_thr)
    return remove, [float(t) for t in thr], n


def compute_obs_to_remove(err, p, predef_thr=None, min_thr=1.0):
    """
    ref:bundle_adjust/ba_outliers.py:112-155: (C_new, cam_thr, n_detected_outliers) with the outlier observations blanked
    (NaN) in a copy of the correspondence matrix p.C.
    """
    remove, cam_thr, n = compute_obs_mask(err, p, predef_thr, min_thr)
    C_new = p.C.copy()
    if n:
        C_new[2 * p.cam_ind[remove], p.pts_ind[remove]] = np.nan
        C_new[2 * p.cam_ind[remove] + 1, p.pts_ind[remove]] = np.nan
    return C_new, cam_thr, n


def rm_outliers(err, p, predef_thr=None, min_thr=1.0, verbose=False, pts3d=None):
    """
    New BundleAdjustmentParameters without the outlier observations (ref:bundle_adjust/ba_outliers.py:158-185 and :61-109):
    tracks left with fewer than two observations, or without any pair of pairs_to_triangulate, are dropped; fixed points
    that survive stay fixed and first.  The surviving tracks are re-triangulated from their remaining observations
    (ref:bundle_adjust/ba_outliers.py:89-93: init_pts3d, fixed points keep p.pts3d); pts3d: (N, 3) coordinates to carry over
    instead (e.g. the points of the solve that has just finished).
    """
    from .ba_params import BundleAdjustmentParameters

    remove, cam_thr, n = compute_obs_mask(err, p, predef_thr, min_thr)
    keep = ~remove
    pts_ind, cam_ind, pts2d = p.pts_ind[keep], p.cam_ind[keep], p.pts2d[keep]
    n_per_track = np.bincount(pts_ind, minlength=p.n_pts)
    ok = n_per_track >= 2
    # a track must still be seen by both cameras of at least one pair to triangulate (ft_utils.py:38-62)
    order = np.argsort(cam_ind, kind="stable")
    bounds = np.searchsorted(cam_ind[order], np.arange(p.n_cam + 1))
    has_pair = np.zeros(p.n_pts, dtype=bool)
    for a, b in p.pairs_to_triangulate:
        pa, pb = pts_ind[order[bounds[a]: bounds[a + 1]]], pts_ind[order[bounds[b]: bounds[b + 1]]]
        has_pair[np.intersect1d(pa, pb, assume_unique=True)] = True
    ok &= has_pair
    left = np.nonzero(ok)[0]
    new_index = np.full(p.n_pts, -1, dtype=np.int64)
    new_index[left] = np.arange(left.size)
    sel = ok[pts_ind]
    n_fix_new = int(np.sum(left < p.n_pts_fix))
    if pts3d is None:
        from .ft_triangulate import init_pts3d_from_observations

        pts = init_pts3d_from_observations(new_index[pts_ind[sel]], cam_ind[sel], pts2d[sel], left.size, p.cameras, p.cam_model,
                                           p.pairs_to_triangulate)
        if n_fix_new > 0:
            pts[:n_fix_new] = np.asarray(p.pts3d)[left[:n_fix_new]]
    else:
        pts = np.asarray(pts3d)[left]
    d = {"n_cam_fix": int(p.n_cam_fix), "n_pts_fix": n_fix_new, "reduce": False, "verbose": verbose,
         "correction_params": p.cam_params_to_optimize, "ref_cam_weight": p.ref_cam_weight}
    new_p = BundleAdjustmentParameters.from_observations(new_index[pts_ind[sel]], cam_ind[sel], pts2d[sel], pts, p.cameras, p.cam_model,
                                                         p.pairs_to_triangulate, p.camera_centers, d)
    new_p.pts_prev_indices = np.asarray(p.pts_prev_indices)[left]
    if verbose:
        print("Deleted {} observations ({:.2f}%) and {} tracks ({:.2f}%)".format(
            n, 100.0 * n / max(p.n_obs, 1), p.n_pts - left.size, 100.0 * (p.n_pts - left.size) / max(p.n_pts, 1)))
        print("     - Reprojection error threshold per camera: {}".format(cam_thr))
    return new_p
