"""
Outlier rejection between the soft-L1 and the L2 solve (the names of ref:bundle_adjust/ba_outliers.py).

`get_elbow_value` and `compute_obs_to_remove` keep the reference's signatures and return values.  The per-camera work of
`compute_obs_to_remove` -- reprojection errors grouped by camera, sort, elbow of the sorted curve, comparison -- runs on the
device (satba_outliers, csrc/satba_outliers.h) and is index-exact against the reference (tests/golden/outliers_*.npz);
`compute_obs_mask` is the same without the dense correspondence matrix (problems built by
BundleAdjustmentParameters.from_observations have none); `reset_ba_params_after_outlier_removal` and `rm_outliers` rebuild the
parameters the way ref:bundle_adjust/ba_outliers.py:61-109, 158-185 do: the surviving tracks are re-triangulated from the
observations left (satba.ft_triangulate, on the device) unless the caller hands over coordinates to carry on with (`pts3d=`), and
`rm_outliers` returns `p` itself when nothing was detected.
"""
import os

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


def _tracks_with_a_listed_pair(pts_ind, cam_ind, n_pts, n_cam, pairs_to_triangulate):
    """
    has_pair[q]: track q is seen by both cameras of at least one pair (i, j), i < j, of pairs_to_triangulate -- the test of
    ref:bundle_adjust/feature_tracks/ft_utils.py:37-62 (`filter_C_using_pairs_to_triangulate`), which only intersects with the
    track's pairs written with i < j: a pair listed as (j, i) never matches there, and does not here.
    Host version (small problems, dense C): vectorised over the observations, one pass per offset between the two observations
    of a pair inside a track.  Large problems take the count of applicable pairs that the device triangulation returns per
    track (rm_outliers).
    """
    table = np.zeros((n_cam, n_cam), dtype=bool)
    for a, b in pairs_to_triangulate:
        if 0 <= a < b < n_cam:
            table[a, b] = True
    has_pair = np.zeros(n_pts, dtype=bool)
    if not table.any() or pts_ind.size == 0:
        return has_pair
    # observations are point-major with cameras ascending: partner (q, b) of an observation (q, a) with table[a, b] exists iff
    # some later observation of the same track has a listed partner camera.  Tracks are short (<= n_cam): walk the offsets
    # k = 1, 2, ... between the two observations of a pair, vectorised over all observations at once.
    K = pts_ind.size
    max_len = int(np.bincount(pts_ind, minlength=n_pts).max())
    for k in range(1, max_len):
        same = pts_ind[k:] == pts_ind[: K - k]
        hit = same & table[cam_ind[: K - k], cam_ind[k:]]
        has_pair[pts_ind[: K - k][hit]] = True
    return has_pair


def _surviving_tracks(pts_ind, cam_ind, n_pts, n_cam, pairs_to_triangulate):
    """Tracks that keep two or more observations (ref:bundle_adjust/ba_outliers.py:74-76) and a listed pair (:79-82)."""
    ok = np.bincount(pts_ind, minlength=n_pts) >= 2
    ok &= _tracks_with_a_listed_pair(pts_ind, cam_ind, n_pts, n_cam, pairs_to_triangulate)
    return ok


def _options_like(p, n_pts_fix, verbose):
    return {"n_cam_fix": int(p.n_cam_fix), "n_pts_fix": int(n_pts_fix), "reduce": False, "verbose": verbose,
            "correction_params": p.cam_params_to_optimize, "ref_cam_weight": p.ref_cam_weight}


def reset_ba_params_after_outlier_removal(C_new, p, verbose=True, pts3d=None):
    """
    ref:bundle_adjust/ba_outliers.py:61-109: new BundleAdjustmentParameters coherent with the correspondence matrix C_new (p.C with
    some observations blanked).  Tracks left with fewer than two observations, or without any pair of pairs_to_triangulate, are
    dropped; fixed points that survive stay fixed and first; the survivors are re-triangulated from the observations left
    (satba.ft_triangulate.init_pts3d, on the device; fixed points keep p.pts3d).  The returned object carries the filtered C,
    like the reference's.  pts3d (not in the reference): (N, 3) coordinates to carry over instead of re-triangulating.
    """
    from .ba_params import BundleAdjustmentParameters, observations_from_C
    from .ft_triangulate import init_pts3d

    C_new = np.asarray(C_new)
    pts_ind, cam_ind, _ = observations_from_C(C_new)
    left = np.nonzero(_surviving_tracks(pts_ind, cam_ind, C_new.shape[1], C_new.shape[0] // 2, p.pairs_to_triangulate))[0]
    C_left = C_new[:, left]
    n_fix_new = int(np.sum(left < p.n_pts_fix))
    if pts3d is None:
        pts = init_pts3d(C_left, p.cameras, p.cam_model, p.pairs_to_triangulate, verbose=verbose)
        if n_fix_new > 0:
            pts[:n_fix_new] = np.asarray(p.pts3d)[left[:n_fix_new]]
    else:
        pts = np.asarray(pts3d)[left]
    new_p = BundleAdjustmentParameters(C_left, pts, p.cameras, p.cam_model, p.pairs_to_triangulate, p.camera_centers,
                                       _options_like(p, n_fix_new, verbose))
    new_p.pts_prev_indices = np.asarray(p.pts_prev_indices)[left]
    return new_p


def rm_outliers(err, p, predef_thr=None, min_thr=1.0, verbose=False, pts3d=None):
    """
    New BundleAdjustmentParameters without the outlier observations (ref:bundle_adjust/ba_outliers.py:158-185).  Nothing detected:
    `p` itself is returned, as the reference does (:170-173).  Otherwise the parameters are rebuilt the way
    reset_ba_params_after_outlier_removal does -- through the dense correspondence matrix when p has one (the returned object then
    has `.C` too, which ref:bundle_adjust/ba_pipeline.py:582-591 and a second rm_outliers call read), from the observation lists
    when p was built by BundleAdjustmentParameters.from_observations (no C: 3.2 GB at the headline shape).
    pts3d (not in the reference): (N, 3) coordinates to carry over instead of re-triangulating the surviving tracks.
    """
    from .ba_params import BundleAdjustmentParameters

    remove, cam_thr, n = compute_obs_mask(err, p, predef_thr, min_thr)
    if n == 0:
        new_p = p
    elif p.C is not None:
        C_new = p.C.copy()
        C_new[2 * p.cam_ind[remove], p.pts_ind[remove]] = np.nan
        C_new[2 * p.cam_ind[remove] + 1, p.pts_ind[remove]] = np.nan
        new_p = reset_ba_params_after_outlier_removal(C_new, p, verbose=verbose, pts3d=pts3d)
    else:
        keep = ~remove
        pts_ind, cam_ind, pts2d = p.pts_ind[keep], p.cam_ind[keep], p.pts2d[keep]
        two = np.bincount(pts_ind, minlength=p.n_pts) >= 2
        pairs = list(p.pairs_to_triangulate)
        canonical = all(0 <= a < b < p.n_cam for a, b in pairs)
        pts = None
        from . import ba_core

        if pts3d is None and canonical and not os.environ.get("SATBA_TRI_UPLOAD") and ba_core.cached_engine(p) is not None:
            # one device call on the handle's RESIDENT observations (the mask goes up, the tracks do not go up again) gives both the
            # re-triangulated points and, per track, how many listed pairs apply to it: a track survives iff that count is positive
            # (pairs written i < j, as filter_C_using_pairs_to_triangulate requires)
            from .ft_triangulate import init_pts3d_resident

            pts_all, info = init_pts3d_resident(p, pairs, remove=remove, return_info=True)
            ok = info["n_tri"] > 0
            pts = pts_all[ok]
        elif pts3d is None and canonical:
            # no single-rank engine holds p's tracks (a sharded run caches only its (rank, world) engines; or nothing has run on p
            # yet), or SATBA_TRI_UPLOAD=1 (tests): the same through the stand-alone entry point, which uploads the surviving tracks
            from .ft_triangulate import init_pts3d_from_observations

            cand = np.nonzero(two)[0]
            idx = np.full(p.n_pts, -1, dtype=np.int64)
            idx[cand] = np.arange(cand.size)
            s2 = two[pts_ind]
            pts_c, info = init_pts3d_from_observations(idx[pts_ind[s2]], cam_ind[s2], pts2d[s2], cand.size, p.cameras, p.cam_model, pairs,
                                                       return_info=True)
            has = info["n_tri"] > 0
            ok = np.zeros(p.n_pts, dtype=bool)
            ok[cand[has]] = True
            pts = pts_c[has]
        else:
            ok = two & _tracks_with_a_listed_pair(pts_ind, cam_ind, p.n_pts, p.n_cam, pairs)
        left = np.nonzero(ok)[0]
        new_index = np.full(p.n_pts, -1, dtype=np.int64)
        new_index[left] = np.arange(left.size)
        sel = ok[pts_ind]
        n_fix_new = int(np.sum(left < p.n_pts_fix))
        if pts3d is not None:
            pts = np.asarray(pts3d)[left]
        elif pts is None:
            from .ft_triangulate import init_pts3d_from_observations

            pts = init_pts3d_from_observations(new_index[pts_ind[sel]], cam_ind[sel], pts2d[sel], left.size, p.cameras, p.cam_model, pairs)
        if pts3d is None and n_fix_new > 0:
            pts[:n_fix_new] = np.asarray(p.pts3d)[left[:n_fix_new]]
        new_p = BundleAdjustmentParameters.from_observations(new_index[pts_ind[sel]], cam_ind[sel], pts2d[sel], pts, p.cameras, p.cam_model,
                                                             p.pairs_to_triangulate, p.camera_centers, _options_like(p, n_fix_new, verbose))
        new_p.pts_prev_indices = np.asarray(p.pts_prev_indices)[left]
    if verbose:
        n_rm_tracks = p.n_pts - new_p.n_pts
        print("Reprojection error threshold per camera: {} px".format(cam_thr))
        print("Deleted {} observations ({:.2f}%) and {} tracks ({:.2f}%)".format(
            n, 100.0 * n / max(p.n_obs, 1), n_rm_tracks, 100.0 * n_rm_tracks / max(p.n_pts, 1)))
    return new_p
