"""
RPC re-fit after the solve (the names of ref:bundle_adjust/ba_rpcfit.py): a new RPC that reproduces the
corrected projection of a camera on a 3-D grid over its image.

`weighted_lsq`, `fit_Rt_corrected_rpc`, `fit_rpc_from_projection_matrix`, `check_errors`, `poly_vect`, `scaling_params` keep the
reference's signatures and return values (RPC objects are `satba.rpc_model.RPCModel`, which carries the attribute names of
`rpcm.RPCModel`).  The fit itself (`satba_rpc_fit`, csrc/satba_rpcfit.h) and the localisation of the grid through the original RPC
(`satba_rpc_localization`) run on the device; `fit_Rt_corrected_rpcs` does the whole loop of the pipeline's output step for all
cameras on the device (`satba_rpc_refit`: meshes, localisation, corrected projection, fit, errors, coverage test).  There is no CPU
fallback.

The coverage test of the reference (`check_correspondences_are_good`: the convex hull of the re-projected grid must contain the
image rectangle, via shapely there) is done with scipy.spatial.ConvexHull here: a convex hull contains a rectangle exactly when it
contains its four corners.
"""
import os

import numpy as np

from . import ba_core, cam_utils, geo_utils
from . import engine_hip as E
from .rpc_model import RPCModel


def poly_vect(x, y, z):
    """ref:bundle_adjust/ba_rpcfit.py:17-44: the 19 non-constant cubic monomials in RPC00B order (y first)."""
    return np.array([y, x, z, y * x, y * z, x * z, y * y, x * x, z * z, x * y * z, y * y * y, y * x * x, y * z * z, y * y * x, x * x * x,
                     x * z * z, y * y * z, x * x * z, z * z * z])


def scaling_params(vect):
    """ref:bundle_adjust/ba_rpcfit.py:156-164: (scale, offset) from the extrema."""
    lo, hi = min(vect), max(vect)
    scale = (hi - lo) / 2
    return scale, lo + scale


def weighted_lsq_batch(targets, input_locs, h=1e-3, tol=1e-2, max_iter=20, return_info=False):
    """weighted_lsq for a batch: targets (M, n, 2) col / row, input_locs (M, n, 3) lon / lat / alt -> list of M RPCModel."""
    lib = E.load_library()
    t = np.ascontiguousarray(targets, dtype=np.float64); x = np.ascontiguousarray(input_locs, dtype=np.float64)
    if t.ndim != 3 or x.ndim != 3 or t.shape[:2] != x.shape[:2] or t.shape[2] != 2 or x.shape[2] != 3:
        raise ValueError("targets must be (M, n, 2) and input_locs (M, n, 3)")
    M, n = t.shape[:2]
    tables = np.zeros((M, 90)); rmse = np.zeros(M); iters = np.zeros(M, dtype=np.int32)
    E._check(lib, lib.satba_rpc_fit(M, n, E._ptr(t), E._ptr(x), float(h), float(tol), int(max_iter), E._ptr(tables), E._ptr(rmse),
                                    E._ptr(iters, E._ip), int(os.environ.get("LOCAL_RANK", "0"))))
    if not np.isfinite(tables).all():  # degenerate samples (a constant coordinate, fewer distinct points than unknowns): the
        raise np.linalg.LinAlgError("Singular matrix")  # reference's numpy.linalg.inv raises the same
    rpcs = [RPCModel.from_table(tab) for tab in tables]
    return (rpcs, {"rmse": rmse, "iters": iters}) if return_info else rpcs


def weighted_lsq(target, input_locs, h=1e-3, tol=1e-2, max_iter=20):
    """
    ref:bundle_adjust/ba_rpcfit.py:88-153: regularised iteratively re-weighted least squares calibrating an RPC model.
    target (N, 2) column / row image coordinates of the N points input_locs (N, 3) lon / lat / alt.  Returns the RPC model.
    """
    return weighted_lsq_batch(np.asarray(target)[None], np.asarray(input_locs)[None], h, tol, max_iter)[0]


def check_errors(rpc_calib, input_locs, target, plot=False):
    """ref:bundle_adjust/ba_rpcfit.py:357-370: reprojection error of the calibrated model per correspondence."""
    col, row = rpc_calib.projection(input_locs[:, 0], input_locs[:, 1], input_locs[:, 2])
    return np.linalg.norm(np.stack([np.asarray(col).reshape(-1), np.asarray(row).reshape(-1)], 1) - target, axis=1)


def check_correspondences_are_good(target, image_corners):
    """ref:bundle_adjust/ba_rpcfit.py:347-355: True when the convex hull of the (N, 2) pixel coordinates covers the image (corners (4, 2))."""
    from scipy.spatial import ConvexHull

    hull = ConvexHull(np.asarray(target, dtype=np.float64))
    eq = hull.equations  # a x + b y + c <= 0 inside
    return bool(np.all(eq[:, :2] @ np.asarray(image_corners, dtype=np.float64).T + eq[:, 2:3] <= 1e-9))


def _grid_through_rpc(original_rpc, crop_offset, alt_range, margin, n_samples, global_transform):
    x0, y0, w, h = crop_offset["col0"], crop_offset["row0"], crop_offset["width"], crop_offset["height"]
    cols, lins, alts = cam_utils.generate_point_mesh([x0 - margin, x0 + w + margin, n_samples], [y0 - margin, y0 + h + margin, n_samples], alt_range)
    lons, lats = original_rpc.localization(cols, lins, alts)
    x, y, z = geo_utils.latlon_to_ecef_custom(lats, lons, alts)
    grid = np.vstack([x, y, z]).T
    pts3d = grid + global_transform if global_transform is not None else grid.copy()
    return grid, pts3d, np.vstack([lons, lats, alts]).T


def _fit_with_growing_margin(make_target, original_rpc, crop_offset, alt_range, n_samples, global_transform):
    """the reference's loop (ba_rpcfit.py:233-267, 308-345): fit on the grid, double the margin until the grid covers the image"""
    x0, y0, w, h = crop_offset["col0"], crop_offset["row0"], crop_offset["width"], crop_offset["height"]
    corners = np.array([[x0, y0], [x0, y0 + h], [x0 + w, y0 + h], [x0 + w, y0]], dtype=np.float64)
    margin = 10
    while True:
        grid, pts3d, input_locs = _grid_through_rpc(original_rpc, crop_offset, alt_range, margin, n_samples, global_transform)
        target = make_target(pts3d)
        rpc_calib = weighted_lsq(target, input_locs)
        err = check_errors(rpc_calib, input_locs, target)
        covered = check_correspondences_are_good(cam_utils.apply_rpc_projection(rpc_calib, grid), corners)
        if margin > 1000 or covered:
            return rpc_calib, err, margin
        margin *= 2


def fit_Rt_corrected_rpc(Rt_vec, global_transform, original_rpc, crop_offset, pts3d_ba, n_samples=10):
    """
    ref:bundle_adjust/ba_rpcfit.py:270-345: RPC of the corrected mapping x = P(R (X - T - C) + C), P the original RPC's projection.
    Rt_vec (1, 9) = [Euler angles, T, C]; crop_offset dict col0 / row0 / width / height; pts3d_ba (N, 3) ECEF points of the area
    (only their median altitude is looked at, for the reference's warning).  Returns (rpc_calib, err, margin).
    """
    pts = np.asarray(pts3d_ba, dtype=np.float64)
    pts = pts - global_transform if global_transform is not None else pts
    _, _, alts = geo_utils.ecef_to_latlon_custom(pts[:, 0], pts[:, 1], pts[:, 2])
    dev = abs(original_rpc.alt_offset - np.median(alts))
    if dev > 5:
        print("warning: median altitude of bundle adjustment points is {:.2f} meters deviated from the original rpc alt_offset".format(dev))
    alt_range = [original_rpc.alt_offset - original_rpc.alt_scale, original_rpc.alt_offset + original_rpc.alt_scale, n_samples]
    Rt = np.asarray(Rt_vec, dtype=np.float64).reshape(1, 9)
    return _fit_with_growing_margin(lambda X: cam_utils.apply_rpc_projection(original_rpc, ba_core.adjust_pts3d(X, Rt)), original_rpc, crop_offset,
                                    alt_range, n_samples, global_transform)


def fit_rpc_from_projection_matrix(P, global_transform, original_rpc, crop_offset, pts3d_ba, n_samples=10):
    """
    ref:bundle_adjust/ba_rpcfit.py:201-267: RPC copying the 3 x 4 projection matrix P (which maps to crop coordinates) over the image.
    Returns (rpc_calib, err, margin).
    """
    pts = np.asarray(pts3d_ba, dtype=np.float64)
    pts = pts - global_transform if global_transform is not None else pts
    _, _, alts = geo_utils.ecef_to_latlon_custom(pts[:, 0], pts[:, 1], pts[:, 2])
    alt_offset, alt_scale = np.median(alts), max(8000, original_rpc.alt_scale)
    alt_range = [alt_offset - alt_scale, alt_offset + alt_scale, n_samples]
    shift = np.array([crop_offset["col0"], crop_offset["row0"]], dtype=np.float64)
    return _fit_with_growing_margin(lambda X: cam_utils.apply_projection_matrix(np.asarray(P, dtype=np.float64), X) + shift, original_rpc,
                                    crop_offset, alt_range, n_samples, global_transform)


def fit_Rt_corrected_rpcs(Rt_vecs, global_transform, original_rpcs, crop_offsets, n_samples=10, return_info=False):
    """
    fit_Rt_corrected_rpc for a list of cameras, device resident (what ba_pipeline.save_corrected_rpcs loops over,
    ref:bundle_adjust/ba_pipeline.py:406-423): mesh generation, localisation through the original RPCs, corrected projection, fit,
    reprojection errors and the coverage test all run on the device for all cameras at once (satba_rpc_refit, csrc/satba_rpcfit.h);
    the host only doubles the margins of the cameras whose mesh does not cover their crop yet.  Returns a list of
    (rpc_calib, err, margin) like fit_Rt_corrected_rpc; with return_info also a dict with the last mesh of every camera
    (input_locs (M, n^3, 3), target (M, n^3, 2)).
    """
    lib = E.load_library()
    M, n3 = len(original_rpcs), int(n_samples) ** 3
    if M == 0:
        return ([], {}) if return_info else []
    tabs = np.ascontiguousarray(np.stack([np.asarray(r.to_table(), dtype=np.float64) for r in original_rpcs]))
    rt = np.ascontiguousarray(np.stack([np.asarray(v, dtype=np.float64).reshape(9) for v in Rt_vecs]))
    crops = np.ascontiguousarray(np.array([[c["col0"], c["row0"], c["width"], c["height"]] for c in crop_offsets], dtype=np.float64))
    alts = np.ascontiguousarray(np.array([[r.alt_offset - r.alt_scale, r.alt_offset + r.alt_scale] for r in original_rpcs], dtype=np.float64))
    gt = None if global_transform is None else np.ascontiguousarray(np.asarray(global_transform, dtype=np.float64).reshape(3))
    tables = np.zeros((M, 90)); err = np.zeros((M, n3)); margins = np.zeros(M)
    locs = np.zeros((M, n3, 3)) if return_info else None
    target = np.zeros((M, n3, 2)) if return_info else None
    rc = lib.satba_rpc_refit(M, E._ptr(tabs), E._ptr(rt), E._ptr(crops), E._ptr(alts), E._ptr(gt) if gt is not None else None, int(n_samples),
                             1e-3, 1e-2, 20, E._ptr(tables), E._ptr(err), E._ptr(margins), E._ptr(locs) if return_info else None,
                             E._ptr(target) if return_info else None, int(os.environ.get("LOCAL_RANK", "0")))
    if rc == -4:
        raise np.linalg.LinAlgError("Singular matrix")  # the reference's numpy.linalg.inv raises the same on degenerate samples
    E._check(lib, rc)
    out = [(RPCModel.from_table(tables[k]), err[k], int(margins[k])) for k in range(M)]
    return (out, {"input_locs": locs, "target": target}) if return_info else out
