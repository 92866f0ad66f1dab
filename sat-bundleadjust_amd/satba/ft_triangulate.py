"""
Initial 3-D points of the feature tracks by triangulation, on the device (the names of
ref:bundle_adjust/feature_tracks/ft_triangulate.py).

`init_pts3d`, `linear_triangulation_multiple_pts` and `rpc_triangulation` keep the reference's signatures and return values.
The arithmetic is in csrc/satba_triangulate.h behind `satba_init_pts3d` / `satba_triangulate_pairwise` (include/satba.h); there
is no CPU fallback: without libsatba_hip.so or without a GPU the calls raise.

`init_pts3d_from_observations` is the same computation for callers that hold observation lists (BundleAdjustmentParameters:
pts_ind / cam_ind / pts2d) instead of the dense NaN-sparse correspondence matrix.
"""
import ctypes as C

import numpy as np

from . import engine_hip as E


def _camera_table(cameras, cam_model):
    """(M, 12) projection matrices or (M, 90) RPC records (include/satba.h: SATBA_RPC_TABLE_LEN)."""
    if cam_model == "rpc":
        return np.ascontiguousarray(np.stack([np.asarray(c.to_table() if hasattr(c, "to_table") else _rpcm_table(c), dtype=np.float64)
                                              for c in cameras]))
    return np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.float64).reshape(12) for c in cameras]))


def _rpcm_table(rpc):
    """record of an rpcm.RPCModel-like object (attribute names of ref:bundle_adjust/s2p/triangulation.py:43-61)"""
    return np.concatenate([rpc.col_num, rpc.col_den, rpc.row_num, rpc.row_den,
                           [rpc.lon_offset, rpc.lon_scale, rpc.lat_offset, rpc.lat_scale, rpc.alt_offset, rpc.alt_scale,
                            rpc.col_offset, rpc.col_scale, rpc.row_offset, rpc.row_scale]])


def _device(device):
    if device is not None:
        return int(device)
    import os
    return int(os.environ.get("LOCAL_RANK", "0"))


def _pairwise(cam_model, cam_i, cam_j, pts1, pts2, want_err, device=None):
    lib = E.load_library()
    pts1 = np.ascontiguousarray(pts1, dtype=np.float64).reshape(-1, 2)
    pts2 = np.ascontiguousarray(pts2, dtype=np.float64).reshape(-1, 2)
    if pts1.shape != pts2.shape:
        raise ValueError("pts1 and pts2 must have the same shape")
    n = pts1.shape[0]
    tab = _camera_table([cam_i, cam_j], cam_model)
    out = np.zeros((n, 3), dtype=np.float64)
    err = np.zeros(n, dtype=np.float32) if want_err else None
    ms = C.c_float(0.0)
    E._check(lib, lib.satba_triangulate_pairwise(E.CAM_MODELS[cam_model], E._ptr(tab[0:1]), E._ptr(tab[1:2]), n, E._ptr(pts1), E._ptr(pts2),
                                                 E._ptr(out), err.ctypes.data_as(C.POINTER(C.c_float)) if want_err else None,
                                                 _device(device), C.byref(ms)))
    return out, err, ms.value


def linear_triangulation_multiple_pts(P1, P2, pts1, pts2):
    """
    ref:bundle_adjust/feature_tracks/ft_triangulate.py:18-34: linear triangulation of N correspondences between two 3 x 4
    projection matrices; pts1, pts2 (N, 2) image coordinates (col, row).  Returns pts3d (N, 3) float64.
    """
    return _pairwise("perspective", P1, P2, pts1, pts2, False)[0]


def rpc_triangulation(rpc1, rpc2, pts1, pts2):
    """
    ref:bundle_adjust/feature_tracks/ft_triangulate.py:37-54: triangulation of N correspondences between two RPC models.
    Returns (pts3d (N, 3) float64 ECEF, err (N, 1) float32 as the reference's stereo_corresp_to_xyz returns it).
    """
    out, err, _ = _pairwise("rpc", rpc1, rpc2, pts1, pts2, True)
    return out, err.reshape(-1, 1)


def init_pts3d_from_observations(pts_ind, cam_ind, pts2d, n_pts, cameras, cam_model, pairs_to_triangulate, device=None, reps=1,
                                 return_info=False):
    """
    init_pts3d for observation lists: pts_ind (K,), cam_ind (K,), pts2d (K, 2); any order (grouped by track here).
    Returns avg_pts3d (n_pts, 3) float32; with return_info also a dict with the kernel time and the triangulations per track.
    """
    lib = E.load_library()
    pts_ind = np.asarray(pts_ind, dtype=np.int64); cam_ind = np.asarray(cam_ind, dtype=np.int64)
    pts2d = np.asarray(pts2d, dtype=np.float64).reshape(-1, 2)
    n_pts = int(n_pts)
    if pts_ind.size and (pts_ind.min() < 0 or pts_ind.max() >= n_pts):
        raise ValueError("pts_ind out of range")
    if np.any(pts_ind[1:] < pts_ind[:-1]):
        order = np.argsort(pts_ind, kind="stable")
        pts_ind, cam_ind, pts2d = pts_ind[order], cam_ind[order], pts2d[order]
    ofs = np.zeros(n_pts + 1, dtype=np.int64)
    np.cumsum(np.bincount(pts_ind, minlength=n_pts), out=ofs[1:])
    cam32 = np.ascontiguousarray(cam_ind, dtype=np.int32)
    obs = np.ascontiguousarray(pts2d)
    tab = _camera_table(cameras, cam_model)
    pairs = np.ascontiguousarray(np.asarray(list(pairs_to_triangulate), dtype=np.int32).reshape(-1, 2))
    out = np.zeros((n_pts, 3), dtype=np.float32)
    n_tri = np.zeros(n_pts, dtype=np.int32)
    ms = C.c_float(0.0)
    fp = C.POINTER(C.c_float)
    E._check(lib, lib.satba_init_pts3d(E.CAM_MODELS[cam_model], tab.shape[0], n_pts, ofs.ctypes.data_as(C.POINTER(C.c_int64)),
                                       E._ptr(cam32, E._ip), E._ptr(obs), E._ptr(tab), pairs.shape[0], E._ptr(pairs, E._ip),
                                       out.ctypes.data_as(fp), E._ptr(n_tri, E._ip), _device(device), int(reps), C.byref(ms)))
    if return_info:
        return out, {"kernel_ms": ms.value, "n_tri": n_tri}
    return out


def init_pts3d_resident(p, pairs_to_triangulate=None, cameras=None, remove=None, return_info=False):
    """
    init_pts3d for the tracks of a BundleAdjustmentParameters object whose observations are already on the device (a single-rank
    engine exists for it, ba_core.cached_engine; one is built when there is none -- callers that should not pay for that check
    first, as ba_outliers.rm_outliers does: in a sharded run only the (rank, world) engines exist and a whole-problem upload per rank
    would be the opposite of "resident") -- what ref:bundle_adjust/ba_outliers.py:89-93 needs right after the outlier rejection.
    remove: (K,) bool, observations to treat as absent (compute_obs_mask's mask).  cameras / pairs default to p's.
    Returns avg_pts3d (p.n_pts, 3) float32 (zero rows where no listed pair applies); with return_info also {"kernel_ms", "n_tri"}.
    """
    from . import ba_core

    eng = ba_core.cached_engine(p) or ba_core.get_engine(p)
    cams = p.cameras if cameras is None else cameras
    pairs = p.pairs_to_triangulate if pairs_to_triangulate is None else pairs_to_triangulate
    tab = _camera_table(list(cams)[: p.n_cam], p.cam_model)
    out, n_tri, ms = eng.init_pts3d(tab, np.asarray(list(pairs), dtype=np.int32).reshape(-1, 2), remove)
    if return_info:
        return out, {"kernel_ms": ms, "n_tri": n_tri}
    return out


def init_pts3d(C_mat, cameras, cam_model, pairs_to_triangulate, verbose=False):
    """
    ref:bundle_adjust/feature_tracks/ft_triangulate.py:57-127: the 3-D point of every feature track = float32 running mean of
    the points triangulated from every pair of pairs_to_triangulate (in list order) whose two cameras see the track.
    C_mat: (2 * n_cam, n_tracks) correspondence matrix, NaN = not observed.  Returns avg_pts3d (n_tracks, 3) float32.
    """
    C_mat = np.asarray(C_mat)
    n_pts, n_cam = C_mat.shape[1], C_mat.shape[0] // 2
    seen = ~np.isnan(C_mat[::2])                 # (n_cam, n_pts), the reference's mask (ft_triangulate.py:91)
    pts_ind, cam_ind = np.nonzero(seen.T)        # grouped by track, cameras ascending
    pts2d = np.stack([C_mat[2 * cam_ind, pts_ind], C_mat[2 * cam_ind + 1, pts_ind]], axis=1)
    if verbose:
        print("Computing {} points 3d from feature tracks...".format(n_pts), flush=True)
    out = init_pts3d_from_observations(pts_ind, cam_ind, pts2d, n_pts, list(cameras)[:n_cam], cam_model, pairs_to_triangulate)
    if verbose:
        print("done!", flush=True)
    return out
