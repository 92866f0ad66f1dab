"""
Variable packing for the bundle-adjustment least-squares problem.

Host-side mirror of ref:bundle_adjust/ba_params.py: same class name, constructor signature,
attributes, methods and module functions, so code written against the reference
(`ba_pipeline`, `ba_outliers`, `ft_ranking`, ...) runs on it unchanged.  What differs is how the
state is built: the observation lists are produced by one vectorised pass over the NaN mask of C
(the reference walks points x cameras in Python, ref:bundle_adjust/ba_params.py:139-148: 11 s at
1 M observations) and a sparse entry point (`from_observations`) avoids the dense 2M x N matrix
altogether at the 10 M-observation scale.

Variable vector layout (ref:bundle_adjust/ba_params.py:152-172):
    params_opt = [ cam 0 (n_params) | ... | cam M-1 (n_params) | pt 0 (3) | ... | pt N-1 (3) ]
with n_params = 3 for ["R"], 5 (affine) or 6 (perspective, rpc) for ["R", "T"].
Per-camera parameter rows `cam_params` (ref:bundle_adjust/ba_params.py:19-44):
    affine       [a, b, g, t0, t1, fx, fy, skew]                     (8)
    perspective  [a, b, g, t0, t1, t2, fx, fy, skew, cx, cy]         (11)
    rpc          [a, b, g, T0, T1, T2, C0, C1, C2]  (angles, T start at 0; C = camera centre)  (9)
"""
import numpy as np

from . import ba_rotate, cam_utils


class Error(Exception):
    pass


_N_T = {"affine": 2, "perspective": 3, "rpc": 3}


def load_cam_params_from_camera(camera, camera_center, cam_model):
    """Camera model -> row of `cam_params` (ref:bundle_adjust/ba_params.py:19-44)."""
    if cam_model == "affine":
        K, R, vecT = cam_utils.decompose_affine_camera(camera)
        intr = [K[0, 0], K[1, 1], K[0, 1]]
    elif cam_model == "perspective":
        K, R, vecT, _ = cam_utils.decompose_perspective_camera(camera)
        K = K / K[2, 2]
        intr = [K[0, 0], K[1, 1], K[0, 1], K[0, 2], K[1, 2]]
    else:
        # rpc: the correction starts at identity; only the rotation centre is data
        return np.hstack([np.zeros(6, dtype=np.float32), camera_center])
    angles = np.array(ba_rotate.euler_angles_from_R(R))
    return np.hstack((angles.ravel(), np.asarray(vecT).ravel(), *intr))


def load_camera_from_cam_params(cam_params, cam_model):
    """Row of `cam_params` -> camera model (ref:bundle_adjust/ba_params.py:47-75); rpc rows pass through as (1, 9)."""
    if cam_model == "rpc":
        return cam_params.reshape((1, 9))
    R = ba_rotate.euler_angles_to_R(*cam_params[0:3].tolist())
    if cam_model == "affine":
        fx, fy, skew = cam_params[5:8]
        P = cam_utils.compose_affine_camera(np.array([[fx, skew], [0, fy]]), R, cam_params[3:5])
    else:
        fx, fy, skew, cx, cy = cam_params[6:11]
        K = np.array([[fx, skew, cx], [0, fy, cy], [0, 0, 1]])
        P = K @ np.hstack((R, cam_params[3:6].reshape((3, 1))))
    return P / P[2, 3]


def observations_from_C(C):
    """
    (pts_ind, cam_ind, pts2d) of a correspondence matrix, in the reference's order: for each point,
    cameras ascending (ref:bundle_adjust/ba_params.py:142-147).
    """
    seen = ~np.isnan(C[::2, :])  # (M, N)
    pts_ind, cam_ind = np.nonzero(seen.T)  # row-major over (point, camera) == point-major order
    pts2d = np.stack((C[2 * cam_ind, pts_ind], C[2 * cam_ind + 1, pts_ind]), axis=1)
    return pts_ind, cam_ind, pts2d


class BundleAdjustmentParameters:
    def __init__(self, C, pts3d, cameras, cam_model, pairs_to_triangulate, camera_centers, d):
        """
        Same contract as ref:bundle_adjust/ba_params.py:79-181.

        Args:
            C: (2M, N) correspondence matrix, NaN where a track is not observed
            pts3d: (N, 3) initial ECEF coordinates of the tracks (float32 or float64)
            cameras: list of M 3x4 matrices (affine / perspective) or RPC models
            cam_model: "affine" | "perspective" | "rpc"
            pairs_to_triangulate: list of camera index pairs
            camera_centers: list of M 3-vectors
            d: options -- "n_cam_fix", "n_pts_fix", "reduce" (True), "verbose" (True),
               "correction_params" (["R"]), "ref_cam_weight" (1.0)
        """
        self._init_options(cam_model, d)
        self.C = C.copy()
        self.pts3d = pts3d.copy()
        self.cameras = cameras.copy()
        self.pairs_to_triangulate = pairs_to_triangulate.copy()
        self.camera_centers = camera_centers.copy()

        verbose = d.get("verbose", True)
        if verbose:
            print("\nDefining bundle adjustment parameters...")
            print("     - cam_params_to_optimize: {}\n".format(self.cam_params_to_optimize))

        self.n_cam, self.n_pts = C.shape[0] // 2, C.shape[1]
        self.n_cam_opt = self.n_cam - self.n_cam_fix
        self.n_pts_opt = self.n_pts - self.n_pts_fix
        self.cam_prev_indices = np.arange(self.n_cam)
        self.pts_prev_indices = np.arange(self.n_pts)
        if d.get("reduce", True):
            self.reduce(C, pts3d, cameras, pairs_to_triangulate, camera_centers)
            if verbose:
                print("C.shape before reduce", C.shape)
                print("C.shape after reduce", self.C.shape)

        self.pts_ind, self.cam_ind, self.pts2d = observations_from_C(self.C)
        self._finish(verbose)

    @classmethod
    def from_observations(cls, pts_ind, cam_ind, pts2d, pts3d, cameras, cam_model, pairs_to_triangulate,
                          camera_centers, d):
        """
        Sparse entry point (not in the reference): build the same object from observation lists
        instead of the dense C (which would be 3.2 GB at 200 cams x 1 M points).  Observations must be
        point-major with cameras ascending inside a point, i.e. the order the reference derives
        from C.  `self.C` is left as None; `reduce` is not applied.
        """
        self = cls.__new__(cls)
        self._init_options(cam_model, d)
        self.C = None
        self.pts3d = pts3d.copy()
        self.cameras = list(cameras)
        self.pairs_to_triangulate = list(pairs_to_triangulate)
        self.camera_centers = list(camera_centers)
        self.n_cam, self.n_pts = len(cameras), pts3d.shape[0]
        self.n_cam_opt = self.n_cam - self.n_cam_fix
        self.n_pts_opt = self.n_pts - self.n_pts_fix
        self.cam_prev_indices = np.arange(self.n_cam)
        self.pts_prev_indices = np.arange(self.n_pts)
        pts_ind = np.asarray(pts_ind, dtype=np.int64)
        cam_ind = np.asarray(cam_ind, dtype=np.int64)
        key = pts_ind * self.n_cam + cam_ind
        if key.size and np.any(np.diff(key) <= 0):
            raise Error("observations must be sorted by (point, camera) without duplicates")
        self.pts_ind, self.cam_ind = pts_ind, cam_ind
        self.pts2d = np.array(pts2d, dtype=np.float64).reshape(-1, 2)
        self._finish(d.get("verbose", True))
        return self

    def _init_options(self, cam_model, d):
        if cam_model not in _N_T:
            raise Error("unknown cam_model {!r}".format(cam_model))
        self.cam_model = cam_model
        self.cam_params_to_optimize = d.get("correction_params", ["R"])
        self.ref_cam_weight = d.get("ref_cam_weight", 1.0)
        self.n_cam_fix = d.get("n_cam_fix", 0)
        self.n_pts_fix = d.get("n_pts_fix", 0)
        if "K" in self.cam_params_to_optimize:
            # ref:bundle_adjust/ba_params.py:163 slices the T columns a second time instead of K
            # (SURVEY.md section 0, fact 3): there is no well-defined behaviour to reproduce.
            raise Error('correction_params containing "K" / "COMMON_K" are not supported')

    def _finish(self, verbose):
        self.cam_params = np.array(
            [load_cam_params_from_camera(c, oC, self.cam_model) for c, oC in zip(self.cameras, self.camera_centers)]
        )
        self.n_obs = self.pts2d.shape[0]

        # variables to optimise: T is honoured only together with R (ref:bundle_adjust/ba_params.py:153-163)
        self.n_params = 0
        if "R" in self.cam_params_to_optimize:
            self.n_params = 3
            if "T" in self.cam_params_to_optimize:
                self.n_params += _N_T[self.cam_model]
        else:
            # the reference builds `cam_params_opt = []` here and fails on `cam_params_opt.ravel()` (AttributeError,
            # ref:bundle_adjust/ba_params.py:152-170): there is no behaviour to reproduce, only a clearer message
            raise Error('correction_params must contain "R" (the reference raises AttributeError without it)')
        cam_params_opt = self.cam_params[:, : self.n_params]
        self.params_opt = np.hstack((cam_params_opt.ravel(), self.pts3d.ravel()))

        self.pts2d_w = np.ones(self.n_obs)
        if self.ref_cam_weight > 1.0:
            self.pts2d_w[self.cam_ind == 0] = self.ref_cam_weight

        if verbose:
            print("{} 3d points, {} fixed and {} to be optimized".format(self.n_pts, self.n_pts_fix, self.n_pts_opt))
            print("{} cameras, {} fixed and {} to be optimized".format(self.n_cam, self.n_cam_fix, self.n_cam_opt))
            print("{} parameters to optimize per camera\n".format(self.n_params))

    def reduce(self, C, pts3d, cameras, pairs_to_triangulate, camera_centers):
        """
        Keep only the tracks seen by at least one camera to optimise, then drop cameras left without
        observations; remap counters, cameras and pairs (ref:bundle_adjust/ba_params.py:183-219).
        The slices below use the reference's own `[-n:]` expressions on purpose: with n == 0 they
        select everything, and callers may rely on that.
        """
        seen = ~np.isnan(C[::2, :])
        keep_pts = seen[-self.n_cam_opt :].sum(axis=0).astype(bool)
        self.C = C[:, keep_pts].copy()
        self.pts_prev_indices = np.arange(self.n_pts, dtype=int)[keep_pts]
        self.n_pts_fix -= np.sum(~keep_pts[: self.n_pts_fix])
        self.n_pts_opt -= np.sum(~keep_pts[-self.n_pts_opt :])
        self.pts3d = pts3d[self.pts_prev_indices, :].copy()

        keep_cams = np.sum(~np.isnan(self.C[::2]), axis=1) > 0
        self.cam_prev_indices = np.arange(self.n_cam, dtype=int)[keep_cams]
        self.C = self.C[np.repeat(keep_cams, 2), :]
        self.n_cam, self.n_pts = self.C.shape[0] // 2, self.C.shape[1]
        self.n_cam_fix -= np.sum(~keep_cams[: self.n_cam_fix])
        self.n_cam_opt -= np.sum(~keep_cams[-self.n_cam_opt :])
        self.cameras = [cameras[i] for i in self.cam_prev_indices]
        self.camera_centers = [camera_centers[i] for i in self.cam_prev_indices]

        new_index = np.full(len(keep_cams), -1)
        new_index[keep_cams] = np.arange(np.sum(keep_cams))
        self.pairs_to_triangulate = [
            (new_index[a], new_index[b]) for [a, b] in pairs_to_triangulate if keep_cams[a] and keep_cams[b]
        ]

    def get_vars_ready_for_fun(self, v):
        """
        v -> (pts3d (N, 3), cam_params (M, c_p)) with frozen rows restored from the initial state
        (ref:bundle_adjust/ba_params.py:221-257).  As in the reference the frozen camera rows are written
        through a view, i.e. into the caller's v.
        """
        n_c = self.n_cam * self.n_params
        pts3d = v[n_c:].reshape((self.n_pts, 3)).copy()
        if self.n_pts_fix > 0:
            pts3d[: self.n_pts_fix, :] = self.pts3d[: self.n_pts_fix, :]
        cam_params_opt = v[:n_c].reshape((self.n_cam, self.n_params))
        if self.n_cam_fix > 0:
            cam_params_opt[: self.n_cam_fix, :] = self.cam_params[: self.n_cam_fix, : self.n_params]
        cam_params = np.hstack((cam_params_opt, self.cam_params[:, self.n_params :]))
        return pts3d, cam_params

    def reconstruct_vars(self, v, pts3d, cameras):
        """
        Unpack a solution into corrected points / cameras and scatter them back to the caller's full-size
        containers (ref:bundle_adjust/ba_params.py:259-286).  Sets pts3d_ba, cameras_ba, estimated_params.
        """
        self.pts3d_ba, cam_params = self.get_vars_ready_for_fun(v)
        self.cameras_ba = [load_camera_from_cam_params(cam_params[i, :], self.cam_model) for i in range(self.n_cam)]

        self.estimated_params = []
        for row in cam_params:
            est = {}
            if "R" in self.cam_params_to_optimize:
                est["R"] = row[:3]
            if "T" in self.cam_params_to_optimize:
                est["T"] = row[3:6]
            if self.cam_model == "rpc":
                est["C"] = row[6:9]
            self.estimated_params.append(est)

        print("\n")
        corrected_pts3d, corrected_cameras = pts3d.copy(), cameras.copy()
        corrected_pts3d[self.pts_prev_indices] = self.pts3d_ba
        for ba_idx, prev_idx in enumerate(self.cam_prev_indices):
            corrected_cameras[prev_idx] = self.cameras_ba[ba_idx]
        return corrected_pts3d, corrected_cameras
