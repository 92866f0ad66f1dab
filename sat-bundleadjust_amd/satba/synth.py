"""
Seeded synthetic bundle-adjustment scenes (the protocol of SURVEY.md section 8d).

Used by the parity tests, by tools/gen_golden.py (which feeds the same inputs to the reference) and by
bench.py.  Scenes are produced in sparse form (point-major observation lists); `to_dense_C` builds the
reference's (2M, N) correspondence matrix for the small cases that go through the reference itself.

Named shapes (affine):  C2 = (10, 5000, 6, seed 1) -> 29 878 observations
                        C3 = (50, 100 000, 10, seed 1) -> ~1 M
                        C4 = (200, 1 000 000, 10, seed 1) -> ~10 M
"""
import os

import numpy as np

from . import ba_rotate, cam_utils, geo_utils
from .rpc_model import RPCModel

SCENE_CENTRE = np.array([1.7e6, -5.9e6, 1.2e6])  # ECEF-scale magnitudes, metres

SHAPES = {"C2": (10, 5000, 6), "C3": (50, 100000, 10), "C4": (200, 1000000, 10)}
# BASELINE.json configs by name: (camera model, correction_params, n_cam, n_pts, obs_per_pt)
CONFIGS = {"C2": ("affine", ["R", "T"], 10, 5000, 6), "C3": ("affine", ["R", "T"], 50, 100000, 10),
           "C4": ("affine", ["R", "T"], 200, 1000000, 10), "C5": ("rpc", ["R"], 50, 100000, 10),
           "P3": ("perspective", ["R", "T"], 50, 100000, 10),
           # between C3 and C4 (tuning of thresholds that depend on the number of cameras; not bench lines)
           "M100": ("affine", ["R", "T"], 100, 300000, 10), "M150": ("affine", ["R", "T"], 150, 600000, 10)}


class Scene:
    """Plain container: cameras (initial), pts3d (initial), observation lists, ground truth."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    @property
    def n_obs(self):
        return self.pts_ind.size

    def to_dense_C(self):
        C = np.full((2 * self.n_cam, self.n_pts), np.nan)
        C[2 * self.cam_ind, self.pts_ind] = self.pts2d[:, 0]
        C[2 * self.cam_ind + 1, self.pts_ind] = self.pts2d[:, 1]
        return C


def _visibility(rng, n_cam, n_pts, obs_per_pt, chunk=200000):
    """Bernoulli(obs_per_pt / n_cam) mask per (point, camera), at least two observations per point."""
    p = min(1.0, obs_per_pt / n_cam)
    pts_ind, cam_ind = [], []
    # one reusable draw buffer: first-touch page faults on fresh 300 MB arrays cost far more than the RNG itself
    draws = np.empty((min(chunk, n_pts), n_cam))
    mask_buf = np.empty(draws.shape, dtype=bool)
    for start in range(0, n_pts, chunk):
        n = min(chunk, n_pts - start)
        rng.random(out=draws[:n])
        mask = np.less(draws[:n], p, out=mask_buf[:n])
        short = np.nonzero(mask.sum(axis=1) < 2)[0]
        for i in short:  # rare: force two distinct cameras
            mask[i, rng.choice(n_cam, size=2, replace=False)] = True
        pi, ci = np.nonzero(mask)
        pts_ind.append(pi + start)
        cam_ind.append(ci)
    return np.concatenate(pts_ind), np.concatenate(cam_ind)


def _project_linear(Ps, rows, cam_ind, X, chunk=1000000):
    """Ps[cam, :rows, :3] @ X + Ps[cam, :rows, 3] per observation, in chunks (bounded temporaries)."""
    out = np.empty((cam_ind.size, rows))
    for s in range(0, cam_ind.size, chunk):
        c = cam_ind[s: s + chunk]
        out[s: s + chunk] = np.einsum("kij,kj->ki", Ps[c][:, :rows, :3], X[s: s + chunk]) + Ps[c][:, :rows, 3]
    return out


def rotate_points(X, angles):
    """R(angles) X for (K, 3) points and (K, 3) or (3,) Euler angles; R = Rz Ry Rx."""
    angles = np.broadcast_to(angles, X.shape)
    ca, sa = np.cos(angles[:, 0]), np.sin(angles[:, 0])
    cb, sb = np.cos(angles[:, 1]), np.sin(angles[:, 1])
    cg, sg = np.cos(angles[:, 2]), np.sin(angles[:, 2])
    x, y, z = X[:, 0], X[:, 1], X[:, 2]
    y, z = ca * y - sa * z, sa * y + ca * z
    x, z = cb * x + sb * z, -sb * x + cb * z
    x, y = cg * x - sg * y, sg * x + cg * y
    return np.stack((x, y, z), axis=1)


def make_affine_scene(n_cam, n_pts, obs_per_pt, seed=1, sigma_theta=2e-6, noise_px=0.3, pts_noise_m=2.0,
                      pts_float32=False):
    rng = np.random.default_rng(seed)
    c = SCENE_CENTRE
    pts_true = c + rng.uniform(-5e3, 5e3, (n_pts, 3))
    angles = rng.uniform(-0.5, 0.5, (n_cam, 3))
    u = rng.uniform(-1.0, 1.0, (n_cam, 3))
    shift = rng.uniform(2000.0, 4000.0, (n_cam, 2))
    dtheta = rng.normal(0.0, sigma_theta, (n_cam, 3))
    cams_true, cams_init = [], []
    for i in range(n_cam):
        K = np.array([[1 + 0.05 * u[i, 0], 0.01 * u[i, 1]], [0.0, 1 + 0.05 * u[i, 2]]])
        R = ba_rotate.euler_angles_to_R(*angles[i])
        T = -R[:2] @ c + shift[i]
        cams_true.append(cam_utils.compose_affine_camera(K, R, T))
        cams_init.append(cam_utils.compose_affine_camera(K, ba_rotate.euler_angles_to_R(*(angles[i] + dtheta[i])), T))
    pts_ind, cam_ind = _visibility(rng, n_cam, n_pts, obs_per_pt)
    proj = _project_linear(np.stack(cams_true), 2, cam_ind, pts_true[pts_ind])
    pts2d = proj + rng.normal(0.0, noise_px, proj.shape)
    pts_init = pts_true + rng.normal(0.0, pts_noise_m, pts_true.shape)
    if pts_float32:
        pts_init = pts_init.astype(np.float32)
    return Scene(cam_model="affine", n_cam=n_cam, n_pts=n_pts, cameras=cams_init, cameras_true=cams_true,
                 pts3d=pts_init, pts3d_true=pts_true, pts_ind=pts_ind, cam_ind=cam_ind, pts2d=pts2d,
                 camera_centers=[np.zeros(3) for _ in range(n_cam)], pairs_to_triangulate=[(0, 1)])


def make_perspective_scene(n_cam, n_pts, obs_per_pt, seed=1, sigma_theta=2e-6, noise_px=0.3, pts_noise_m=2.0):
    """Pinhole cameras ~600 km above the scene looking at its centre, ~1 px per metre."""
    rng = np.random.default_rng(seed)
    c = SCENE_CENTRE
    up = c / np.linalg.norm(c)
    pts_true = c + rng.uniform(-5e3, 5e3, (n_pts, 3))
    cams_true, cams_init, centers = [], [], []
    for i in range(n_cam):
        d = up + 0.3 * rng.uniform(-1, 1, 3)
        oC = c + 6.0e5 * d / np.linalg.norm(d)
        fwd = (c - oC) / np.linalg.norm(c - oC)
        right = np.cross(fwd, rng.uniform(-1, 1, 3))
        right /= np.linalg.norm(right)
        R = np.vstack((right, np.cross(fwd, right), fwd))
        f = 6.0e5 * (1 + 0.05 * rng.uniform(-1, 1))
        K = np.array([[f, 10.0 * rng.uniform(-1, 1), 2000 + 100 * rng.uniform(-1, 1)],
                      [0.0, f * (1 + 0.01 * rng.uniform(-1, 1)), 2000 + 100 * rng.uniform(-1, 1)], [0, 0, 1.0]])
        a = np.array(ba_rotate.euler_angles_from_R(R))
        vecT = -(R @ oC)
        Rn = ba_rotate.euler_angles_to_R(*(a + rng.normal(0, sigma_theta, 3)))
        for Ri, out in ((R, cams_true), (Rn, cams_init)):
            P = K @ np.hstack((Ri, vecT.reshape(3, 1)))
            out.append(P / P[2, 3])
        centers.append(oC)
    pts_ind, cam_ind = _visibility(rng, n_cam, n_pts, obs_per_pt)
    h = _project_linear(np.stack(cams_true), 3, cam_ind, pts_true[pts_ind])
    pts2d = h[:, :2] / h[:, 2:3] + rng.normal(0.0, noise_px, (h.shape[0], 2))
    pts_init = pts_true + rng.normal(0.0, pts_noise_m, pts_true.shape)
    return Scene(cam_model="perspective", n_cam=n_cam, n_pts=n_pts, cameras=cams_init, cameras_true=cams_true,
                 pts3d=pts_init, pts3d_true=pts_true, pts_ind=pts_ind, cam_ind=cam_ind, pts2d=pts2d,
                 camera_centers=centers, pairs_to_triangulate=[(0, 1)])


def default_rpc_files():
    """The two SkySat RPCs shipped with the reference's tests, kept as data fixtures under tests/golden/rpc."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "rpc")
    return sorted(os.path.join(root, f) for f in os.listdir(root) if f.endswith(".rpc"))


def make_rpc_scene(n_cam, n_pts, obs_per_pt, seed=1, sigma_theta=1e-6, noise_px=0.2, pts_noise_m=2.0,
                   rpc_files=None):
    """
    RPC cameras: file i mod 2 of the shipped pair with row/col offsets jittered by +-50 px; ground points around
    the RPC footprint; the observations come from cameras carrying a small true corrective rotation
    about a centre ~500 km above the scene, and the solve starts from the identity correction.
    """
    rng = np.random.default_rng(seed)
    base = [RPCModel.from_file(f) for f in (rpc_files or default_rpc_files())]
    r0 = base[0]
    lon = r0.lon_offset + rng.uniform(-0.01, 0.01, n_pts)
    lat = r0.lat_offset + rng.uniform(-0.005, 0.005, n_pts)
    alt = r0.alt_offset + rng.uniform(-100.0, 100.0, n_pts)
    pts_true = np.stack(geo_utils.latlon_to_ecef_custom(lat, lon, alt), axis=1)
    mean = pts_true.mean(axis=0)
    up = mean / np.linalg.norm(mean)
    rpcs, centers = [], []
    for i in range(n_cam):
        r = base[i % len(base)].copy()
        r.row_offset += rng.uniform(-50, 50)
        r.col_offset += rng.uniform(-50, 50)
        rpcs.append(r)
        centers.append(mean + 5.0e5 * up + rng.uniform(-5e4, 5e4, 3))
    true_angles = rng.normal(0.0, sigma_theta, (n_cam, 3))
    pts_ind, cam_ind = _visibility(rng, n_cam, n_pts, obs_per_pt)
    Cs = np.stack(centers)[cam_ind]
    X = rotate_points(pts_true[pts_ind] - Cs, true_angles[cam_ind]) + Cs
    la, lo, al = geo_utils.ecef_to_latlon_custom(X[:, 0], X[:, 1], X[:, 2])
    pts2d = np.zeros((pts_ind.size, 2))
    for i in range(n_cam):
        sel = cam_ind == i
        col, row = rpcs[i].projection(lo[sel], la[sel], al[sel])
        pts2d[sel, 0], pts2d[sel, 1] = col, row
    pts2d += rng.normal(0.0, noise_px, pts2d.shape)
    pts_init = pts_true + rng.normal(0.0, pts_noise_m, pts_true.shape)
    return Scene(cam_model="rpc", n_cam=n_cam, n_pts=n_pts, cameras=rpcs, cameras_true=rpcs, true_angles=true_angles,
                 pts3d=pts_init, pts3d_true=pts_true, pts_ind=pts_ind, cam_ind=cam_ind, pts2d=pts2d,
                 camera_centers=centers, pairs_to_triangulate=[(0, 1)])


def make_scene(cam_model, n_cam, n_pts, obs_per_pt, seed=1, **kw):
    return {"affine": make_affine_scene, "perspective": make_perspective_scene, "rpc": make_rpc_scene}[cam_model](
        n_cam, n_pts, obs_per_pt, seed=seed, **kw)


def make_params(scene, d=None, dense=False):
    """BundleAdjustmentParameters of a scene, through the dense C (reference route) or the sparse entry point."""
    from .ba_params import BundleAdjustmentParameters

    d = dict({"verbose": False}, **(d or {}))
    if dense:
        return BundleAdjustmentParameters(scene.to_dense_C(), scene.pts3d, scene.cameras, scene.cam_model,
                                          scene.pairs_to_triangulate, scene.camera_centers, d)
    return BundleAdjustmentParameters.from_observations(
        scene.pts_ind, scene.cam_ind, scene.pts2d, scene.pts3d, scene.cameras, scene.cam_model,
        scene.pairs_to_triangulate, scene.camera_centers, d)
