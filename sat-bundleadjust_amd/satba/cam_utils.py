"""
Camera (de)composition helpers used by ba_params to pack / unpack camera matrices
(ref:bundle_adjust/cam_utils.py:45-75, 78-89, 92-143, 201-231).  Host numpy only.
"""
import numpy as np

from . import geo_utils


def decompose_perspective_camera(P):
    """
    P = K R [I | -oC] with K upper-triangular, positive diagonal (Hartley & Zisserman 6.2.4;
    ref:bundle_adjust/cam_utils.py:45-75).  Returns K, R, vecT = -R oC, oC.
    """
    from scipy.linalg import rq

    M = P[:, :3]
    K, R = rq(M)
    sgn = np.sign(np.diag(K))
    K = K * sgn[np.newaxis, :]  # K diag(sgn)
    R = sgn[:, np.newaxis] * R  # diag(sgn) R
    oC = -np.linalg.solve(M, P[:, 3])
    vecT = -(R @ oC)
    return K, R, vecT, oC


def compose_perspective_camera(K, R, oC):
    """P = K R [I | -oC]  (ref:bundle_adjust/cam_utils.py:78-89)."""
    return K @ R @ np.hstack((np.eye(3), -np.asarray(oC, dtype=np.float64).reshape(3, 1)))


def decompose_affine_camera(P):
    """
    Affine camera P = [[K, 0], [0, 1]] @ [[R[:2], vecT], [0, 1]], K = [[fx, s], [0, fy]]
    (Hartley & Zisserman 6.3.3; ref:bundle_adjust/cam_utils.py:92-126).  Returns K (2x2), R (3x3), vecT (2x1).
    """
    M = P[:2, :3]
    G = M @ M.T
    fy = np.sqrt(G[1, 1])
    s = G[1, 0] / fy
    fx = np.sqrt(G[0, 0] - s * s)
    K = np.array([[fx, s], [0.0, fy]])
    Kinv = np.linalg.inv(K)
    R2 = Kinv @ M
    R = np.vstack((R2, np.cross(R2[0], R2[1])))
    vecT = Kinv @ P[:2, 3:4]
    return K, R, vecT


def compose_affine_camera(K, R, vecT):
    """Inverse of decompose_affine_camera (ref:bundle_adjust/cam_utils.py:129-143)."""
    P = np.zeros((3, 4))
    P[:2, :3] = K @ R[:2]
    P[:2, 3] = K @ np.asarray(vecT, dtype=np.float64).reshape(2)
    P[2, 3] = 1.0
    return P


def apply_projection_matrix(P, pts3d):
    """Project Nx3 points with a 3x4 matrix (ref:bundle_adjust/cam_utils.py:201-214)."""
    h = pts3d @ P[:, :3].T + P[:, 3]
    return h[:, :2] / h[:, 2:3]


def apply_rpc_projection(rpc, pts3d):
    """ECEF -> geodetic -> rpc.projection (ref:bundle_adjust/cam_utils.py:217-231)."""
    lat, lon, alt = geo_utils.ecef_to_latlon_custom(pts3d[:, 0], pts3d[:, 1], pts3d[:, 2])
    col, row = rpc.projection(lon, lat, alt)
    return np.vstack((col, row)).T


def generate_point_mesh(col_range, row_range, alt_range):
    """(col, row, alt) coordinates of the n_col x n_row x n_alt grid given by three (min, max, n) triplets, flattened with the
    columns running fastest and the altitudes slowest (ref:bundle_adjust/cam_utils.py:280-306)."""
    cols, rows, alts = [np.linspace(v[0], v[1], int(v[2])) for v in (col_range, row_range, alt_range)]
    a, r, c = np.meshgrid(alts, rows, cols, indexing="ij")
    return c.reshape(-1), r.reshape(-1), a.reshape(-1)
