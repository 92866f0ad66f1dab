"""
TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's initial triangulation.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(sat-bundleadjust_amd/satba/ft_triangulate.py -> libsatba_hip.so) never does.

What it restates (reference file:line):
  * init_pts3d                 ref:bundle_adjust/feature_tracks/ft_triangulate.py:57-127   (pairs in list order, float32 running mean)
  * linear triangulation       ref:...ft_triangulate.py:18-34 -> cv2.triangulatePoints
  * rpc triangulation          ref:...ft_triangulate.py:37-54 -> ref:bundle_adjust/s2p/triangulation.py:82-135 (float32 keypoints,
                               delta = 0.1) -> ref:c/disp_to_h.c:40-64 -> ref:c/rpc.c:480-514 (rpc_height),
                               ref:c/rpc.c:372-408 (iterative localisation), ref:c/rpc.c:279-298 (cubic term order)
  * geodetic -> ECEF           ref:bundle_adjust/geo_utils.py:218-233

Pinning.  The RPC branch is checked against the reference's own C, compiled from where it lies into oracle/_ref/disp_to_h.so
(oracle/Makefile; tests/test_oracle_golden.py).  The linear branch calls OpenCV (`opencv-contrib-python`, unpinned in
ref:requirements.txt:8), which is absent from this image: **parity unpinned against cv2** for that call.  Its published algorithm
(modules/calib3d triangulate: per point the 4 x 4 matrix with rows x P[2] - P[0], y P[2] - P[1] of both views, SVD, the right
singular vector of the smallest singular value, de-homogenised by the caller at ft_triangulate.py:32) is restated with the
one-sided Jacobi SVD OpenCV itself implements (_null_vector_jacobi below).  The running mean and the pair / mask logic are
pinned by fixtures generated from the imported reference function itself (tools/gen_golden.py: init_pts3d.npz), with
cv2.triangulatePoints supplied by this restatement and stereo_corresp_to_xyz bound to oracle/_ref/disp_to_h.so.
"""
import numpy as np

RPCH_MAXIT = 100          # ref:c/rpc.c:476
RPCH_LAMBDA_STOP = 0.00001  # ref:c/rpc.c:478
LOC_TOL = 1e-18           # ref:c/rpc.c:394
LOC_MAXIT = 100           # the reference loops without a bound; a non-converging input would hang it


def linear_triangulation_multiple_pts(P1, P2, pts1, pts2):
    """ref:ft_triangulate.py:18-34; cv2.triangulatePoints restated (see the module header)."""
    P1 = np.asarray(P1, dtype=np.float64); P2 = np.asarray(P2, dtype=np.float64)
    pts1 = np.asarray(pts1, dtype=np.float64); pts2 = np.asarray(pts2, dtype=np.float64)
    n = pts1.shape[0]
    A = np.empty((n, 4, 4))
    A[:, 0] = pts1[:, 0:1] * P1[2] - P1[0]
    A[:, 1] = pts1[:, 1:2] * P1[2] - P1[1]
    A[:, 2] = pts2[:, 0:1] * P2[2] - P2[0]
    A[:, 3] = pts2[:, 1:2] * P2[2] - P2[1]
    X = _null_vector_jacobi(A)
    return X[:, :3] / X[:, 3:4]


def _null_vector_jacobi(A):
    """Right singular vector of the smallest singular value of every 4 x 4 matrix of A (n, 4, 4) by one-sided Jacobi rotations of
    the columns -- the method of OpenCV's built-in SVD (modules/core/src/lapack.cpp, JacobiSVDImpl_), which is what
    cv2.triangulatePoints runs on its fixed-size matrices.  Unlike a bidiagonalising SVD (numpy.linalg.svd is 4 - 9 mm away from
    the exact null vector at ECEF magnitudes, where the homogeneous column is ~1e6 times the others: tools/tri_accuracy.py) it is
    accurate to the last digits, so that two implementations of it agree to ~1e-9 m."""
    A = A.copy()
    n = A.shape[0]
    V = np.broadcast_to(np.eye(4), (n, 4, 4)).copy()
    for _ in range(30):
        rotated = False
        for p in range(3):
            for q in range(p + 1, 4):
                al = np.einsum("ni,ni->n", A[:, :, p], A[:, :, p]); be = np.einsum("ni,ni->n", A[:, :, q], A[:, :, q])
                ga = np.einsum("ni,ni->n", A[:, :, p], A[:, :, q])
                act = (np.abs(ga) > 1e-16 * np.sqrt(al * be)) & (ga != 0.0)
                if not act.any():
                    continue
                rotated = True
                g = np.where(act, ga, 1.0)
                zeta = (be - al) / (2.0 * g)
                t = np.copysign(1.0, zeta) / (np.abs(zeta) + np.sqrt(1.0 + zeta * zeta))
                c = np.where(act, 1.0 / np.sqrt(1.0 + t * t), 1.0)
                s = np.where(act, c * t, 0.0)
                for M in (A, V):
                    mp, mq = M[:, :, p].copy(), M[:, :, q].copy()
                    M[:, :, p] = c[:, None] * mp - s[:, None] * mq
                    M[:, :, q] = s[:, None] * mp + c[:, None] * mq
        if not rotated:
            break
    k = np.argmin(np.einsum("nij,nij->nj", A, A), axis=1)
    return V[np.arange(n), :, k]


def _pol20(c, x, y, z):
    """ref:c/rpc.c:279-298: note the x <-> y exchange ('inversion here')."""
    col, lig, alt = y, x, z
    m = [1.0, lig, col, alt, lig * col, lig * alt, col * alt, lig * lig, col * col, alt * alt, col * lig * alt, lig * lig * lig,
         lig * col * col, lig * alt * alt, lig * lig * col, col * col * col, col * alt * alt, lig * lig * alt, col * col * alt,
         alt * alt * alt]
    r = 0.0
    for i in range(20):
        r = r + c[i] * m[i]
    return r


class _Rpc:
    """struct rpc as ref:bundle_adjust/s2p/triangulation.py:43-61 fills it from an rpcm-style model (no direct model: numx = nan)."""
    def __init__(self, m, delta=0.1):
        self.offset = (m.col_offset, m.row_offset, m.alt_offset)
        self.scale = (m.col_scale, m.row_scale, m.alt_scale)
        self.ioffset = (m.lon_offset, m.lat_offset, m.alt_offset)
        self.iscale = (m.lon_scale, m.lat_scale, m.alt_scale)
        self.inumx, self.idenx = np.asarray(m.col_num, float), np.asarray(m.col_den, float)
        self.inumy, self.ideny = np.asarray(m.row_num, float), np.asarray(m.row_den, float)
        self.delta = delta

    def nrpci(self, x, y, z):  # ref:c/rpc.c:337-348
        return _pol20(self.inumx, x, y, z) / _pol20(self.idenx, x, y, z), _pol20(self.inumy, x, y, z) / _pol20(self.ideny, x, y, z)

    def nrpc_iterative(self, x, y, z):  # ref:c/rpc.c:372-408, vectorised: converged entries stop moving
        x = np.asarray(x, float); y = np.asarray(y, float); z = np.asarray(z, float)
        delta = self.delta if self.delta else 1.0
        lon = np.full(x.shape, -delta); lat = np.full(x.shape, -delta)
        eps = 2.0 * delta
        x0 = self.nrpci(lon, lat, z); x1 = self.nrpci(lon + eps, lat, z); x2 = self.nrpci(lon, lat + eps, z)
        for _ in range(LOC_MAXIT):
            act = (x0[0] - x) ** 2 + (x0[1] - y) ** 2 > LOC_TOL
            if not act.any():
                break
            u0, u1 = x - x0[0], y - x0[1]
            e10, e11 = x1[0] - x0[0], x1[1] - x0[1]
            e20, e21 = x2[0] - x0[0], x2[1] - x0[1]
            det = e10 * e21 - e11 * e20        # ref:c/rpc.c:359-370
            a0 = (e21 * u0 - e20 * u1) / det
            a1 = (-e11 * u0 + e10 * u1) / det
            lon = np.where(act, lon + a0 * eps, lon)
            lat = np.where(act, lat + a1 * eps, lat)
            eps = 0.1
            n0 = self.nrpci(lon, lat, z); n1 = self.nrpci(lon + eps, lat, z); n2 = self.nrpci(lon, lat + eps, z)
            x0 = (np.where(act, n0[0], x0[0]), np.where(act, n0[1], x0[1]))
            x1 = (np.where(act, n1[0], x1[0]), np.where(act, n1[1], x1[1]))
            x2 = (np.where(act, n2[0], x2[0]), np.where(act, n2[1], x2[1]))
        return lon, lat

    def eval_rpc(self, x, y, z):  # localisation, ref:c/rpc.c:428-438
        nx, ny, nz = (x - self.offset[0]) / self.scale[0], (y - self.offset[1]) / self.scale[1], (z - self.offset[2]) / self.scale[2]
        a, b = self.nrpc_iterative(nx, ny, nz)
        return a * self.iscale[0] + self.ioffset[0], b * self.iscale[1] + self.ioffset[1]

    def eval_rpci(self, x, y, z):  # projection, ref:c/rpc.c:441-451
        nx, ny, nz = (x - self.ioffset[0]) / self.iscale[0], (y - self.ioffset[1]) / self.iscale[1], (z - self.ioffset[2]) / self.iscale[2]
        a, b = self.nrpci(nx, ny, nz)
        return a * self.scale[0] + self.offset[0], b * self.scale[1] + self.offset[1]


def _rpc_pair(ra, rb, x, y, z):  # ref:c/rpc.c:454-461
    lon, lat = ra.eval_rpc(x, y, z)
    return rb.eval_rpci(lon, lat, z)


def rpc_height(ra, rb, xa, ya, xb, yb):
    """ref:c/rpc.c:480-514, vectorised over the correspondences (entries stop at their own |lambda| < 1e-5)."""
    xa = np.asarray(xa, float)
    h = np.zeros(xa.shape); err = np.zeros(xa.shape)
    act = np.ones(xa.shape, bool)
    for _ in range(RPCH_MAXIT):
        if not act.any():
            break
        idx = np.nonzero(act)[0]
        hs = h[idx]
        p = _rpc_pair(ra, rb, xa[idx], ya[idx], hs)
        q = _rpc_pair(ra, rb, xa[idx], ya[idx], hs + 1.0)
        a0, a1 = q[0] - p[0], q[1] - p[1]
        b0, b1 = xb[idx] - p[0], yb[idx] - p[1]
        lam = (a0 * b0 + a1 * b1) / (a0 * a0 + a1 * a1)
        z0, z1 = p[0] + lam * a0, p[1] + lam * a1
        err[idx] = np.hypot(z0 - xb[idx], z1 - yb[idx])
        h[idx] = hs + lam * 1.0
        act[idx] = ~(np.abs(lam) < RPCH_LAMBDA_STOP)
    return h, err


def stereo_corresp_to_lonlatalt(rpc1, rpc2, pts1, pts2):
    """ref:bundle_adjust/s2p/triangulation.py:82-135 + ref:c/disp_to_h.c:40-64: keypoints go through float32."""
    ra, rb = _Rpc(rpc1, 0.1), _Rpc(rpc2, 0.1)
    a = np.asarray(pts1).astype(np.float32).astype(np.float64)
    b = np.asarray(pts2).astype(np.float32).astype(np.float64)
    z, err = rpc_height(ra, rb, a[:, 0], a[:, 1], b[:, 0], b[:, 1])
    lon, lat = ra.eval_rpc(a[:, 0], a[:, 1], z)
    return np.stack([lon, lat, z], axis=1), err.astype(np.float32)


def latlon_to_ecef(lat, lon, alt):
    """ref:bundle_adjust/geo_utils.py:218-233."""
    rad_lat = lat * (np.pi / 180.0); rad_lon = lon * (np.pi / 180.0)
    a = 6378137.0
    f = 1 / 298.257223563
    e2 = 1 - (1 - f) * (1 - f)
    v = a / np.sqrt(1 - e2 * np.sin(rad_lat) * np.sin(rad_lat))
    return ((v + alt) * np.cos(rad_lat) * np.cos(rad_lon), (v + alt) * np.cos(rad_lat) * np.sin(rad_lon),
            (v * (1 - e2) + alt) * np.sin(rad_lat))


def rpc_triangulation(rpc1, rpc2, pts1, pts2):
    """ref:ft_triangulate.py:37-54."""
    lla, err = stereo_corresp_to_lonlatalt(rpc1, rpc2, pts1, pts2)
    x, y, z = latlon_to_ecef(lla[:, 1], lla[:, 0], lla[:, 2])
    return np.vstack((x, y, z)).T, err


def init_pts3d(C, cameras, cam_model, pairs_to_triangulate, triangulate=None):
    """ref:ft_triangulate.py:57-127: float32 running mean over the pairs in list order.  `triangulate(c_i, c_j, obs_i, obs_j)` overrides
    the per-pair triangulation (tests use it to feed the same float64 points to both sides)."""
    n_pts, n_cam = C.shape[1], C.shape[0] // 2
    avg = np.zeros((n_pts, 3), dtype=np.float32)
    cnt = np.zeros(n_pts, dtype=np.float32)
    mask = ~np.isnan(C[::2])
    for c_i, c_j in pairs_to_triangulate:
        if not (c_i < n_cam and c_j < n_cam):
            continue
        t = np.where(mask[c_i] & mask[c_j])[0]
        if t.shape[0] == 0:
            continue
        oi = C[2 * c_i:2 * c_i + 2, t].T; oj = C[2 * c_j:2 * c_j + 2, t].T
        if triangulate is not None:
            new = triangulate(c_i, c_j, oi, oj)
        elif cam_model in ("affine", "perspective"):
            new = linear_triangulation_multiple_pts(cameras[c_i], cameras[c_j], oi, oj)
        else:
            new, _ = rpc_triangulation(cameras[c_i], cameras[c_j], oi, oj)
        new32 = np.zeros((n_pts, 3), dtype=np.float32)
        new32[t] = new
        cnt[t] += 1.0
        avg[t] = ((cnt[t, np.newaxis] - 1.0) * avg[t] + new32[t]) / cnt[t, np.newaxis]
    return avg
