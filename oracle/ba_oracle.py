"""
ORACLE -- test infrastructure, not product code.

CPU (numpy + the installed scipy) restatement of the reference's bundle-adjustment hot path:
residual function `fun`, Jacobian sparsity pattern and the `scipy.optimize.least_squares` driver,
following ref:bundle_adjust/ba_core.py:36-183 (projections, fun), :186-219 (sparsity), :244-332 (driver),
ref:bundle_adjust/ba_params.py:221-257 (unpacking of the variable vector),
ref:bundle_adjust/cam_utils.py:217-231 and ref:bundle_adjust/geo_utils.py:236-255 (RPC projection chain).

Third-party arithmetic on the path that is NOT under /root/reference:
  * scipy (unpinned in ref:requirements.txt:10; 1.15.3 in this image) -- called here exactly as the
    reference calls it (ref:bundle_adjust/ba_core.py:284-297), not restated;
  * rpcm.RPCModel.projection (ref:requirements.txt:9, branch localization-origin, not vendored) -- restated in
    `rpc_projection` from the in-tree statements of the same polynomial (ref:bundle_adjust/ba_rpcfit.py:17-44,
    ref:c/rpc.c:279-298, 442-452) and cross-checked against the reference's own C (oracle/_ref/librpc.so,
    built by oracle/Makefile from ref:c/rpc.c).

Pinning: the reference's tests hold no vectors for this path, so this file is pinned
against outputs of the reference itself, captured by tools/gen_golden.py (which imports
/root/reference in the build container) into tests/golden/*.npz; tests/test_oracle_golden.py replays them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

WGS84_A = 6378137.0
WGS84_E = 8.1819190842622e-2


# ----------------------------------------------------------------------------- projections

def rotate_euler(pts, euler):
    """Row-wise Rz(g) Ry(b) Rx(a) X, in the reference's operation order (ref:bundle_adjust/ba_core.py:36-56)."""
    ca, sa = np.cos(euler[:, 0]), np.sin(euler[:, 0])
    cb, sb = np.cos(euler[:, 1]), np.sin(euler[:, 1])
    cg, sg = np.cos(euler[:, 2]), np.sin(euler[:, 2])
    x0, y0, z0 = pts[:, 0], pts[:, 1], pts[:, 2]
    x1, y1, z1 = x0, ca * y0 - sa * z0, sa * y0 + ca * z0
    x2, y2, z2 = cb * x1 + sb * z1, y1, -sb * x1 + cb * z1
    return np.stack((cg * x2 - sg * y2, sg * x2 + cg * y2, z2), axis=1)


def project_affine(pts3d, cam_params, pts_ind, cam_ind):
    """ref:bundle_adjust/ba_core.py:59-81; cam row = [a, b, g, t0, t1, fx, fy, skew]."""
    cp = cam_params[cam_ind]
    q = rotate_euler(pts3d[pts_ind], cp[:, :3])[:, :2] + cp[:, 3:5]
    fx, fy, skew = cp[:, 5], cp[:, 6], cp[:, 7]
    return np.stack((fx * q[:, 0] + skew * q[:, 1], fy * q[:, 1]), axis=1)


def project_perspective(pts3d, cam_params, pts_ind, cam_ind):
    """ref:bundle_adjust/ba_core.py:84-107; cam row = [a, b, g, t0, t1, t2, fx, fy, skew, cx, cy]."""
    cp = cam_params[cam_ind]
    q = rotate_euler(pts3d[pts_ind], cp[:, :3]) + cp[:, 3:6]
    fx, fy, skew, cx, cy = cp[:, 6], cp[:, 7], cp[:, 8], cp[:, 9], cp[:, 10]
    u = fx * q[:, 0] + skew * q[:, 1] + cx * q[:, 2]
    v = fy * q[:, 1] + cy * q[:, 2]
    return np.stack((u / q[:, 2], v / q[:, 2]), axis=1)


def adjust_pts3d(pts3d, Rt_vec):
    """X' = R (X - T - C) + C  (ref:bundle_adjust/ba_core.py:110-130); Rt_vec rows = [angles, T, C]."""
    d = pts3d - Rt_vec[:, 3:6]
    d = d - Rt_vec[:, 6:9]
    return rotate_euler(d, Rt_vec[:, :3]) + Rt_vec[:, 6:9]


def ecef_to_latlon(x, y, z):
    """ref:bundle_adjust/geo_utils.py:236-255."""
    a, e = WGS84_A, WGS84_E
    asq, esq = a ** 2, e ** 2
    b = np.sqrt(asq * (1 - esq))
    bsq = b ** 2
    ep = np.sqrt((asq - bsq) / bsq)
    p = np.sqrt(x ** 2 + y ** 2)
    th = np.arctan2(a * z, b * p)
    lon = np.arctan2(y, x)
    lat = np.arctan2(z + ep ** 2 * b * np.sin(th) ** 3, p - esq * a * np.cos(th) ** 3)
    N = a / np.sqrt(1 - esq * np.sin(lat) ** 2)
    alt = p / np.cos(lat) - N
    return lat * 180 / np.pi, lon * 180 / np.pi, alt


def rpc_polynomial(c, lat, lon, alt):
    """20-term cubic in the order of ref:bundle_adjust/ba_rpcfit.py:17-44 with x = lat, y = lon, z = alt."""
    x, y, z = lat, lon, alt
    return (c[0] + c[1] * y + c[2] * x + c[3] * z + c[4] * y * x + c[5] * y * z + c[6] * x * z + c[7] * y * y
            + c[8] * x * x + c[9] * z * z + c[10] * x * y * z + c[11] * y * y * y + c[12] * y * x * x
            + c[13] * y * z * z + c[14] * y * y * x + c[15] * x * x * x + c[16] * x * z * z + c[17] * y * y * z
            + c[18] * x * x * z + c[19] * z * z * z)


def rpc_projection(rpc, lon, lat, alt):
    """(col, row) of rpcm.RPCModel.projection, call site ref:bundle_adjust/cam_utils.py:229."""
    nlon = (lon - rpc.lon_offset) / rpc.lon_scale
    nlat = (lat - rpc.lat_offset) / rpc.lat_scale
    nalt = (alt - rpc.alt_offset) / rpc.alt_scale
    col = rpc_polynomial(rpc.col_num, nlat, nlon, nalt) / rpc_polynomial(rpc.col_den, nlat, nlon, nalt)
    row = rpc_polynomial(rpc.row_num, nlat, nlon, nalt) / rpc_polynomial(rpc.row_den, nlat, nlon, nalt)
    return col * rpc.col_scale + rpc.col_offset, row * rpc.row_scale + rpc.row_offset


def project_rpc(pts3d, rpcs, cam_params, pts_ind, cam_ind, store_dtype=np.float32):
    """
    ref:bundle_adjust/ba_core.py:133-154, including the float32 store of the projections (:150).
    store_dtype=np.float64 gives the unquantised function (used to validate analytic Jacobians).
    """
    X = adjust_pts3d(pts3d[pts_ind], cam_params[cam_ind])
    out = np.zeros((pts_ind.shape[0], 2), dtype=store_dtype)
    for c in np.unique(cam_ind).tolist():
        sel = cam_ind == c
        lat, lon, alt = ecef_to_latlon(X[sel, 0], X[sel, 1], X[sel, 2])
        col, row = rpc_projection(rpcs[c], lon, lat, alt)
        out[sel] = np.vstack((col, row)).T
    return out


# ----------------------------------------------------------------------------- fun and driver

def unpack(v, p):
    """ref:bundle_adjust/ba_params.py:221-257 without the in-place write into v (K / COMMON_K unsupported)."""
    n_c = p.n_cam * p.n_params
    pts3d = v[n_c:].reshape((p.n_pts, 3)).copy()
    if p.n_pts_fix > 0:
        pts3d[: p.n_pts_fix] = p.pts3d[: p.n_pts_fix]
    cam_opt = v[:n_c].reshape((p.n_cam, p.n_params)).copy()
    if p.n_cam_fix > 0:
        cam_opt[: p.n_cam_fix] = p.cam_params[: p.n_cam_fix, : p.n_params]
    return pts3d, np.hstack((cam_opt, p.cam_params[:, p.n_params:]))


def project(v, p, rpc_store_dtype=np.float32):
    pts3d, cam_params = unpack(v, p)
    if p.cam_model == "perspective":
        return project_perspective(pts3d, cam_params, p.pts_ind, p.cam_ind)
    if p.cam_model == "affine":
        return project_affine(pts3d, cam_params, p.pts_ind, p.cam_ind)
    return project_rpc(pts3d, p.cameras, cam_params, p.pts_ind, p.cam_ind, rpc_store_dtype)


def fun(v, p, rpc_store_dtype=np.float32):
    """Residual vector [x0, y0, x1, y1, ...] = repeat(w, 2) * (proj - obs)  (ref:bundle_adjust/ba_core.py:157-183)."""
    proj = project(v, p, rpc_store_dtype)
    return np.repeat(p.pts2d_w, 2, axis=0) * (proj - p.pts2d).ravel()


def jacobian_sparsity(p):
    """
    Structure of ref:bundle_adjust/ba_core.py:186-219: rows 2k, 2k+1 of observation k touch the n_params columns of
    its camera and the 3 columns of its point.  Built in CSR directly (the reference fills a lil_matrix).
    """
    from scipy.sparse import csr_matrix

    K, n_p = p.pts_ind.size, p.n_params
    n = p.n_cam * n_p + p.n_pts * 3
    cols = np.hstack((p.cam_ind[:, None] * n_p + np.arange(n_p), p.n_cam * n_p + p.pts_ind[:, None] * 3 + np.arange(3)))
    cols = np.repeat(cols, 2, axis=0).ravel()
    indptr = np.arange(0, 2 * K * (n_p + 3) + 1, n_p + 3)
    return csr_matrix((np.ones(cols.size, dtype=int), cols, indptr), shape=(2 * K, n))


DEFAULT_LS = {"loss": "linear", "ftol": 1e-4, "xtol": 1e-10, "f_scale": 1.0, "max_iter": 300, "verbose": 1}


def solve_scipy(p, ls_params=None, tight=False, x0=None, rpc_store_dtype=np.float32, max_nfev=None):
    """
    The reference's solver call (ref:bundle_adjust/ba_core.py:284-297).  tight=True applies the parity protocol of
    SURVEY.md section 8c: ftol = xtol = gtol = 1e-15 and LSMR atol = btol = 1e-12, everything else unchanged.
    Returns the scipy OptimizeResult.
    """
    from scipy.optimize import least_squares

    cfg = dict(DEFAULT_LS)
    cfg.update(ls_params or {})
    kw = dict(jac_sparsity=jacobian_sparsity(p), verbose=0, x_scale="jac", method="trf", ftol=cfg["ftol"],
              xtol=cfg["xtol"], loss=cfg["loss"], f_scale=cfg["f_scale"], max_nfev=cfg["max_iter"])
    if tight:
        kw.update(ftol=1e-15, xtol=1e-15, gtol=1e-15, tr_options={"atol": 1e-12, "btol": 1e-12})
    if max_nfev is not None:
        kw["max_nfev"] = max_nfev
    x0 = p.params_opt.copy() if x0 is None else x0.copy()
    return least_squares(lambda v: fun(v, p, rpc_store_dtype), x0, **kw)


def reprojection_error(residuals, pts2d_w=None):
    """ref:bundle_adjust/ba_core.py:335-349."""
    w = np.ones(residuals.size) if pts2d_w is None else np.repeat(pts2d_w, 2, axis=0)
    return np.linalg.norm(np.abs(residuals / w).reshape(-1, 2), axis=1)
