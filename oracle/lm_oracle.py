"""
ORACLE -- test infrastructure, not product code.

Second half of the CPU oracle: everything the device solver computes that the reference itself never forms,
restated in numpy so each HIP kernel can be checked in isolation:

  * analytic per-observation Jacobian blocks J_c (2 x n_p), J_p (2 x 3) of the three projection models of
    ref:bundle_adjust/ba_core.py:59-154 (math in SURVEY.md appendix B); validated against 3-point finite
    differences of the reference's `fun` captured in tests/golden/fun_*.npz (tools/gen_golden.py);
  * robust-loss row scaling exactly as scipy applies it (scipy:optimize/_lsq/least_squares.py:181-238,
    scipy:optimize/_lsq/common.py:720-731);
  * normal-equation blocks, Schur complement, damped step, 2-D subspace quantities;
  * `OracleEngine`: the engine interface of satba/trf.py implemented on the CPU, so the host-side
    trust-region driver and the multi-rank sharding / all-reduce placement can be exercised without a GPU
    (gloo, world_size 2) and so GPU phases can be compared one by one.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

from . import ba_oracle as O

EPS = np.finfo(float).eps
LOSSES = ("linear", "soft_l1", "huber", "cauchy", "arctan")


# ----------------------------------------------------------------------------- analytic Jacobians

def _rot_chain(X, ang):
    """y1 = Rx X, y2 = Ry y1, y3 = Rz y2 and the three angle derivatives of y3 (each (K, 3))."""
    ca, sa = np.cos(ang[:, 0]), np.sin(ang[:, 0])
    cb, sb = np.cos(ang[:, 1]), np.sin(ang[:, 1])
    cg, sg = np.cos(ang[:, 2]), np.sin(ang[:, 2])
    x0, y0, z0 = X[:, 0], X[:, 1], X[:, 2]
    y1 = np.stack((x0, ca * y0 - sa * z0, sa * y0 + ca * z0), 1)
    y2 = np.stack((cb * y1[:, 0] + sb * y1[:, 2], y1[:, 1], -sb * y1[:, 0] + cb * y1[:, 2]), 1)
    y3 = np.stack((cg * y2[:, 0] - sg * y2[:, 1], sg * y2[:, 0] + cg * y2[:, 1], y2[:, 2]), 1)

    def Ry(v):
        return np.stack((cb * v[:, 0] + sb * v[:, 2], v[:, 1], -sb * v[:, 0] + cb * v[:, 2]), 1)

    def Rz(v):
        return np.stack((cg * v[:, 0] - sg * v[:, 1], sg * v[:, 0] + cg * v[:, 1], v[:, 2]), 1)

    zero = np.zeros_like(x0)
    d_a = Rz(Ry(np.stack((zero, -y1[:, 2], y1[:, 1]), 1)))  # Rz Ry Rx' X
    d_b = Rz(np.stack((y2[:, 2], zero, -y2[:, 0]), 1))  # Rz Ry' y1
    d_g = np.stack((-y3[:, 1], y3[:, 0], zero), 1)  # Rz' y2
    # full rotation matrix rows, R = Rz Ry Rx, as (K, 3, 3)
    R = np.empty((X.shape[0], 3, 3))
    R[:, 0, 0], R[:, 0, 1], R[:, 0, 2] = cg * cb, cg * sb * sa - sg * ca, cg * sb * ca + sg * sa
    R[:, 1, 0], R[:, 1, 1], R[:, 1, 2] = sg * cb, sg * sb * sa + cg * ca, sg * sb * ca - cg * sa
    R[:, 2, 0], R[:, 2, 1], R[:, 2, 2] = -sb, cb * sa, cb * ca
    return y3, (d_a, d_b, d_g), R


def geodetic_with_jacobian(X):
    """(lat_deg, lon_deg, alt) of ref:bundle_adjust/geo_utils.py:236-255 and G = d(lat_deg, lon_deg, alt)/dX, (K, 3, 3)."""
    a, esq = O.WGS84_A, O.WGS84_E ** 2
    b = np.sqrt(a * a * (1 - esq))
    ep2 = (a * a - b * b) / (b * b)
    x, y, z = X[:, 0], X[:, 1], X[:, 2]
    zero, one = np.zeros_like(x), np.ones_like(x)
    p = np.sqrt(x * x + y * y)
    dp = np.stack((x / p, y / p, zero), 1)
    u, w = a * z, b * p
    th = np.arctan2(u, w)
    dth = (w[:, None] * np.stack((zero, zero, a * one), 1) - u[:, None] * b * dp) / (u * u + w * w)[:, None]
    s, c = np.sin(th), np.cos(th)
    num = z + ep2 * b * s ** 3
    den = p - esq * a * c ** 3
    dnum = np.stack((zero, zero, one), 1) + (ep2 * b * 3 * s * s * c)[:, None] * dth
    dden = dp + (esq * a * 3 * c * c * s)[:, None] * dth
    lat = np.arctan2(num, den)
    dlat = (den[:, None] * dnum - num[:, None] * dden) / (num * num + den * den)[:, None]
    lon = np.arctan2(y, x)
    dlon = np.stack((-y, x, zero), 1) / (x * x + y * y)[:, None]
    sl, cl = np.sin(lat), np.cos(lat)
    t = 1 - esq * sl * sl
    Nv = a / np.sqrt(t)
    dN = (a * esq * sl * cl * t ** -1.5)[:, None] * dlat
    alt = p / cl - Nv
    dalt = dp / cl[:, None] + (p * sl / (cl * cl))[:, None] * dlat - dN
    k = 180 / np.pi
    return lat * k, lon * k, alt, np.stack((dlat * k, dlon * k, dalt), 1)


def _monomials_with_grad(L, P, H):
    """m (20, K) and dm/dL, dm/dP, dm/dH in RPC00B order (see satba/rpc_model.py)."""
    o, z = np.ones_like(L), np.zeros_like(L)
    m = np.stack([o, L, P, H, L * P, L * H, P * H, L * L, P * P, H * H, P * L * H, L ** 3, L * P * P, L * H * H,
                  L * L * P, P ** 3, P * H * H, L * L * H, P * P * H, H ** 3])
    dL = np.stack([z, o, z, z, P, H, z, 2 * L, z, z, P * H, 3 * L * L, P * P, H * H, 2 * L * P, z, z, 2 * L * H, z, z])
    dP = np.stack([z, z, o, z, L, z, H, z, 2 * P, z, L * H, z, 2 * L * P, z, L * L, 3 * P * P, H * H, z, 2 * P * H, z])
    dH = np.stack([z, z, z, o, z, L, P, z, z, 2 * H, P * L, z, z, 2 * L * H, z, z, 2 * P * H, L * L, P * P, 3 * H * H])
    return m, dL, dP, dH


def rpc_project_with_jacobian(rpc, lat, lon, alt):
    """(col, row) and d(col, row)/d(lat_deg, lon_deg, alt) as (K, 2, 3)."""
    L = (lon - rpc.lon_offset) / rpc.lon_scale
    P = (lat - rpc.lat_offset) / rpc.lat_scale
    H = (alt - rpc.alt_offset) / rpc.alt_scale
    m, dL, dP, dH = _monomials_with_grad(L, P, H)
    out, jac = [], []
    for num, den, scale, off in ((rpc.col_num, rpc.col_den, rpc.col_scale, rpc.col_offset),
                                 (rpc.row_num, rpc.row_den, rpc.row_scale, rpc.row_offset)):
        num, den = np.asarray(num), np.asarray(den)
        pn, pd = num @ m, den @ m
        out.append(scale * pn / pd + off)
        g = [scale * ((num @ dm) * pd - pn * (den @ dm)) / (pd * pd) for dm in (dL, dP, dH)]
        # order the columns as (lat, lon, alt), each divided by its normalisation scale
        jac.append(np.stack((g[1] / rpc.lat_scale, g[0] / rpc.lon_scale, g[2] / rpc.alt_scale), 1))
    return np.stack(out, 1), np.stack(jac, 1)


def jacobian_blocks(v, p):
    """
    Unweighted projection and Jacobian blocks at v: proj (K, 2) float64, Jc (K, 2, n_params), Jp (K, 2, 3).
    Columns of frozen cameras / points are NOT zeroed here (see `weighted_system`).
    """
    pts3d, cam_params = O.unpack(v, p)
    cp = cam_params[p.cam_ind]
    X = pts3d[p.pts_ind]
    n_p = p.n_params
    K = X.shape[0]
    if p.cam_model == "affine":
        y3, dth, R = _rot_chain(X, cp[:, :3])
        fx, fy, sk = cp[:, 5], cp[:, 6], cp[:, 7]
        A = np.zeros((K, 2, 3))
        A[:, 0, 0], A[:, 0, 1], A[:, 1, 1] = fx, sk, fy
        q = y3[:, :2] + cp[:, 3:5]
        proj = np.einsum("kij,kj->ki", A[:, :, :2], q)
        Jfull = np.concatenate([np.einsum("kij,kj->ki", A, d)[:, :, None] for d in dth] + [A[:, :, :2]], axis=2)
        Jp = np.einsum("kij,kjl->kil", A, R)
    elif p.cam_model == "perspective":
        y3, dth, R = _rot_chain(X, cp[:, :3])
        q = y3 + cp[:, 3:6]
        fx, fy, sk, cx, cy = cp[:, 6], cp[:, 7], cp[:, 8], cp[:, 9], cp[:, 10]
        K2 = np.zeros((K, 2, 3))
        K2[:, 0, 0], K2[:, 0, 1], K2[:, 0, 2], K2[:, 1, 1], K2[:, 1, 2] = fx, sk, cx, fy, cy
        proj = np.einsum("kij,kj->ki", K2, q) / q[:, 2:3]
        D = K2.copy()
        D[:, :, 2] -= proj
        D /= q[:, 2][:, None, None]
        Jfull = np.concatenate([np.einsum("kij,kj->ki", D, d)[:, :, None] for d in dth] + [D], axis=2)
        Jp = np.einsum("kij,kjl->kil", D, R)
    else:
        C = cp[:, 6:9]
        d0 = X - cp[:, 3:6] - C
        y3, dth, R = _rot_chain(d0, cp[:, :3])
        Xa = y3 + C
        lat, lon, alt, G = geodetic_with_jacobian(Xa)
        proj = np.zeros((K, 2))
        Dp = np.zeros((K, 2, 3))
        for c in np.unique(p.cam_ind).tolist():
            sel = p.cam_ind == c
            proj[sel], Dp[sel] = rpc_project_with_jacobian(p.cameras[c], lat[sel], lon[sel], alt[sel])
        D = np.einsum("kij,kjl->kil", Dp, G)  # d proj / d X'
        Jp = np.einsum("kij,kjl->kil", D, R)
        Jfull = np.concatenate([np.einsum("kij,kj->ki", D, d)[:, :, None] for d in dth] + [-Jp], axis=2)
    return proj, Jfull[:, :, :n_p], Jp


# ----------------------------------------------------------------------------- robust loss (scipy semantics)

def loss_rho(loss, z):
    """rho(z), rho'(z), rho''(z) of scipy:optimize/_lsq/least_squares.py:172-207."""
    if loss == "soft_l1":
        t = 1 + z
        return 2 * (np.sqrt(t) - 1), t ** -0.5, -0.5 * t ** -1.5
    if loss == "huber":
        m = z <= 1
        zs = np.where(m, 1.0, z)
        return np.where(m, z, 2 * np.sqrt(zs) - 1), np.where(m, 1.0, zs ** -0.5), np.where(m, 0.0, -0.5 * zs ** -1.5)
    if loss == "cauchy":
        t = 1 + z
        return np.log1p(z), 1 / t, -1 / t ** 2
    if loss == "arctan":
        t = 1 + z * z
        return np.arctan(z), 1 / t, -2 * z / t ** 2
    raise ValueError(loss)


def robust_scale(f, loss, f_scale):
    """cost, scaled residuals and Jacobian row factors (scipy:optimize/_lsq/common.py:720-731)."""
    if loss == "linear":
        return 0.5 * np.dot(f, f), f, np.ones_like(f)
    z = (f / f_scale) ** 2
    r0, r1, r2 = loss_rho(loss, z)
    r2 = r2 / f_scale ** 2
    js = np.sqrt(np.maximum(r1 + 2 * r2 * f * f, EPS))
    return 0.5 * f_scale ** 2 * np.sum(r0), f * r1 / js, js


def robust_cost(f, loss, f_scale):
    if loss == "linear":
        return 0.5 * np.dot(f, f)
    return 0.5 * f_scale ** 2 * np.sum(loss_rho(loss, (f / f_scale) ** 2)[0])


def weighted_system(v, p, loss="linear", f_scale=1.0, rpc_f32=True):
    """
    Everything one linearisation produces: true residuals f (2K), cost, scaled residuals fs (K, 2) and row-scaled,
    weighted Jacobian blocks with frozen cameras / points zeroed.
    """
    proj, Jc, Jp = jacobian_blocks(v, p)
    if p.cam_model == "rpc" and rpc_f32:
        proj = proj.astype(np.float32).astype(np.float64)  # ref:bundle_adjust/ba_core.py:150
    w2 = np.repeat(p.pts2d_w, 2)
    f = w2 * (proj - p.pts2d).ravel()
    cost, fs, js = robust_scale(f, loss, f_scale)
    rows = (w2 * js).reshape(-1, 2)[:, :, None]
    Jc = Jc * rows * (p.cam_ind >= p.n_cam_fix)[:, None, None]
    Jp = Jp * rows * (p.pts_ind >= p.n_pts_fix)[:, None, None]
    return f, cost, fs.reshape(-1, 2), Jc, Jp


def normal_blocks(fs, Jc, Jp, p):
    """U (M, n_p, n_p), g_c (M, n_p), V (N, 3, 3), g_p (N, 3)."""
    n_p = p.n_params
    U = np.zeros((p.n_cam, n_p, n_p))
    gc = np.zeros((p.n_cam, n_p))
    V = np.zeros((p.n_pts, 3, 3))
    gp = np.zeros((p.n_pts, 3))
    np.add.at(U, p.cam_ind, np.einsum("kri,krj->kij", Jc, Jc))
    np.add.at(gc, p.cam_ind, np.einsum("kri,kr->ki", Jc, fs))
    np.add.at(V, p.pts_ind, np.einsum("kri,krj->kij", Jp, Jp))
    np.add.at(gp, p.pts_ind, np.einsum("kri,kr->ki", Jp, fs))
    return U, gc, V, gp


def schur_parts(Jc, Jp, V, gp, lam, scale_inv_p, p):
    """Local Schur contributions: -sum_p W Vl^-1 W^T (n_c, n_c), -sum_p W Vl^-1 g_p (n_c,), and Vl^-1 (N, 3, 3)."""
    from scipy.sparse import coo_matrix

    n_p, M, N = p.n_params, p.n_cam, p.n_pts
    Vl = V + lam * np.einsum("ni,ij->nij", scale_inv_p.reshape(N, 3) ** 2, np.eye(3))
    Vinv = np.linalg.inv(Vl)
    W = np.einsum("kri,krj->kij", Jc, Jp)  # (K, n_p, 3)
    rows = (p.cam_ind[:, None, None] * n_p + np.arange(n_p)[None, :, None]) + np.zeros((1, 1, 3), dtype=int)
    cols = (p.pts_ind[:, None, None] * 3 + np.arange(3)[None, None, :]) + np.zeros((1, n_p, 1), dtype=int)
    E = coo_matrix((W.ravel(), (rows.ravel(), cols.ravel())), shape=(M * n_p, 3 * N)).tocsr()
    T = np.einsum("kij,kjl->kil", W, Vinv[p.pts_ind])
    Y = coo_matrix((T.ravel(), (rows.ravel(), cols.ravel())), shape=(M * n_p, 3 * N)).tocsr()
    S = -(Y @ E.T).toarray()
    rhs = -(Y @ gp.ravel())
    return S, rhs, Vinv


# ----------------------------------------------------------------------------- engine

class OracleEngine:
    """
    CPU implementation of the solver-engine interface (see satba/trf.py for the contract).  Holds one shard:
    all cameras, a contiguous range of points and their observations.  `xb` is a CPU torch tensor so that
    torch.distributed (gloo) can reduce it in place.
    """

    HDR_FIXED = 16

    def __init__(self, p, rank=0, world=1, rpc_f32=True):
        import torch

        self.p, self.rank, self.world, self.rpc_f32 = p, rank, world, rpc_f32
        self.n_cam, self.n_p, self.n_pts = p.n_cam, p.n_params, p.n_pts
        self.n_c = self.n_cam * self.n_p
        self.n = self.n_c + 3 * self.n_pts
        self.n_total = self.n
        self.hdr = self.HDR_FIXED + world + (world % 2)
        self.xb = torch.zeros(self.hdr + self.n_c * self.n_c + self.n_c, dtype=torch.float64)
        self._xb = self.xb.numpy()
        self.len_lin = self.hdr + self.n_cam * self.n_p ** 2 + self.n_c
        self.len_schur = self.hdr + self.n_c ** 2 + self.n_c
        self.x = p.params_opt.astype(np.float64).copy()
        self.scale_inv = None
        self.lead = 1.0 if rank == 0 else 0.0  # camera-side terms are contributed once

    # -- state transfer
    def set_x(self, x):
        self.x = np.array(x, dtype=np.float64)

    def get_x(self):
        return self.x.copy()

    def read_header(self):
        return self._xb[: self.hdr].copy()

    def configure(self, loss, f_scale):
        self.loss, self.f_scale = loss, float(f_scale)

    # -- phases
    def linearize(self):
        p = self.p
        self.f, cost, self.fs, self.Jc, self.Jp = weighted_system(self.x, p, self.loss, self.f_scale, self.rpc_f32)
        U, gc, self.V, self.gp = normal_blocks(self.fs, self.Jc, self.Jp, p)
        b = self._xb
        b[: self.hdr] = 0
        b[0] = cost
        b[self.HDR_FIXED + self.rank] = np.abs(self.gp).max() if self.gp.size else 0.0
        nU = self.n_cam * self.n_p ** 2
        b[self.hdr: self.hdr + nU] = U.ravel()
        b[self.hdr + nU: self.hdr + nU + self.n_c] = gc.ravel()

    def prepare(self, first):
        b = self._xb
        self._keep = np.zeros(7)
        self._keep[0] = b[0]
        self._keep[1] = np.max(b[self.HDR_FIXED: self.HDR_FIXED + self.world])
        nU = self.n_cam * self.n_p ** 2
        self.U = b[self.hdr: self.hdr + nU].reshape(self.n_cam, self.n_p, self.n_p).copy()
        self.gc = b[self.hdr + nU: self.hdr + nU + self.n_c].copy()
        diag = np.concatenate((np.einsum("mii->mi", self.U).ravel(), np.einsum("nii->ni", self.V).ravel()))
        si = np.sqrt(diag)
        if first:
            si[si == 0] = 1.0
            self.scale_inv = si
        else:
            self.scale_inv = np.maximum(si, self.scale_inv)
        self.scale = 1.0 / self.scale_inv
        self.g = np.concatenate((self.gc, self.gp.ravel()))
        self.g_h = self.g * self.scale
        jg = self._jvp(self.scale * self.g_h)
        b[: self.hdr] = 0
        b[1] = self._dot(self.g_h, self.g_h)
        b[2] = np.sum(jg * jg)
        b[3] = self._dot(self.x * self.scale_inv, self.x * self.scale_inv)
        b[4] = self.lead * (np.abs(self.gc).max() if self.gc.size else 0.0)

    def schur(self, lam):
        S, rhs, self.Vinv = schur_parts(self.Jc, self.Jp, self.V, self.gp, lam, self.scale_inv[self.n_c:], self.p)
        if self.rank == 0:
            Ul = self.U + lam * np.einsum("mi,ij->mij", self.scale_inv[: self.n_c].reshape(self.n_cam, self.n_p) ** 2,
                                          np.eye(self.n_p))
            for m in range(self.n_cam):
                sl = slice(m * self.n_p, (m + 1) * self.n_p)
                S[sl, sl] += Ul[m]
            rhs = rhs + self.gc
        b = self._xb
        b[: self.hdr] = 0
        b[self.hdr: self.hdr + self.n_c ** 2] = S.ravel()
        b[self.hdr + self.n_c ** 2: self.len_schur] = rhs

    def schur_auto(self, Delta, lam_floor=0.0):
        """Damping from the (all-reduced) prepare header, as satba/trf.py computes it on the host for schur()."""
        b, k = self._xb, self._keep
        gh_sq, jg_sq, xs_sq = b[1], b[2], b[3]
        k[1] = max(k[1], b[4])
        k[2:5] = gh_sq, jg_sq, xs_sq
        if not Delta > 0:
            Delta = np.sqrt(xs_sq) or 1.0
        a, bb, ub = 0.5 * jg_sq, -gh_sq, Delta / np.sqrt(gh_sq)
        best = min(0.0, a * ub * ub + bb * ub)
        if a != 0:
            ext = -0.5 * bb / a
            if 0 < ext < ub:
                best = min(best, a * ext * ext + bb * ext)
        lam = -best / Delta ** 2
        if not lam >= lam_floor:
            lam = lam_floor
        k[5], k[6] = lam, Delta
        self.schur(lam)

    def solve(self):
        b = self._xb
        S = b[self.hdr: self.hdr + self.n_c ** 2].reshape(self.n_c, self.n_c)
        rhs = b[self.hdr + self.n_c ** 2: self.len_schur]
        fail = 0.0
        try:
            L = np.linalg.cholesky(0.5 * (S + S.T))
            dc = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
        except np.linalg.LinAlgError:
            dc, fail = np.zeros(self.n_c), 1.0
        u = np.einsum("kri,ki->kr", self.Jc, dc.reshape(self.n_cam, self.n_p)[self.p.cam_ind])
        wtd = np.zeros((self.n_pts, 3))
        np.add.at(wtd, self.p.pts_ind, np.einsum("kri,kr->ki", self.Jp, u))
        dp = np.einsum("nij,nj->ni", self.Vinv, self.gp - wtd)
        self.gn_h = np.concatenate((dc, dp.ravel())) * self.scale_inv
        b[: self.hdr] = 0
        b[1] = self._dot(self.g_h, self.g_h)
        b[2] = self._dot(self.g_h, self.gn_h)
        b[3] = self._dot(self.gn_h, self.gn_h)
        b[4] = self.lead * fail
        b[8:15] = self.lead * getattr(self, "_keep", np.zeros(7))

    def subspace(self, alpha, inv_norm_g):
        self.q1 = self.g_h * inv_norm_g
        self.w = self.gn_h - alpha * self.g_h
        b = self._xb
        b[: self.hdr] = 0
        b[1] = self._dot(self.w, self.w)
        b[2] = self._dot(self.w, self.q1)
        b[6] = self._dot(self.g_h, self.w)

    def subspace_products(self):
        j1 = self._jvp(self.scale * self.q1)
        j2 = self._jvp(self.scale * self.w)
        b = self._xb
        b[: self.hdr] = 0
        b[3], b[4], b[5] = np.sum(j1 * j1), np.sum(j1 * j2), np.sum(j2 * j2)

    def trial(self, p0, p1):
        step = self.scale * (p0 * self.q1 + p1 * self.w)
        self.x_new = self.x + step
        f_new = O.fun(self.x_new, self.p, np.float32 if self.rpc_f32 else np.float64)
        b = self._xb
        b[: self.hdr] = 0
        b[1] = robust_cost(f_new, self.loss, self.f_scale)
        b[2] = self._dot(step, step)
        b[3] = self._dot(self.x, self.x)

    def trial_gn(self, ca, cb):
        step = self.scale * (ca * self.g_h + cb * self.gn_h)
        self.x_new = self.x + step
        f_new = O.fun(self.x_new, self.p, np.float32 if self.rpc_f32 else np.float64)
        b = self._xb
        b[: self.hdr] = 0
        b[1] = robust_cost(f_new, self.loss, self.f_scale)
        b[2] = self._dot(step, step)
        b[3] = self._dot(self.x, self.x)

    def accept(self):
        self.x = self.x_new

    def residuals(self):
        return O.fun(self.x, self.p, np.float32 if self.rpc_f32 else np.float64)

    # -- helpers
    def _dot(self, a, b):
        return self.lead * np.dot(a[: self.n_c], b[: self.n_c]) + np.dot(a[self.n_c:], b[self.n_c:])

    def _jvp(self, v):
        vc = v[: self.n_c].reshape(self.n_cam, self.n_p)[self.p.cam_ind]
        vp = v[self.n_c:].reshape(self.n_pts, 3)[self.p.pts_ind]
        return np.einsum("kri,ki->kr", self.Jc, vc) + np.einsum("kri,ki->kr", self.Jp, vp)
