"""
TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's RPC re-fit.  Only tests/ may import this module.

What it restates (reference file:line):
  * weighted_lsq           ref:bundle_adjust/ba_rpcfit.py:88-153   regularised iteratively re-weighted least squares, 39 unknowns per axis
  * poly_vect              ref:bundle_adjust/ba_rpcfit.py:17-44    (RPC00B term order without the constant)
  * scaling_params / initialize_rpc   ref:...ba_rpcfit.py:156-198
  * check_errors           ref:...ba_rpcfit.py:357-370

Pinned on tests/golden/rpcfit.npz: outputs of the imported reference function on seeded correspondences (tools/gen_golden.py rpcfit;
`rpcm.RPCModel` -- branch localization-origin, absent from the image -- is only a container with a `projection` method there and is
supplied by satba.rpc_model.RPCModel).  The normal matrices of this fit are ill-conditioned (cond ~1e13 - 1e17): the reference
inverts them with numpy.linalg.inv, and two correct solvers agree on the fitted PROJECTION (1e-6 px), not on the coefficients.
"""
import numpy as np


def poly_vect(x, y, z):
    return np.array([y, x, z, y * x, y * z, x * z, y * y, x * x, z * z, x * y * z, y * y * y, y * x * x, y * z * z, y * y * x, x * x * x,
                     x * z * z, y * y * z, x * x * z, z * z * z])


def scaling_params(v):
    lo, hi = min(v), max(v)
    scale = (hi - lo) / 2
    return scale, lo + scale


def project(m, lon, lat, alt):
    """m: dict with the fitted model; RPC00B projection (ref:bundle_adjust/ba_rpcfit.py:17-44 term order)."""
    L = (lon - m["lon_offset"]) / m["lon_scale"]; P = (lat - m["lat_offset"]) / m["lat_scale"]; H = (alt - m["alt_offset"]) / m["alt_scale"]
    mono = np.vstack([np.ones_like(L), poly_vect(x=P, y=L, z=H)])
    col = (m["col_num"] @ mono) / (m["col_den"] @ mono) * m["col_scale"] + m["col_offset"]
    row = (m["row_num"] @ mono) / (m["row_den"] @ mono) * m["row_scale"] + m["row_offset"]
    return col, row


def rmse_row_col(m, input_locs, target):
    col, row = project(m, input_locs[:, 0], input_locs[:, 1], input_locs[:, 2])
    mse = np.mean((np.stack([col, row], 1) - target) ** 2, axis=0)
    return np.sqrt(np.mean(mse))


def weighted_lsq(target, input_locs, h=1e-3, tol=1e-2, max_iter=20):
    """ref:bundle_adjust/ba_rpcfit.py:88-153.  Returns (model dict, iterations of the re-weighting loop that ran)."""
    m = {}
    m["row_scale"], m["row_offset"] = scaling_params(target[:, 1]); m["col_scale"], m["col_offset"] = scaling_params(target[:, 0])
    m["lat_scale"], m["lat_offset"] = scaling_params(input_locs[:, 1]); m["lon_scale"], m["lon_offset"] = scaling_params(input_locs[:, 0])
    m["alt_scale"], m["alt_offset"] = scaling_params(input_locs[:, 2])
    reg = (h ** 2) * np.eye(39)
    C = ((target[:, 0] - m["col_offset"]) / m["col_scale"])[:, None]; R = ((target[:, 1] - m["row_offset"]) / m["row_scale"])[:, None]
    lon = (input_locs[:, 0] - m["lon_offset"]) / m["lon_scale"]; lat = (input_locs[:, 1] - m["lat_offset"]) / m["lat_scale"]
    alt = (input_locs[:, 2] - m["alt_offset"]) / m["alt_scale"]
    pv = poly_vect(x=lat, y=lon, z=alt).T
    one = np.ones((lon.shape[0], 1))
    MC = np.hstack([one, pv, -C * pv]); MR = np.hstack([one, pv, -R * pv])
    JR = np.linalg.inv(MR.T @ MR) @ (MR.T @ R); JC = np.linalg.inv(MC.T @ MC) @ (MC.T @ C)

    def update(JR, JC):
        coefs = np.vstack([JR[:20], 1, JR[20:], JC[:20], 1, JC[20:]]).reshape(-1)
        m.update(row_num=coefs[:20], row_den=coefs[20:40], col_num=coefs[40:60], col_den=coefs[60:])
        return coefs
    coefs = update(JR, JC)
    rmse = rmse_row_col(m, input_locs, target)
    n_iter = 0
    for n_iter in range(1, max_iter + 1):
        WR2 = np.diagflat(1 / ((MR[:, :20] @ coefs[20:40]) ** 2))
        JR = np.linalg.inv((MR.T @ WR2 @ MR) + reg) @ (MR.T @ WR2 @ R)
        WC2 = np.diagflat(1 / ((MC[:, :20] @ coefs[60:80]) ** 2))
        JC = np.linalg.inv((MC.T @ WC2 @ MC) + reg) @ (MC.T @ WC2 @ C)
        coefs = update(JR, JC)
        prev, rmse = rmse, rmse_row_col(m, input_locs, target)
        if np.abs(prev - rmse) < tol:
            break
    return m, n_iter


def check_errors(m, input_locs, target):
    col, row = project(m, input_locs[:, 0], input_locs[:, 1], input_locs[:, 2])
    return np.linalg.norm(np.stack([col, row], 1) - target, axis=1)
