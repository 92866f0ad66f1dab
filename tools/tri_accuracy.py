"""Runs ON the GPU box: how far the device's, the oracle's and LAPACK's linear triangulations are from the exact null vector of the
DLT matrix (one-sided Jacobi in numpy.longdouble), in metres.   usage: python tools/tri_accuracy.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT]
from oracle import triangulate_oracle as T  # noqa: E402
from satba import ft_triangulate as FT  # noqa: E402
from satba import synth  # noqa: E402


def exact_null(P1, P2, a, b):
    L = np.longdouble
    P1, P2 = P1.astype(L), P2.astype(L)
    out = np.zeros((len(a), 3))
    for n in range(len(a)):
        A = np.stack([L(a[n, 0]) * P1[2] - P1[0], L(a[n, 1]) * P1[2] - P1[1], L(b[n, 0]) * P2[2] - P2[0], L(b[n, 1]) * P2[2] - P2[1]])
        V = np.eye(4, dtype=L)
        for _ in range(60):
            rot = False
            for p in range(3):
                for q in range(p + 1, 4):
                    al, be, ga = A[:, p] @ A[:, p], A[:, q] @ A[:, q], A[:, p] @ A[:, q]
                    if ga == 0 or abs(ga) <= L(1e-19) * np.sqrt(al * be):
                        continue
                    rot = True
                    z = (be - al) / (2 * ga)
                    t = np.sign(z) / (abs(z) + np.sqrt(1 + z * z)) if z != 0 else L(1)
                    c = 1 / np.sqrt(1 + t * t); s = c * t
                    A[:, [p, q]] = np.stack([c * A[:, p] - s * A[:, q], s * A[:, p] + c * A[:, q]], 1)
                    V[:, [p, q]] = np.stack([c * V[:, p] - s * V[:, q], s * V[:, p] + c * V[:, q]], 1)
            if not rot:
                break
        k = int(np.argmin((A * A).sum(0)))
        out[n] = (V[:3, k] / V[3, k]).astype(np.float64)
    return out


for model in ("affine", "perspective"):
    scene = synth.make_scene(model, 4, 1500, 4, seed=9)
    C = scene.to_dense_C()
    t = np.where(~np.isnan(C[0]) & ~np.isnan(C[2]))[0][:300]
    oi, oj = C[0:2, t].T, C[2:4, t].T
    P1, P2 = np.asarray(scene.cameras[0], float), np.asarray(scene.cameras[1], float)
    ex = exact_null(P1, P2, oi, oj)
    dev = FT.linear_triangulation_multiple_pts(P1, P2, oi, oj)
    orc = T.linear_triangulation_multiple_pts(P1, P2, oi, oj)
    A = np.stack([oi[:, 0:1] * P1[2] - P1[0], oi[:, 1:2] * P1[2] - P1[1], oj[:, 0:1] * P2[2] - P2[0], oj[:, 1:2] * P2[2] - P2[1]], axis=1)
    vt = np.linalg.svd(A)[2][:, 3, :]
    lap = vt[:, :3] / vt[:, 3:4]
    e = lambda x, y: np.linalg.norm(x - y, axis=1)
    print(model, "n", len(t), "| device - exact: median %.3g max %.3g m | oracle (Jacobi) - exact: max %.3g m | numpy.linalg.svd - exact: median %.3g max %.3g m"
          % (np.median(e(dev, ex)), e(dev, ex).max(), e(orc, ex).max(), np.median(e(lap, ex)), e(lap, ex).max()))
