import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import cases
from oracle import lm_oracle as L
from satba import ba_core, trf
from satba.engine_hip import HipEngine

_, p, g = cases.fun_case("rpc_RT")
v = ba_core._frozen_vars(g["v"][1].copy(), p)
dev, ora = HipEngine(p, rpc_f32=False), L.OracleEngine(p, rpc_f32=False)
for e in (dev, ora):
    e.configure("linear", 1.0); e.set_x(v); e.linearize(); e.prepare(True)
h = dev.read_header(); ho = ora.read_header()
print("prep dev", h[:6]); print("prep ora", ho[:6])
Delta = np.sqrt(h[3]); _, ag = trf.minimize_quadratic_1d(0.5*h[2], -h[1], 0.0, Delta/np.sqrt(h[1])); lam = -ag/Delta**2
print("lam", lam)
for e in (dev, ora): e.schur(lam)
n = dev.n_c
S = dev.get_exchange(dev.hdr, n*n).reshape(n, n).T; rhs = dev.get_exchange(dev.hdr+n*n, n)
So = ora._xb[ora.hdr:ora.hdr+n*n].reshape(n, n); rhso = ora._xb[ora.hdr+n*n:ora.len_schur]
low = np.tril_indices(n)
print("S nan", np.isnan(S).sum(), "rel S", np.abs(S[low]-So[low]).max()/np.abs(So).max(), "rhs rel", np.abs(rhs-rhso).max()/np.abs(rhso).max())
Sl = np.tril(S) + np.tril(S, -1).T
d = np.sqrt(np.diag(Sl)); Sn = Sl/np.outer(d, d)
ev = np.linalg.eigvalsh(Sn); print("scaled eig min/max", ev[0], ev[-1])
for e in (dev, ora): e.solve()
print("solve dev", dev.read_header()[:6]); print("solve ora", ora.read_header()[:6])
gn = dev.get_vector("gn_h"); print("gn nan count", np.isnan(gn).sum(), "first nan idx", np.argmax(np.isnan(gn)) if np.isnan(gn).any() else None, "n_c", n)
