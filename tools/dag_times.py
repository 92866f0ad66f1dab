"""Per-task timing of the dataflow Cholesky (SATBA_DAG_TIMES=1): where the critical path goes."""
import ctypes as C
import os
import sys

os.environ["SATBA_DAG_TIMES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT]
import numpy as np
from satba import synth
from satba.engine_hip import HipEngine

scene = synth.make_affine_scene(200, 20000, 10, seed=1)
p = synth.make_params(scene, {"correction_params": ["R", "T"], "n_cam_fix": 1})
eng = HipEngine(p)
eng.configure("linear", 1.0)
for _ in range(2):
    eng.linearize(); eng.prepare(True); eng.schur(1e-3); eng.solve(); eng.read_header()
n = C.c_int32()
eng.lib.satba_debug_dag_times.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_int32)]
eng.lib.satba_debug_dag_times(eng._h, None, C.byref(n))
buf = np.zeros(4 * n.value, dtype=np.int64)
eng.lib.satba_debug_dag_times(eng._h, buf.ctypes.data_as(C.POINTER(C.c_longlong)), C.byref(n))
t = buf.reshape(-1, 4)
t0 = t[:, 0].min()
typ = (t[:, 3] >> 48) & 0xffff; ti = (t[:, 3] >> 32) & 0xffff; tj = (t[:, 3] >> 16) & 0xffff; tk = t[:, 3] & 0xffff
names = ["F", "T", "U", "Tb", "Ub"]
tick = 1e-8 * 1e6  # s_memtime / readcyclecounter ticks at 100 MHz -> microseconds
print("total span %.1f us" % ((t[:, 2].max() - t0) * tick))
for k in range(5):
    m = typ == k
    print("%-3s n=%4d  exec avg %.2f us  max %.2f  wait avg %.2f us" % (names[k], m.sum(), ((t[m, 2] - t[m, 1]).mean()) * tick,
          ((t[m, 2] - t[m, 1]).max()) * tick, ((t[m, 1] - t[m, 0]).mean()) * tick))
print("critical chain (k: F ready, F done, T(k+1,k) ready/done, U(k+1,k+1,k) ready/done) in us:")
for k in range(0, int(tk.max()) + 1):
    f = t[(typ == 0) & (tk == k)][0]
    row = [(f[1] - t0) * tick, (f[2] - t0) * tick]
    for (ty, i, j) in ((1, k + 1, k), (2, k + 1, k + 1)):
        m = (typ == ty) & (ti == i) & (tj == j) & (tk == k)
        if m.any():
            r = t[m][0]
            row += [(r[1] - t0) * tick, (r[2] - t0) * tick]
    print(k, " ".join("%8.1f" % v for v in row))
