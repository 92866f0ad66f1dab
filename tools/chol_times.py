"""Where the time of one Cholesky panel step (k_chol_step) goes: wall-clock stamps taken inside the kernel."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT]
import numpy as np
from satba import synth
from satba.engine_hip import HipEngine

scene = synth.make_affine_scene(200, 20000, 10, seed=1)
p = synth.make_params(scene, {"correction_params": ["R", "T"], "n_cam_fix": 1})
eng = HipEngine(p)
eng.configure("linear", 1.0)
eng.lib.satba_debug_chol_times.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_int32)]
buf = np.zeros(8 * 256, dtype=np.int64)
n = C.c_int32()
for _ in range(3):
    eng.linearize(); eng.prepare(True); eng.schur(1e-3)
    eng.lib.satba_debug_chol_times(eng._h, buf.ctypes.data_as(C.POINTER(C.c_longlong)), C.byref(n))
t = buf.reshape(-1, 8)[: n.value].astype(np.float64) * 1e-2  # us
t0 = t[0, 0]
print("steps", n.value, "span %.1f us" % (t[:, 3].max() - t0))
print("step  start  | tile(0,0): update  potrf  rest | tile(1,0): start-lag update wait-flag trsm | gap to next step")
for k in range(n.value):
    a = t[k]
    nxt = t[k + 1, 0] - max(a[3], a[7]) if k + 1 < n.value else 0.0
    print("%3d %8.1f | %6.1f %6.1f %6.1f | %6.1f %6.1f %6.1f %6.1f | %6.1f" % (
        k, a[0] - t0, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[0], a[5] - a[4], a[6] - a[5], a[7] - a[6], nxt))
