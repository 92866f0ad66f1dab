"""Where the time of the dense Cholesky goes: wall-clock stamps taken inside the kernels (satba_debug_chol_times), in the
factorisation mode selected by SATBA_CHOL (0: two panels per launch, 2: single steps)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT]
import numpy as np
from satba import synth
from satba.engine_hip import HipEngine

TS = 24
mode = int(os.environ.get("SATBA_CHOL", "0"))
scene = synth.make_affine_scene(200, 20000, 10, seed=1)
p = synth.make_params(scene, {"correction_params": ["R", "T"], "n_cam_fix": 1})
eng = HipEngine(p)
eng.configure("linear", 1.0)
eng.lib.satba_debug_chol_times.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_int32)]
buf = np.zeros(TS * 256, dtype=np.int64)
n = C.c_int32()
for _ in range(3):
    eng.linearize(); eng.prepare(True); eng.schur(1e-3)
    eng.lib.satba_debug_chol_times(eng._h, buf.ctypes.data_as(C.POINTER(C.c_longlong)), C.byref(n))
t = buf.reshape(-1, TS).astype(np.float64) * 1e-2  # us
launches = int((t[:, 0] > 0).sum())
t = t[:launches]
t0 = t[0, 0]
print("mode", mode, "launches", launches, "span %.1f us" % (t.max() - t0))
if mode == 2:
    print("step  start  | tile(0,0): update  potrf  rest | tile(1,0): start-lag update wait-flag trsm | gap to next step")
    for k in range(launches):
        a = t[k]
        nxt = t[k + 1, 0] - max(a[3], a[7]) if k + 1 < launches else 0.0
        print("%3d %8.1f | %6.1f %6.1f %6.1f | %6.1f %6.1f %6.1f %6.1f | %6.1f" % (
            k, a[0] - t0, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[0], a[5] - a[4], a[6] - a[5], a[7] - a[6], nxt))
else:
    # per launch: times relative to the start of row tile 0; '-' where a stamp was not taken
    print("launch start | tile 0: upd D0 D1 | tile 1: start upd X0 X1 | tile 2: start upd X0 X1 | next launch")
    def f(v, ref):
        return "%5.1f" % (v - ref) if v > 0 else "    -"
    for k in range(launches):
        a = t[k]
        ref = a[0]
        nxt = f(t[k + 1, 0], ref) if k + 1 < launches else "    -"
        print("%3d %8.1f | %s %s %s | %s %s %s %s | %s %s %s %s | %s" % (
            k, a[0] - t0, f(a[1], ref), f(a[2], ref), f(a[3], ref), f(a[8], ref), f(a[9], ref), f(a[10], ref), f(a[11], ref),
            f(a[16], ref), f(a[17], ref), f(a[18], ref), f(a[19], ref), nxt))
