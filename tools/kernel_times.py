"""
Per-kernel times (HIP events, satba_time_kernel) of one configuration, one JSON line.  Environment switches that the
library reads (SATBA_BPC, SATBA_SCHUR_CHUNKS, ...) are taken from the caller's environment: run once per variant.
    python tools/kernel_times.py [shape] [loss] [reps]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT]
from satba import synth  # noqa: E402
from satba.engine_hip import HipEngine  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "C4"
loss = sys.argv[2] if len(sys.argv) > 2 else "linear"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
model, corr, n_cam, n_pts, opp = synth.CONFIGS[shape]
scene = synth.make_scene(model, n_cam, n_pts, opp, seed=1, sigma_theta=1e-4 if model != "rpc" else 1e-6)
p = synth.make_params(scene, {"correction_params": corr, "n_cam_fix": 1})
t0 = time.perf_counter()
eng = HipEngine(p)
t_create = time.perf_counter() - t0
eng.configure(loss, 1.0)
out = {"shape": shape, "loss": loss, "env": {k: v for k, v in os.environ.items() if k.startswith("SATBA_")}, "create_s": t_create,
       "info": eng.info()}
if "--loop" in sys.argv:
    # LM iterations as a solve runs them (satba_lm_step: k_linearize with the point part of the prepare phase fused in) -- for the PMC
    # passes that measure the traffic of the kernel as it runs inside the loop
    eng.snapshot_x(False)
    for i in range(reps):
        if i % 3 == 0:
            eng.snapshot_x(True)
        eng.lm_step(i % 3 == 0, 1.0, 1e-14)
    print(json.dumps({"loop_iterations": reps}))
    sys.exit(0)
names = ("linearize", "residual", "jvp", "backsub", "schur", "cholesky")
if "--only" in sys.argv:  # (an ablation build's sums are wrong: only the kernel in question is run)
    names = (sys.argv[sys.argv.index("--only") + 1],)
for name in names:
    eng.linearize()
    if len(names) > 1:
        eng.prepare(False); eng.schur(1e-6)
    out[name] = round(eng.time_kernel(name, reps if name not in ("schur", "cholesky") else max(2, reps // 4)), 5)
print(json.dumps({k: out[k] for k in names}), json.dumps(out["env"]), flush=True)
if "-v" in sys.argv:
    print(json.dumps(out))
