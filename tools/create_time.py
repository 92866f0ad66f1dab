"""Wall time of satba_problem_create (device-side layout build included) at the headline shape, split as satba_get_info reports it."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT]
from satba import synth  # noqa: E402
from satba.engine_hip import HipEngine  # noqa: E402

out = {}
for shape in sys.argv[1:] or ["C4", "C3", "C5"]:
    model, corr, n_cam, n_pts, opp = synth.CONFIGS[shape]
    scene = synth.make_scene(model, n_cam, n_pts, opp, seed=1)
    p = synth.make_params(scene, {"correction_params": corr, "n_cam_fix": 1})
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        eng = HipEngine(p)
        ts.append(time.perf_counter() - t0)
        info = eng.info()
        eng.close()
    out[shape] = {"HipEngine_s": ts, "n_obs": int(p.n_obs), "satba_problem_create_ms": info["ms_create"],
                  "of_which_ms": {"host_to_device_copies_queued": info["ms_uploads"], "sizes_known": info["ms_sizes"],
                                  "ell_and_camera_lists": info["ms_ell"], "pair_lists": info["ms_pairs"]},
                  "pair_entries": info["pair_entries"], "ell_len": info["ell_len"]}
print(json.dumps(out))
