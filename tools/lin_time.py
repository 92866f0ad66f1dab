"""
Times the fused residual+Jacobian kernel (k_linearize) of one or more builds of libsatba_hip.so on the headline shape.

    python tools/lin_time.py [--shape C4] [--loss linear] lib1.so lib2.so ...

Each library is loaded in a child process (SATBA_LIB) so that ablation builds (-DSATBA_ABLATE_*, see the Makefile's
EXTRA variable) can be compared in one GPU call.  Prints ms per launch (HIP events on the launch stream, 20 launches).
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for path in (os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT):
    if path not in sys.path:
        sys.path.insert(0, path)


def child(shape, loss, kernels):
    import torch
    from satba import sharding, synth
    from satba.engine_hip import HipEngine
    torch.cuda.set_device(0)
    model, corr, n_cam, n_pts, opp = synth.CONFIGS[shape]
    scene = synth.make_scene(model, n_cam, n_pts, opp, seed=1, sigma_theta=1e-4 if model != "rpc" else 1e-6)
    p = synth.make_params(scene, {"correction_params": corr, "n_cam_fix": 1})
    eng = HipEngine(p, sharding.make_shard(p, 0, 1))
    eng.configure(loss, 1.0)
    eng.set_x(p.params_opt.copy())
    eng.linearize(); eng.prepare(True); eng.schur(1e-6)
    out = []
    for k in kernels:
        eng.linearize(); eng.prepare(False); eng.schur(1e-6)
        eng.time_kernel(k, 5)  # clocks up
        eng.linearize(); eng.prepare(False); eng.schur(1e-6)
        out.append("{} {:.4f}".format(k, eng.time_kernel(k, 20)))
    print(os.environ.get("SATBA_LIB", "default"), " ".join(out), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="C4")
    ap.add_argument("--loss", default="linear")
    ap.add_argument("--kernels", default="linearize")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("libs", nargs="*")
    args = ap.parse_args()
    if args.child:
        child(args.shape, args.loss, args.kernels.split(","))
    else:
        for lib in args.libs or [""]:
            env = dict(os.environ)
            if lib:
                env["SATBA_LIB"] = os.path.abspath(lib)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--shape", args.shape, "--loss", args.loss,
                            "--kernels", args.kernels], env=env, check=False)
