"""One process, a one-rank group (argv[1]: gloo or nccl): the Schur exchange in messages with real collectives against the sequential front.
nccl (RCCL): bit-identical; gloo on device tensors synchronises the device and runs into the factorisation's arrival time-out when forced to overlap
(round 6: why TorchComm.solve_in_messages sums every message first with host-blocking backends)."""
import os, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch, torch.distributed as dist
backend = sys.argv[1]
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SATBA_PIPELINE_TIMEOUT_MS="3000")
torch.cuda.set_device(0)
if backend == "nccl": dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
else: dist.init_process_group("gloo", rank=0, world_size=1)
from satba import synth as sy, trf as tr
from satba.engine_hip import HipEngine as Eng
p = sy.make_params(sy.make_scene("affine", 30, 4000, 6, seed=11), {"correction_params": ["R", "T"], "n_cam_fix": 1})
comm = tr.TorchComm(always=True)
eng = Eng(p); eng.use_torch_stream(); eng.configure("linear", 1.0)
out = {}
for mode in ("sequential", "messages"):
    comm.pipeline, comm.pipeline_min = mode == "messages", 0
    eng.linearize(); comm.allreduce(eng, eng.len_lin); eng.prepare(True); comm.allreduce(eng, eng.hdr)
    eng.schur_auto(-1.0, 0.0)
    if mode == "sequential":
        comm.allreduce_schur(eng); eng.solve()
    else:
        assert comm.solve_in_messages(eng)
    torch.cuda.synchronize()
    out[mode] = (eng.get_vector("gn_h")[: eng.n_c].copy(), eng.read_header().copy())
d = np.abs(out["sequential"][0] - out["messages"][0])
print(backend, "world 1: max |d dc|", d.max(), "of", np.abs(out["sequential"][0]).max(), "fail", out["messages"][1][tr.CHOL_FAIL], flush=True)
eng.close()
if len(sys.argv) > 2:  # timing at a benchmark shape (argv[2]: C3, C4, ...): ms per front, HIP events, both exchanges -- zero network latency (one rank)
    model, corr, n_cam, n_pts, opp = sy.CONFIGS[sys.argv[2]]
    p = sy.make_params(sy.make_scene(model, n_cam, n_pts, opp, seed=1, sigma_theta=1e-4), {"correction_params": corr, "n_cam_fix": 1})
    eng = Eng(p); eng.use_torch_stream(); eng.configure("linear", 1.0)
    for mode in ("sequential", "messages", "sequential", "messages"):
        comm.pipeline, comm.pipeline_min = mode == "messages", 0
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        t_front, t_solve = 0.0, 0.0
        reps = 30
        for it in range(reps + 3):
            ev[0].record()
            eng.linearize(); comm.allreduce(eng, eng.len_lin); eng.prepare(it == 0); comm.allreduce(eng, eng.hdr)
            eng.schur_auto(-1.0, 1e-14)
            ev[1].record()
            if not comm.solve_in_messages(eng):
                comm.allreduce_schur(eng); eng.solve()
            ev[2].record()
            torch.cuda.synchronize()
            if it >= 3:
                t_front += ev[0].elapsed_time(ev[2]); t_solve += ev[1].elapsed_time(ev[2])
        print(sys.argv[2], mode, "ms per front %.4f, of it exchange + solve phase %.4f" % (t_front / reps, t_solve / reps), "fail", eng.read_header()[tr.CHOL_FAIL], flush=True)
    eng.close()
dist.destroy_process_group()
