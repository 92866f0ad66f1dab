"""bench.py (no CPU leg, no e2e) over the values of one environment variable, one line per run: how the launch parameters in
csrc/satba_capi.hip were chosen (ten runs fit one GPU call).  usage: sweep_env.py TAG VAR v1,v2,... [bench args...]   ("-": variable unset)"""
import json, os, subprocess, sys
tag, var, vals = sys.argv[1], sys.argv[2], sys.argv[3].split(",")
args = sys.argv[4:]
root = os.environ.get("GRAFT_REPO_ROOT", ".")
for v in vals:
    env = dict(os.environ)
    if v != "-": env[var] = v
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-sample-pts", "0", "--no-e2e"] + args, env=env, capture_output=True, text=True, timeout=600)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(tag, var, v, " ".join(args), "->", round(d["value"], 1), "it/s", round(d["ms_per_step"], 4), "ms", "beside", d.get("chol_beside"), flush=True)
    except Exception as e:
        print(tag, var, v, "failed", e, r.stderr[-300:], flush=True)
