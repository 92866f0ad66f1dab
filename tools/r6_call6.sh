#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6f; mkdir -p $out
bash tools/gpu.sh r6f tests
grep -E "^FAILED|^ERROR" $out/pytest_gpu.log | head
bash tools/chol/run_stamps.sh r6f
bash tools/gpu.sh r6f stats --steps 200 --warmup 20 --no-e2e
python tools/sweep_env.py r6f SATBA_CHOL_BESIDE_WGS -,24,32,40 --steps 200 --warmup 20 2>&1 | tee $out/sweep_wgs.txt
python tools/sweep_env.py r6f SATBA_CHOL_BESIDE_WGS -,8,16,24 --steps 200 --warmup 20 --loss soft_l1 2>&1 | tee -a $out/sweep_wgs.txt
( for cfg in "1 4" "1 8"; do set -- $cfg
    echo "== overlap $1 lanes $2"; SATBA_ERR_OVERLAP=$1 SATBA_COPY_LANES=$2 timeout 300 python tools/e2e_time.py C4
  done ) 2>&1 | grep -v amdgpu.ids > $out/e2e.txt
python3 - <<'PY'
import json
for line in open("gpurun_out/r6f/e2e.txt"):
    if line.startswith("=="): print(line.strip()); continue
    try: d=json.loads(line)
    except Exception: print(line[:200]); continue
    for c in d["calls"][1:]:
        print("  ", c["call"], {k: round(v*1e3,2) for k,v in c.items() if k.endswith("_s")})
PY
