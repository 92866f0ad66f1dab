#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6d; mkdir -p $out
bash tools/gpu.sh r6d tests
grep -E "^FAILED|^ERROR" $out/pytest_gpu.log | head
bash tools/gpu.sh r6d bench --steps 200 --warmup 20
( for cfg in "1 4" "0 4" "1 2" "1 8" "1 1"; do set -- $cfg
    echo "== overlap $1 lanes $2"; SATBA_ERR_OVERLAP=$1 SATBA_COPY_LANES=$2 timeout 300 python tools/e2e_time.py C4
  done; echo "== C3"; timeout 300 python tools/e2e_time.py C3 ) 2>&1 | grep -v amdgpu.ids > $out/e2e.txt
python3 - <<'PY'
import json
for line in open("gpurun_out/r6d/e2e.txt"):
    if line.startswith("=="): print(line.strip()); continue
    try: d=json.loads(line)
    except Exception: print(line[:200]); continue
    for c in d["calls"][1:]:
        print("  ", c["call"], {k: round(v*1e3,2) for k,v in c.items() if k.endswith("_s")})
PY
