#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6c; mkdir -p $out
V=$GRAFT_REPO_ROOT/sat-bundleadjust_amd/satba/lib/var
# 1. dense solve: correctness of the new chain at every size, then timings of the variants in alternation
bash tools/chol/run_stamps.sh r6c
( cd tools/chol; for i in 1 2 3; do for b in chol_bench_r5 chol_bench chol_bench_pHANDOVER chol_bench_pINV chol_bench_pDNEXT chol_bench_pALL; do echo "== $b"; timeout 120 ./$b 10 250 1000 2>&1 | grep "driver\|FAIL\|fault"; done; done ) | cut -c1-170 > $out/chol_ab.txt
cat $out/chol_ab.txt | cut -c1-40,100-170
( cd tools/chol; SATBA_STAMPS=1 timeout 120 ./chol_bench_pALL_st 3 1000 ) > $out/chol_stamps_pALL.txt 2>&1
grep -A40 "step: start" $out/chol_harness.txt | sed -n 1,45p | cut -c1-200
echo "---- pALL"; grep -A40 "step: start" $out/chol_stamps_pALL.txt | sed -n 1,45p | cut -c1-200
# 2. bench: is it the copy lanes' streams?
for i in 1 2; do for cfg in lanes direct; do
  if [ $cfg = direct ]; then export SATBA_COPY_DIRECT=1; else unset SATBA_COPY_DIRECT; fi
  timeout 600 python bench.py --steps 200 --warmup 20 --no-e2e --cpu-sample-pts 0 > $out/bench_${cfg}_$i.json 2>> $out/bench.err
  python3 -c "
import json,sys
d=json.loads(open('$out/bench_${cfg}_$i.json').read().strip().splitlines()[-1]); print('$cfg $i', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms; in-loop linearize', round(d['roofline']['ms_per_launch'],4), 'beside', d.get('chol_beside'), d.get('chol_beside_timeouts'))"
done; done
unset SATBA_COPY_DIRECT
# 3. e2e
( for cfg in "1 4" "0 4" "1 2" "0 8" "1 1"; do set -- $cfg
    echo "== overlap $1 lanes $2"; SATBA_ERR_OVERLAP=$1 SATBA_COPY_LANES=$2 timeout 300 python tools/e2e_time.py C4
  done ) 2>&1 | grep -v amdgpu.ids > $out/e2e.txt
python3 - <<'PY'
import json
for line in open("gpurun_out/r6c/e2e.txt"):
    if line.startswith("=="): print(line.strip()); continue
    try: d=json.loads(line)
    except Exception: print(line[:200]); continue
    for c in d["calls"][1:]:
        print("  ", c["call"], {k: round(v*1e3,2) for k,v in c.items() if k.endswith("_s")})
PY
