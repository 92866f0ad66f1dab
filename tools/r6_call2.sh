#!/bin/bash
# round 6, second GPU call: dense-solve harness (new chain build against the round-5 binary), A/B of the bench line (round-5 library
# against this round's), k_linearize variants on one box in alternation, e2e with the copy lanes
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6b; mkdir -p $out
V=$GRAFT_REPO_ROOT/sat-bundleadjust_amd/satba/lib/var
nproc > $out/host.txt; cat /sys/fs/cgroup/cpu.max >> $out/host.txt 2>/dev/null; cat $out/host.txt
# 1. dense solve
bash tools/chol/run_stamps.sh r6b
( cd tools/chol; for i in 1 2 3; do for b in chol_bench_r5 chol_bench; do echo "== $b"; timeout 120 ./$b 10 250 500 1000 2>&1 | grep "driver\|tiles "; done; done ) | cut -c1-170 > $out/chol_ab.txt
cat $out/chol_ab.txt
grep -A40 "step: start" $out/chol_harness.txt | head -60
# 2. bench A/B
for i in 1 2; do for lib in r5 c1; do
  SATBA_LIB=$V/libsatba_$lib.so timeout 600 python bench.py --steps 200 --warmup 20 --no-e2e --cpu-sample-pts 0 > $out/bench_${lib}_$i.json 2>> $out/bench.err
  python3 -c "
import json,sys
d=json.loads(open('$out/bench_${lib}_$i.json').read().strip().splitlines()[-1]); print('$lib $i', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms; in-loop linearize', round(d['roofline']['ms_per_launch'],4), 'beside', d.get('chol_beside'), d.get('chol_beside_timeouts'))"
done; done
# 3. k_linearize variants
( for i in 1 2 3; do
    for lib in c1 noatomics pf3 pf1; do echo -n "$lib: "; SATBA_LIB=$V/libsatba_$lib.so timeout 300 python tools/kernel_times.py C4 linear 30 --only linearize; done
    echo -n "c1 camera sums off (CAMSUMS = false kernel, writes f): "; SATBA_CAM_SUMS=1 SATBA_LIB=$V/libsatba_c1.so timeout 300 python tools/kernel_times.py C4 linear 30 --only linearize
  done ) 2>&1 | grep -v amdgpu.ids > $out/linearize_variants.txt
cat $out/linearize_variants.txt
# 4. e2e
( for cfg in "1 4" "0 4" "1 2" "0 8" "1 1"; do set -- $cfg
    echo "== overlap $1 lanes $2"; SATBA_ERR_OVERLAP=$1 SATBA_COPY_LANES=$2 timeout 300 python tools/e2e_time.py C4
  done ) 2>&1 | grep -v amdgpu.ids > $out/e2e.txt
python3 - <<'PY'
import json
for line in open("gpurun_out/r6b/e2e.txt"):
    if line.startswith("=="): print(line.strip()); continue
    try: d=json.loads(line)
    except Exception: print(line[:200]); continue
    for c in d["calls"][1:]:
        print("  ", c["call"], {k: round(v*1e3,2) for k,v in c.items() if k.endswith("_s")})
PY
