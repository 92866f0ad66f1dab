#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6e; mkdir -p $out
bash tools/chol/run_stamps.sh r6e
( cd tools/chol; for i in 1 2 3; do for b in chol_bench_r5 chol_bench_s1 chol_bench; do echo "== $b"; timeout 120 ./$b 10 250 500 1000 2>&1 | grep "driver\|FAIL\|fault"; done; done ) | cut -c1-170 > $out/chol_ab.txt
cat $out/chol_ab.txt | cut -c1-40,100-170
grep -A40 "step: start" $out/chol_harness.txt | sed -n 1,45p | cut -c1-200
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "dense or chol or beside or factoris or two_ranks or solve_lm or device_resident" 2>&1 | tail -3
bash tools/gpu.sh r6e bench --steps 200 --warmup 20 --no-e2e --cpu-sample-pts 0
for s in C3 P3 C5; do timeout 300 python bench.py --shape $s --cpu-sample-pts 0 --no-e2e > $out/bench_$s.json 2>> $out/bench.err; done
python3 -c "
import json,glob
for f in sorted(glob.glob('$out/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['value'],1), 'it/s', d.get('kernel_ms'))"
