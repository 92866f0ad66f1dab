#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6h; mkdir -p $out
( cd tools/chol; echo "== mirror / hand-over hunt: 60 processes x sizes 65 66 67 70 96 129 130 192 200, 20 repetitions each, every solve checked"
  for i in $(seq 1 60); do SATBA_CHECK_ALL=1 timeout 60 ./chol_bench 20 65 66 67 70 96 129 130 192 200 2>&1 | grep "mirror:\|rep \|FAILED\|all ok" | cut -c1-160; done | sort | uniq -c | sort -rn | head ) | tee $out/chol_hunt.txt
bash tools/chol/run_stamps.sh r6h
( cd tools/chol; for i in 1 2 3; do for b in chol_bench_r5 chol_bench; do echo "== $b"; timeout 120 ./$b 10 250 500 1000 2>&1 | grep "driver\|FAIL\|fault"; done; done ) | cut -c1-170 > $out/chol_ab.txt
cat $out/chol_ab.txt | cut -c1-40,100-170
grep -A40 "step: start" $out/chol_harness.txt | sed -n 1,12p | cut -c1-200
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "dense or chol or beside or factoris or two_ranks or solve_lm or device_resident" 2>&1 | tail -3
bash tools/gpu.sh r6h bench --steps 200 --warmup 20 --no-e2e --cpu-sample-pts 0
