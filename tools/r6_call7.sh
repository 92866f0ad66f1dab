#!/bin/bash
cd $GRAFT_REPO_ROOT/tools/chol
out=$GRAFT_REPO_ROOT/gpurun_out/r6g; mkdir -p $out
for b in chol_bench_yDIAG chol_bench_yRIDER chol_bench_yDNEXT chol_bench_yINV chol_bench_yAUX; do
  echo "== $b" | tee -a $out/mirror_race3.txt
  for i in $(seq 1 25); do timeout 60 ./$b 3 65 66 2>&1 | grep "FAILED\|all ok" | cut -c1-200; done | sort | uniq -c | sort -rn | head -8 | tee -a $out/mirror_race3.txt
done
