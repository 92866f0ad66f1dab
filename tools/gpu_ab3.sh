#!/bin/bash
# A/B of one bench line under switches with chosen kernels' times from a kernel trace: LOSS=.. KPAT=regex tools/gpu_ab3.sh TAG "ENV=.." ...
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1))
  if [ "$e" = "-" ]; then e=""; fi
  for v in $e; do export $v; done
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_$i -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample-pts 0 --steps 40 --warmup 4 --loss ${LOSS:-soft_l1} --shape ${SHAPE:-C4} > $out/bench_$i.json 2> $out/prof_$i.log
  python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $out/prof_$i/stats_results.db > $out/kernel_stats_$i.txt
  echo "[$e] $(python3 -c "import json; d=json.load(open('$out/bench_$i.json')); print(round(d['value'],1), repr(d['final_cost']))")"
  grep -E "${KPAT:-k_schur_diag<}" $out/kernel_stats_$i.txt
  for v in $e; do unset ${v%%=*}; done
done
find $out -name "*.db" -size +2M -delete
