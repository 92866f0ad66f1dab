#!/bin/bash
# the factorisation beside the pair kernel below 8 192 camera pairs (SATBA_SCHUR_MERGE=1), re-measured with round 6's chain
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6l; mkdir -p $out
run() { python3 bench.py --shape $1 --loss $2 --steps 200 --cpu-sample-pts 0 --no-e2e 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), 'beside', d.get('chol_beside'), 'timeouts', d.get('chol_beside_timeouts'), d['final_cost'])"; }
{
for sh in C3 P3 C5; do for loss in linear soft_l1; do
  unset SATBA_SCHUR_MERGE SATBA_CHOL_BESIDE_WGS
  echo "== $sh $loss default: $(run $sh $loss)"
  for w in 8 16 32; do
    export SATBA_SCHUR_MERGE=1 SATBA_CHOL_BESIDE_WGS=$w
    echo "== $sh $loss merge=1 wgs=$w: $(run $sh $loss)"
  done
  unset SATBA_SCHUR_MERGE SATBA_CHOL_BESIDE_WGS
  echo "== $sh $loss default: $(run $sh $loss)"
done; done
} 2>&1 | grep -v amdgpu.ids | tee $out/beside_small.txt
