#!/bin/bash
# diagonal pass + pair kernel in one launch (k_schur_both; fronts with the factorisation behind them) against two launches (SATBA_SCHUR_ONE_LAUNCH=0), same library
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6t; mkdir -p $out
export SATBA_LIB=$GRAFT_REPO_ROOT/sat-bundleadjust_amd/satba/lib/var/libsatba_both.so
run() { python3 bench.py --shape $1 --loss $2 --steps 200 --cpu-sample-pts 0 --no-e2e 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), d['final_cost'], round(d['kernel_ms']['schur'],4))"; }
{
for r in 1 2; do for v in 0 1; do
  export SATBA_SCHUR_ONE_LAUNCH=$v
  for sh in C2 C3 P3 C5; do echo "== one_launch=$v $sh linear: $(run $sh linear)"; done
  echo "== one_launch=$v C5 soft_l1: $(run C5 soft_l1)"
  echo "== one_launch=$v C4 linear: $(run C4 linear)"
done; done
unset SATBA_SCHUR_ONE_LAUNCH
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed" | tail -3
SATBA_CHOL_BESIDE=0 python3 bench.py --steps 100 --cpu-sample-pts 0 --no-e2e 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 sequential front, one launch:', round(d['value'],1), round(d['ms_per_step'],4))"
SATBA_SCHUR_ONE_LAUNCH=0 SATBA_CHOL_BESIDE=0 python3 bench.py --steps 100 --cpu-sample-pts 0 --no-e2e 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 sequential front, two launches:', round(d['value'],1), round(d['ms_per_step'],4))"
} 2>&1 | grep -v amdgpu.ids | tee $out/both.txt
