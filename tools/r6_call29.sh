#!/bin/bash
# k_lin_scales' work in the previous tick's accept launch (k_lm_accept_scales) against a launch of its own (SATBA_SCALES_IN_ACCEPT=0), same library
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6v; mkdir -p $out
export SATBA_LIB=$GRAFT_REPO_ROOT/sat-bundleadjust_amd/satba/lib/var/libsatba_acs.so
run() { python3 bench.py --shape $1 --loss $2 --steps 200 --cpu-sample-pts 0 --no-e2e 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), d['final_cost'], d['solve_shipped_tolerances'])"; }
{
for r in 1 2; do for v in 0 1; do
  export SATBA_SCALES_IN_ACCEPT=$v
  for sh in C2 C3 C4 C5; do echo "== scales_in_accept=$v $sh linear: $(run $sh linear)"; done
  echo "== scales_in_accept=$v C3 soft_l1: $(run C3 soft_l1)"
done; done
unset SATBA_SCALES_IN_ACCEPT
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed" | tail -3
timeout 900 python tools/fuzz_solve.py 60 500 2>&1 | tail -1
} 2>&1 | grep -v amdgpu.ids | tee $out/acs.txt
