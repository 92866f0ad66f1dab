#!/bin/bash
# slots of the ELL stream in flight in k_residual: 2 (shipped) / 3 / 4
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6p; mkdir -p $out
V=$GRAFT_REPO_ROOT/sat-bundleadjust_amd/satba/lib/var
run() { python3 bench.py --shape $1 --loss $2 --steps 200 --cpu-sample-pts 0 --no-e2e 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"; }
{
for r in 1 2; do for lib in base pfr3 pfr4; do
  if [ $lib != base ]; then export SATBA_LIB=$V/libsatba_$lib.so; else unset SATBA_LIB; fi
  echo "== $lib C4 linear: $(run C4 linear)"
  echo "== $lib C4 soft_l1: $(run C4 soft_l1)"
  echo "== $lib C3 linear: $(run C3 linear)"
done; done
} 2>&1 | grep -v amdgpu.ids | tee $out/pfr.txt
