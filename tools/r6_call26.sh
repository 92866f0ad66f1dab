#!/bin/bash
# unit-weight pair kernel at two waves per SIMD against three
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6r; mkdir -p $out
V=$GRAFT_REPO_ROOT/sat-bundleadjust_amd/satba/lib/var
run() { python3 bench.py --shape $1 --loss linear --steps 200 --cpu-sample-pts 0 --no-e2e 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"; }
{
for r in 1 2 3; do for lib in base occ2; do
  if [ $lib != base ]; then export SATBA_LIB=$V/libsatba_$lib.so; else unset SATBA_LIB; fi
  echo "== $lib C4: $(run C4)"
  echo "== $lib C3: $(run C3)"
done; done
} 2>&1 | grep -v amdgpu.ids | tee $out/occ2.txt
