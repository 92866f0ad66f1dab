#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6z; mkdir -p $out
V=$GRAFT_REPO_ROOT/sat-bundleadjust_amd/satba/lib/var
run() { python3 bench.py --shape C4 --loss linear --steps 200 --cpu-sample-pts 0 --no-e2e 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['ms_per_launch'],4), d['final_cost'])"; }
{
for r in 1 2 3; do for lib in base nt; do
  if [ $lib != base ]; then export SATBA_LIB=$V/libsatba_$lib.so; else unset SATBA_LIB; fi
  echo "== $lib C4: $(run)"
done; done
} 2>&1 | grep -v amdgpu.ids | tee $out/nt.txt
