#!/bin/bash
# the trial kernel with / without the call of the second decision compiled in (120 VGPRs + scratch against 56): base fused, base unfused, nocall unfused
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6q; mkdir -p $out
V=$GRAFT_REPO_ROOT/sat-bundleadjust_amd/satba/lib/var
run() { python3 bench.py --shape $1 --loss linear --steps 200 --cpu-sample-pts 0 --no-e2e 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4))"; }
{
for r in 1 2 3; do
  for sh in C4 C3 C2; do
    unset SATBA_LIB; export SATBA_DECIDE2_FUSED=1; echo "== $sh fused (call compiled in): $(run $sh)"
    export SATBA_DECIDE2_FUSED=0; echo "== $sh launch of its own (call compiled in): $(run $sh)"
    export SATBA_LIB=$V/libsatba_nocall.so; echo "== $sh launch of its own (call compiled out): $(run $sh)"
  done
done
} 2>&1 | grep -v amdgpu.ids | tee $out/nocall.txt
