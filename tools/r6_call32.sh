#!/bin/bash
# direction of the slice walk per kernel (SATBA_SLICE_REV bit mask: 1 k_linearize, 2 k_backsub, 4 k_jvp, 8 k_residual), C4
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6y; mkdir -p $out
export SATBA_LIB=$GRAFT_REPO_ROOT/sat-bundleadjust_amd/satba/lib/var/libsatba_rev.so
run() { python3 bench.py --shape $1 --loss $2 --steps 200 --cpu-sample-pts 0 --no-e2e 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['ms_per_launch'],4), d['final_cost'])"; }
{
for r in 1 2; do for m in 0 1 3 5 2 9 6; do
  export SATBA_SLICE_REV=$m
  echo "== rev=$m C4 linear: $(run C4 linear)"
done; done
for m in 0 3; do export SATBA_SLICE_REV=$m; echo "== rev=$m C4 soft_l1: $(run C4 soft_l1)"; echo "== rev=$m C3 linear: $(run C3 linear)"; done
} 2>&1 | grep -v amdgpu.ids | tee $out/rev.txt
