#!/bin/bash
# lane = point diagonal pass (k_diag_fx) against the camera-major one (SATBA_DIAG_FX=0), same library
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6n; mkdir -p $out
export SATBA_LIB=$GRAFT_REPO_ROOT/sat-bundleadjust_amd/satba/lib/var/libsatba_dgfx.so
run() { python3 bench.py --shape $1 --loss $2 --steps 200 --cpu-sample-pts 0 --no-e2e 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), 'nfev', d['solve_shipped_tolerances']['nfev'], d['solve_shipped_tolerances']['cost'], d['final_cost'], {k: round(v,4) for k,v in d['kernel_ms'].items()})"; }
{
for r in 1 2; do for v in 0 1; do
  export SATBA_DIAG_FX=$v
  for sh in C4 C3 C2; do echo "== diag_fx=$v $sh: $(run $sh linear)"; done
done; done
unset SATBA_DIAG_FX
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -v amdgpu | tail -15
} 2>&1 | grep -v amdgpu.ids | tee $out/dgfx.txt
