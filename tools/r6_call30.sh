#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6w; mkdir -p $out
run() { python3 bench.py --shape $1 --loss linear --steps 200 --cpu-sample-pts 0 --no-e2e 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()}, d.get('launch_patterns_executed'), d['solve_shipped_tolerances']['nfev'])"; }
{
for i in 1 2 3 4 5 6 7 8; do echo "C3: $(run C3)"; done
} 2>&1 | grep -v amdgpu.ids | tee $out/bimodal.txt
