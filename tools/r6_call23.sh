#!/bin/bash
# RPC tables in the LDS as aligned 16-byte slots (RPCS 91 -> 94) against the build before
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6o; mkdir -p $out
V=$GRAFT_REPO_ROOT/sat-bundleadjust_amd/satba/lib/var/libsatba_rpc94.so
run() { python3 bench.py --shape $1 --loss $2 --steps 200 --cpu-sample-pts 0 --no-e2e 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), d['final_cost'], {k: round(v,4) for k,v in d['kernel_ms'].items()})"; }
{
for r in 1 2; do for lib in base rpc94; do
  if [ $lib != base ]; then export SATBA_LIB=$V; else unset SATBA_LIB; fi
  echo "== $lib C5 linear: $(run C5 linear)"
  echo "== $lib C5 soft_l1: $(run C5 soft_l1)"
done; done
export SATBA_LIB=$V
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -k "rpc or RPC or C5 or golden or pipeline or triang" 2>&1 | grep -E "passed|failed" | tail -3
} 2>&1 | grep -v amdgpu.ids | tee $out/rpc94.txt
