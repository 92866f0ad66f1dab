#!/bin/bash
# concurrent front (C4): diagonal pass + pair kernel in one launch, the last diagonal workgroup finishing the camera part (SATBA_SCHUR_ONE_LAUNCH_BESIDE=0: three launches)
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r6x; mkdir -p $out
export SATBA_LIB=$GRAFT_REPO_ROOT/sat-bundleadjust_amd/satba/lib/var/libsatba_bb.so
run() { python3 bench.py --shape $1 --loss $2 --steps 200 --cpu-sample-pts 0 --no-e2e 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), d['final_cost'], d.get('chol_beside'), d.get('chol_beside_timeouts'))"; }
{
for r in 1 2 3; do for v in 0 1; do
  export SATBA_SCHUR_ONE_LAUNCH_BESIDE=$v
  echo "== one_launch_beside=$v C4 linear: $(run C4 linear)"
done; done
unset SATBA_SCHUR_ONE_LAUNCH_BESIDE
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed" | tail -3
timeout 900 python tools/fuzz_beside.py 100 300 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample-pts 0 --steps 100 --no-e2e > $out/bench_profiled.json 2> $out/prof.log
cd $GRAFT_REPO_ROOT
python3 tools/rocpd_stats.py $out/prof/stats_results.db > $out/kernel_stats.txt
python3 tools/rocpd_timeline.py $out/prof/stats_results.db > $out/timeline.txt 2>&1
head -12 $out/kernel_stats.txt | cut -c1-120; cat $out/timeline.txt
find $out -name "*.db" -size +2M -delete
} 2>&1 | grep -v amdgpu.ids | tee $out/bb.txt
