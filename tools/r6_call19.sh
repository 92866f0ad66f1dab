#!/bin/bash
# diagonal pass on a third stream beside the pair kernel (SATBA_DIAG_BESIDE=1) against the shipped order: bench lines in alternation, bit identity, time line
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6k; mkdir -p $out
export SATBA_LIB=$GRAFT_REPO_ROOT/sat-bundleadjust_amd/satba/lib/var/libsatba_dgb.so
{
for r in 1 2 3; do for v in 0 1; do
  export SATBA_DIAG_BESIDE=$v
  echo "== diag_beside=$v bench C4"; python3 bench.py --steps 200 --cpu-sample-pts 0 --no-e2e 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('chol_beside'), d.get('chol_beside_timeouts'), d['final_cost'])"
done; done
export SATBA_DIAG_BESIDE=1
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x -k "beside or factoris or repeat_bitwise or device_resident or headline" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
timeout 900 python tools/fuzz_beside.py 30 300 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample-pts 0 --steps 100 --no-e2e > $GRAFT_REPO_ROOT/$out/bench_profiled.json 2> $GRAFT_REPO_ROOT/$out/prof.log
cd $GRAFT_REPO_ROOT
python3 tools/rocpd_stats.py $out/prof/stats_results.db > $out/kernel_stats.txt
python3 tools/rocpd_timeline.py $out/prof/stats_results.db > $out/timeline.txt 2>&1
head -14 $out/kernel_stats.txt; cat $out/timeline.txt
find $out -name "*.db" -size +2M -delete
} 2>&1 | grep -v amdgpu.ids | tee $out/dgb.txt
