#!/bin/bash
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r6m; mkdir -p $out
export SATBA_SCHUR_MERGE=1 SATBA_CHOL_BESIDE_WGS=8
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --shape C3 --cpu-sample-pts 0 --steps 100 --no-e2e > $out/bench_profiled.json 2> $out/prof.log
cd $GRAFT_REPO_ROOT
python3 tools/rocpd_stats.py $out/prof/stats_results.db > $out/kernel_stats.txt
python3 tools/rocpd_timeline.py $out/prof/stats_results.db > $out/timeline.txt 2>&1
head -24 $out/kernel_stats.txt; cat $out/timeline.txt
find $out -name "*.db" -size +2M -delete
