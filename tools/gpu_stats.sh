#!/bin/bash
# Runs ON the GPU box: rocprofv3 --kernel-trace --stats over `python3 bench.py` (no CPU baseline leg); per-kernel table
# to gpurun_out/$1/kernel_stats.txt.  usage: tools/gpu_stats.sh TAG [bench args...]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/prof -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample-pts 0 "$@" > $out/bench_profiled.json 2> $out/prof.log
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $out/prof/stats_results.db > $out/kernel_stats.txt
head -32 $out/kernel_stats.txt
