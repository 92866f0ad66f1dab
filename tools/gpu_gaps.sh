#!/bin/bash
# Runs ON the GPU box: kernel trace of bench.py under the given environment, then the gap analysis.  usage: tools/gpu_gaps.sh TAG "ENV=.." [bench args]
tag=$1; e=$2; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
if [ "$e" != "-" ]; then export $e; fi
rocprofv3 --kernel-trace -d $out/prof -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample-pts 0 --steps 60 --warmup 4 "$@" > $out/bench_profiled.json 2> $out/prof.log
python3 $GRAFT_REPO_ROOT/tools/rocpd_gaps.py $out/prof/stats_results.db 0.4 > $out/gaps.txt
head -40 $out/gaps.txt
find $out -name "*.db" -size +2M -delete
