#!/bin/bash
# Runs ON the GPU box: HIP API statistics of one bench run (host synchronisations per LM iteration).  usage: tools/gpu_hiptrace.sh TAG SHAPE STEPS
tag=$1; shape=${2:-C2}; steps=${3:-400}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --hip-trace --stats -d $out/hip -o hip -f csv -- python3 $GRAFT_REPO_ROOT/bench.py --shape $shape --cpu-sample-pts 0 --steps $steps --warmup 8 > $out/bench_$shape.json 2> $out/hip.log
ls $out/hip | head
f=$(ls $out/hip/*hip_api_stats.csv 2>/dev/null | head -1)
echo "# rocprofv3 --hip-trace --stats -- python3 bench.py --shape $shape --steps $steps --warmup 8 (whole process: set-up, warm-up, $steps timed iterations, per-kernel timing passes)" > $out/hip_api_stats_$shape.txt
head -25 "$f" >> $out/hip_api_stats_$shape.txt
cat $out/hip_api_stats_$shape.txt
find $out -name "*.csv" -size +1M -delete
