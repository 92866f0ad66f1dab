#!/bin/bash
# Runs ON the GPU box: the linear and the robust line of the bench with their kernel statistics.  usage: tools/gpu_pvs.sh TAG [env switches for A/B]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
BENCH_ARGS="--loss soft_l1" bash tools/gpu_ab.sh $tag - "$@"
bash tools/gpu_ab.sh $tag/lin - "$@"
cd /tmp && export TMPDIR=/tmp
for l in soft_l1 linear; do
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_$l -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample-pts 0 --steps 60 --loss $l > $out/bench_profiled_$l.json 2> $out/prof_$l.log
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $out/prof_$l/stats_results.db > $out/kernel_stats_$l.txt
head -5 $out/kernel_stats_$l.txt
done
find $out -name "*.db" -size +2M -delete
