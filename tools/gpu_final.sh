#!/bin/bash
# Runs ON the GPU box: the evidence set of a round -- GPU tests, the default bench line (with the CPU baseline leg),
# the soft_l1 line, rocprofv3 --kernel-trace --stats of `python3 bench.py`, and the two PMC passes (FETCH_SIZE, WRITE_SIZE)
# the roofline `traffic` figure comes from.  usage: tools/gpu_final.sh TAG
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1
grep -E "passed|failed|rror" $out/pytest_gpu.log | tail -3
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --loss soft_l1 --cpu-sample-pts 0 > $out/bench_soft_l1.json 2>> $out/bench.err
for s in C2 C3 P3 C5; do python bench.py --shape $s --cpu-sample-pts 0 > $out/bench_$s.json 2>> $out/bench.err; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/prof -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample-pts 0 > $out/bench_profiled.json 2> $out/prof.log
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $out/prof/stats_results.db > $out/kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $out/pmc_$c -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample-pts 0 --steps 3 --warmup 1 --kernel-reps 2 > $out/pmc_$c.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out > $out/pmc_hbm_traffic.txt
grep -A3 "k_linearize" $out/pmc_hbm_traffic.txt | head -8
head -12 $out/kernel_stats.txt
cat $out/bench.json
