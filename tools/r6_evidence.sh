#!/bin/bash
# round 6 evidence set on one box: tests, bench lines of every shape and loss, kernel statistics + time line, PMC passes, dense-solve harness
cd $GRAFT_REPO_ROOT
tag=${1:-r6ev}
out=gpurun_out/$tag; mkdir -p $out
bash tools/gpu.sh $tag tests
bash tools/gpu.sh $tag benches
bash tools/gpu.sh $tag stats --steps 200 --warmup 20 --no-e2e
cp $out/kernel_stats.txt $out/kernel_stats_C4.txt; cp $out/timeline.txt $out/timeline_C4.txt
for s in C2 C3 C5; do bash tools/gpu.sh $tag stats --shape $s --no-e2e > /dev/null; cp $out/kernel_stats.txt $out/kernel_stats_$s.txt; done
bash tools/gpu.sh $tag stats --loss soft_l1 --steps 100 --warmup 10 --no-e2e > /dev/null; cp $out/kernel_stats.txt $out/kernel_stats_soft_l1.txt; cp $out/timeline.txt $out/timeline_C4_soft_l1.txt
SATBA_CHOL_BESIDE=0 bash tools/gpu.sh $tag stats --steps 100 --warmup 10 --no-e2e > /dev/null; cp $out/kernel_stats.txt $out/kernel_stats_sequential.txt
bash tools/gpu.sh $tag pmc_loop C4 linear > /dev/null; cp $out/pmc_summary.txt $out/pmc_loop_C4.txt
bash tools/gpu.sh $tag pmc_loop C4 soft_l1 > /dev/null; cp $out/pmc_summary.txt $out/pmc_loop_C4_soft_l1.txt
bash tools/gpu.sh $tag pmc_loop C5 linear > /dev/null; cp $out/pmc_summary.txt $out/pmc_loop_C5.txt
bash tools/gpu.sh $tag pmc C4 linear k_linearize > /dev/null; cp $out/pmc_summary.txt $out/pmc_linearize_C4.txt
bash tools/chol/run_stamps.sh $tag
for s in C2 C3 C5 C4; do timeout 300 python tools/e2e_time.py $s > $out/e2e_$s.json 2>/dev/null; done
timeout 300 python tools/e2e_time.py C4 soft_l1 > $out/e2e_C4_soft_l1.json 2>/dev/null
timeout 1500 python bench.py --steps 20 --warmup 5 --cpu-c3 > $out/bench_cpu_c3.json 2> $out/bench_cpu_c3.err; cp gpurun_out/cpu_baseline_C3.json $out/ 2>/dev/null
ls $out; head -30 $out/kernel_stats_C4.txt; cat $out/timeline_C4.txt; head -22 $out/kernel_stats_C2.txt; cat $out/pmc_loop_C4.txt
