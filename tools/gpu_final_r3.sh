#!/bin/bash
# Runs ON the GPU box: the evidence set of round 3.  usage: tools/gpu_final_r3.sh TAG
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout 200 > $out/pytest_gpu.log 2>&1
grep -E "passed|failed|rror" $out/pytest_gpu.log | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 600 python bench.py "$@" > $out/bench.json 2> $out/bench.err
timeout 200 python bench.py --driver native --cpu-sample-pts 0 > $out/bench_device_loop.json 2>> $out/bench.err
timeout 200 python bench.py --camera-major --cpu-sample-pts 0 > $out/bench_camera_major.json 2>> $out/bench.err
timeout 200 python bench.py --loss soft_l1 --cpu-sample-pts 0 > $out/bench_soft_l1.json 2>> $out/bench.err
for s in C2 C3 P3 C5; do timeout 200 python bench.py --shape $s --cpu-sample-pts 0 > $out/bench_$s.json 2>> $out/bench.err; done
for s in C2 C3; do timeout 200 python bench.py --shape $s --driver native-sync --cpu-sample-pts 0 > $out/bench_${s}_host_loop.json 2>> $out/bench.err; done
# repeatability: eight default runs, final_cost must be the same bits
for i in 1 2 3 4 5 6 7 8; do timeout 200 python bench.py --cpu-sample-pts 0 --steps 40 --warmup 4 >> $out/bench_repeat.jsonl 2>> $out/bench.err; done
python tools/create_time.py > $out/create_time.json 2>> $out/bench.err
python tools/rpcfit_bench.py 50 > $out/rpcfit.jsonl 2>> $out/bench.err; python tools/rpcfit_bench.py 200 >> $out/rpcfit.jsonl 2>> $out/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample-pts 0 --steps 100 > $out/bench_profiled.json 2> $out/prof.log
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $out/prof/stats_results.db > $out/kernel_stats.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $out/profs -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample-pts 0 --steps 100 --loss soft_l1 > $out/bench_profiled_soft_l1.json 2> $out/profs.log
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $out/profs/stats_results.db > $out/kernel_stats_soft_l1.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof5 -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --shape C5 --cpu-sample-pts 0 --steps 100 > $out/bench_profiled_C5.json 2> $out/prof5.log
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $out/prof5/stats_results.db > $out/kernel_stats_C5.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof2 -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --shape C2 --cpu-sample-pts 0 --steps 200 > $out/bench_profiled_C2.json 2> $out/prof2.log
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $out/prof2/stats_results.db > $out/kernel_stats_C2.txt
cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc.sh $tag/pmc C4 linear "k_linearize|k_schur_pairs|k_schur_diag|k_residual|k_backsub|k_jvp" > /dev/null 2>&1
bash tools/gpu_pmc.sh $tag/pmcs C4 soft_l1 "k_linearize|k_schur_pairs|k_schur_diag" > /dev/null 2>&1
bash tools/gpu_pmc.sh $tag/pmc5 C5 linear "k_linearize|k_residual|k_schur_pairs" > /dev/null 2>&1
find $out -name "*.db" -size +2M -delete
head -16 $out/kernel_stats.txt
cat $out/bench.json
