#!/bin/bash
# Runs ON the GPU box: the evidence set of a round -- GPU tests, the default bench line (with the CPU baseline leg),
# the soft_l1 line and the other shapes, rocprofv3 --kernel-trace --stats of `python3 bench.py`, and the PMC passes the
# roofline `traffic` figure comes from, and the triangulation evidence (tools/gpu_tri.sh).  usage: tools/gpu_final.sh TAG [--cpu-c3]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1
grep -E "passed|failed|rror" $out/pytest_gpu.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
python bench.py "$@" > $out/bench.json 2> $out/bench.err
python bench.py --loss soft_l1 --cpu-sample-pts 0 > $out/bench_soft_l1.json 2>> $out/bench.err
for s in C2 C3 P3 C5; do python bench.py --shape $s --cpu-sample-pts 0 > $out/bench_$s.json 2>> $out/bench.err; done
python tools/create_time.py > $out/create_time.json 2>> $out/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/prof -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample-pts 0 --steps 100 > $out/bench_profiled.json 2> $out/prof.log
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $out/prof/stats_results.db > $out/kernel_stats.txt
rocprofv3 --kernel-trace --stats -d $out/prof5 -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --shape C5 --cpu-sample-pts 0 --steps 100 > $out/bench_profiled_C5.json 2> $out/prof5.log
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $out/prof5/stats_results.db > $out/kernel_stats_C5.txt
cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc.sh $tag/pmc C4 linear "k_linearize|k_schur_pairs|k_schur_diag|k_residual|k_backsub|k_jvp" > /dev/null 2>&1
bash tools/gpu_pmc.sh $tag/pmc5 C5 linear "k_linearize|k_residual|k_schur_pairs" > /dev/null 2>&1
bash tools/gpu_pmc_mem.sh $tag/pmcm C4 linear "k_schur_pairs<|k_schur_diag<" > /dev/null 2>&1
bash tools/gpu_tri.sh $tag > $out/tri_summary.txt 2>&1
find $out -name "*.db" -size +2M -delete
head -14 $out/kernel_stats.txt
cat $out/bench.json
