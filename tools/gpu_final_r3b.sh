#!/bin/bash
# Runs ON the GPU box: the evidence set of the second half of round 3 (files named r3b_*).  usage: tools/gpu_final_r3b.sh TAG
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout 200 > $out/pytest_gpu.log 2>&1
grep -E "passed|failed|rror" $out/pytest_gpu.log | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
timeout 200 python bench.py --loss soft_l1 --cpu-sample-pts 0 > $out/bench_soft_l1.json 2>> $out/bench.err
timeout 200 python bench.py --camera-major --cpu-sample-pts 0 > $out/bench_camera_major.json 2>> $out/bench.err
for s in C2 C3 P3 C5; do timeout 200 python bench.py --shape $s --cpu-sample-pts 0 > $out/bench_$s.json 2>> $out/bench.err; done
for s in C3 P3 C5; do timeout 200 python bench.py --shape $s --loss soft_l1 --cpu-sample-pts 0 > $out/bench_${s}_soft_l1.json 2>> $out/bench.err; done
cd /tmp && export TMPDIR=/tmp
for l in linear soft_l1; do
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_$l -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample-pts 0 --steps 100 --loss $l > $out/bench_profiled_$l.json 2> $out/prof_$l.log
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $out/prof_$l/stats_results.db > $out/kernel_stats_$l.txt
done
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_C3s -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --shape C3 --loss soft_l1 --cpu-sample-pts 0 --steps 100 > $out/bench_profiled_C3_soft_l1.json 2> $out/prof_C3s.log
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $out/prof_C3s/stats_results.db > $out/kernel_stats_C3_soft_l1.txt
cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc2.sh $tag/pmc C4 soft_l1 "k_schur_pairs<|k_schur_diag<" - > $out/pmc_pairs_soft_l1.txt 2>&1
find $out -name "*.db" -size +2M -delete
head -14 $out/kernel_stats_soft_l1.txt
for f in $out/bench.json $out/bench_soft_l1.json $out/bench_C*.json $out/bench_P3*.json; do python3 -c "
import json,sys
d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value'],1), round(d['ms_per_step'],3), d.get('solve_shipped_tolerances',{}).get('nfev'), d.get('solve_shipped_tolerances',{}).get('status'))"; done
