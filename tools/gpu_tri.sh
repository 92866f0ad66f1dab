#!/bin/bash
# Runs ON the GPU box: the triangulation bench lines (tools/tri_bench.py), its per-kernel table (rocprofv3 --kernel-trace --stats) and
# the VALU counters of k_tri_points, to gpurun_out/$1/.   usage: tools/gpu_tri.sh TAG
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
timeout 600 python3 tools/tri_bench.py C5 C3 P3 > $out/triangulation.jsonl 2> $out/tri.log
cat $out/triangulation.jsonl | cut -c1-330
cd /tmp && export TMPDIR=/tmp
for sh in C5 C3; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/tri_prof_$sh -o stats -- python3 $GRAFT_REPO_ROOT/tools/tri_bench.py $sh > /dev/null 2> $out/tri_prof_$sh.log
  python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $out/tri_prof_$sh/stats_results.db > $out/triangulation_kernel_stats_$sh.txt
  grep -E "k_tri|Scan|NAME|name" $out/triangulation_kernel_stats_$sh.txt | head -12
done
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace -d $out/tri_pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/tools/tri_bench.py C5 > $out/tri_pmc$i.log 2>&1
done
mkdir -p $out/tri_pmc; for k in 1 2 3; do mv $out/tri_pmc$k $out/tri_pmc/pmc$k; done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/tri_pmc "k_tri_points" > $out/triangulation_pmc_C5.txt
cat $out/triangulation_pmc_C5.txt
find $out -name "*.csv" -size +1M -delete
find $out -name "*.db" -size +8M -delete
