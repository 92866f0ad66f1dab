#!/bin/bash
# Runs ON the GPU box: PMC passes over k_linearize (tools/lin_time.py --child), summaries to gpurun_out/$1/
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $out/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/tools/lin_time.py --child > $out/pmc$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out k_linearize > $out/pmc_linearize.txt
cat $out/pmc_linearize.txt
find $out -name "*.csv" -size +1M -delete
