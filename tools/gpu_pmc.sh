#!/bin/bash
# Runs ON the GPU box: rocprofv3 PMC passes (counters only, separate passes, no trace domains besides --kernel-trace) over
# tools/kernel_times.py; per-kernel averages to gpurun_out/$1/pmc_summary.txt.   usage: tools/gpu_pmc.sh TAG SHAPE LOSS [kernel regex]
tag=$1; shape=${2:-C4}; loss=${3:-linear}; pat=${4:-k_linearize}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace -d $out/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/tools/kernel_times.py $shape $loss 4 > $out/pmc$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out "$pat" > $out/pmc_summary.txt
cat $out/pmc_summary.txt
find $out -name "*.csv" -size +1M -delete
find $out -name "*.db" -size +8M -delete
