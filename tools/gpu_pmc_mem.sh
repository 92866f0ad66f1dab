#!/bin/bash
# Runs ON the GPU box: memory-pipeline PMC passes (texture addresser / L1 / L2) over tools/kernel_times.py for one kernel.
# usage: tools/gpu_pmc_mem.sh TAG SHAPE LOSS [kernel regex]
tag=$1; shape=${2:-C4}; loss=${3:-linear}; pat=${4:-k_schur_pairs}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
# (a pass with four TA counters "exceeds the capabilities of the hardware" and rocprofv3 then hangs: two per pass, and a timeout)
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $set --kernel-trace -d $out/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/tools/kernel_times.py $shape $loss 4 > $out/pmc$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out "$pat" > $out/pmc_summary.txt
cat $out/pmc_summary.txt
find $out -name "*.csv" -size +1M -delete
find $out -name "*.db" -size +8M -delete
