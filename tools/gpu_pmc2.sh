#!/bin/bash
# Runs ON the GPU box: a short list of PMC passes over one kernel under environment switches.  usage: tools/gpu_pmc2.sh TAG SHAPE LOSS REGEX "ENV=.." ...
tag=$1; shape=$2; loss=$3; pat=$4; shift 4
cd /tmp && export TMPDIR=/tmp
j=0
for e in "$@"; do
  j=$((j+1))
  if [ "$e" = "-" ]; then e=""; fi
  for v in $e; do export $v; done
  out=$GRAFT_REPO_ROOT/gpurun_out/$tag/v$j
  mkdir -p $out
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SMEM" "FETCH_SIZE" "SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"; do
    i=$((i+1))
    timeout 240 rocprofv3 --pmc $set --kernel-trace -d $out/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/tools/kernel_times.py $shape $loss 4 > $out/pmc$i.log 2>&1
  done
  echo "== [$e]"
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out "$pat" | tee $out/pmc_summary.txt
  find $out -name "*.csv" -size +1M -delete
  find $out -name "*.db" -size +8M -delete
  for v in $e; do unset ${v%%=*}; done
done
