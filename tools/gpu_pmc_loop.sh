#!/bin/bash
# Runs ON the GPU box: HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes) of the kernels inside LM iterations.  usage: tools/gpu_pmc_loop.sh TAG SHAPE LOSS
tag=$1; shape=${2:-C4}; loss=${3:-linear}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace -d $out/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/tools/kernel_times.py $shape $loss 9 --loop > $out/pmc$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out "k_linearize|k_residual|k_schur_pairs|k_schur_diag" > $out/pmc_summary.txt
cat $out/pmc_summary.txt
find $out -name "*.csv" -size +1M -delete
find $out -name "*.db" -size +8M -delete
