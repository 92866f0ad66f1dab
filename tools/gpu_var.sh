#!/bin/bash
# Runs ON the GPU box: per-kernel times of the default library and of every variant build under satba/lib/var/.
# usage: tools/gpu_var.sh [shape] [loss]
sh=${1:-C4}; loss=${2:-linear}
echo default; timeout 300 python tools/kernel_times.py $sh $loss 40 2>&1 | tail -1
for f in sat-bundleadjust_amd/satba/lib/var/*.so; do
  echo $f; SATBA_LIB=$PWD/$f timeout 300 python tools/kernel_times.py $sh $loss 40 2>&1 | tail -1
done
