#!/bin/bash
# on the GPU box: the dense-solve harness at every test size, then the stamped build's step trace at n = 1000 -> gpurun_out/TAG/chol_harness.txt
cd $GRAFT_REPO_ROOT/tools/chol
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
SATBA_CHECK_ALL=1 timeout 300 ./chol_bench 40 9 30 45 50 64 65 66 96 129 130 162 192 200 250 300 500 900 995 1000 1200 1300 > $out/chol_harness.txt 2>&1
SATBA_STAMPS=1 timeout 120 ./chol_bench_st 3 1000 >> $out/chol_harness.txt 2>&1
grep -c "fail 0" $out/chol_harness.txt; grep "all ok\|FAIL" $out/chol_harness.txt
