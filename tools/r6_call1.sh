#!/bin/bash
# round 6, first GPU call: test suite, tight parity, baseline bench, k_linearize floor (ablation), LDS conflict ubench, e2e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6a; mkdir -p $out
bash tools/gpu.sh r6a tests
( echo "# tools/tight_metrics.py on MI355X, round 6 (goldens: 3-point runs restarted to their stationary point)"; timeout 600 python tools/tight_metrics.py 2>&1
  echo "--- camera-major route (SATBA_DETERMINISTIC=1)"; SATBA_DETERMINISTIC=1 timeout 600 python tools/tight_metrics.py 2>&1 ) | grep -v amdgpu.ids > $out/tight_parity.txt
tail -30 $out/tight_parity.txt | cut -c1-260
bash tools/gpu.sh r6a bench --steps 200 --warmup 20
( for shape in C4 C2; do
    echo "== $shape shipped library"; timeout 300 python tools/kernel_times.py $shape linear 20
    echo "== $shape LDS atomics compiled out (-DSATBA_ABLATE_CAM_ATOMICS)"; SATBA_LIB=$GRAFT_REPO_ROOT/sat-bundleadjust_amd/satba/lib/var/libsatba_noatomics.so timeout 300 python tools/kernel_times.py $shape linear 20 --only linearize
  done ) 2>&1 | grep -v amdgpu.ids > $out/linearize_floor.txt
cat $out/linearize_floor.txt
timeout 120 tools/ubench/lds_conflicts.bin > $out/lds_conflicts.txt 2>&1; cat $out/lds_conflicts.txt
( echo "== new"; timeout 300 python tools/e2e_time.py C4; echo "== old path (SATBA_ERR_OVERLAP=0 SATBA_COPY_DIRECT=1)"; SATBA_ERR_OVERLAP=0 SATBA_COPY_DIRECT=1 timeout 300 python tools/e2e_time.py C4
  echo "== C3 new"; timeout 300 python tools/e2e_time.py C3 ) 2>&1 | grep -v amdgpu.ids > $out/e2e.txt
cat $out/e2e.txt | cut -c1-1200
