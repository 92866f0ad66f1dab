#!/bin/bash
# b128 camera-table reads (CAMC 26 / JVP_ROW 14) against the shipped build: per-kernel times and bench lines in alternation, then the GPU suite on the variant
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6j; mkdir -p $out
V=$GRAFT_REPO_ROOT/sat-bundleadjust_amd/satba/lib/var/libsatba_b128.so
{
for r in 1 2; do for lib in base b128; do
  if [ $lib != base ]; then export SATBA_LIB=$V; else unset SATBA_LIB; fi
  echo "== $lib C4 linear"; python3 tools/kernel_times.py C4 linear 20 2>&1 | grep "^{" | cut -c1-200
done; done
for lib in base b128; do
  if [ $lib != base ]; then export SATBA_LIB=$V; else unset SATBA_LIB; fi
  echo "== $lib C4 soft_l1"; python3 tools/kernel_times.py C4 soft_l1 20 2>&1 | grep "^{" | cut -c1-200
  echo "== $lib C3 linear"; python3 tools/kernel_times.py C3 linear 20 2>&1 | grep "^{" | cut -c1-200
  echo "== $lib P3 linear"; python3 tools/kernel_times.py P3 linear 20 2>&1 | grep "^{" | cut -c1-200
  echo "== $lib C5 linear"; python3 tools/kernel_times.py C5 linear 20 2>&1 | grep "^{" | cut -c1-200
done
for r in 1 2; do for lib in base b128; do
  if [ $lib != base ]; then export SATBA_LIB=$V; else unset SATBA_LIB; fi
  for sh in C4 C3 C2; do
    echo "== $lib bench $sh"; python3 bench.py --shape $sh --steps 200 --cpu-sample-pts 0 --no-e2e 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['ms_per_launch'], d['kernel_ms'])"
  done
done; done
export SATBA_LIB=$V
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -x 2>&1 | tail -5
} 2>&1 | grep -v amdgpu.ids | tee $out/b128.txt
