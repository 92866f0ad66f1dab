#!/bin/bash
# Runs ON the GPU box: the dense-solver tests in every factorisation mode, then per-kernel times per mode.
mkdir -p gpurun_out/chol
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cholesky" > gpurun_out/chol/pytest.log 2>&1
tail -5 gpurun_out/chol/pytest.log
for m in 0 5 2; do
  SATBA_CHOL=$m timeout 300 python tools/kernel_times.py C4 linear 40 2>&1 | tail -1
done
SATBA_CHOL=0 timeout 300 python tools/kernel_times.py C3 linear 40 2>&1 | tail -1
