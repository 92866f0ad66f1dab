#!/bin/bash
# Runs ON the GPU box: Schur / alternative-path parity tests, then per-kernel times for the headline shapes.
mkdir -p gpurun_out/schur
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "schur or alternative or phases" > gpurun_out/schur/pytest.log 2>&1
tail -3 gpurun_out/schur/pytest.log
for sh in C4 C3 C2; do timeout 300 python tools/kernel_times.py $sh linear 40 2>&1 | tail -1; done
timeout 300 python tools/kernel_times.py C4 soft_l1 40 2>&1 | tail -1
