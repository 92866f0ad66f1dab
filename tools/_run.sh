set -x
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "tight or cam_sums or blocks or repeat" > gpurun_out/r5b/pytest.log 2>&1; tail -5 gpurun_out/r5b/pytest.log
timeout 600 python -m pytest tests/test_gpu_layout.py -m gpu -q --timeout 300 -k "serialised or beside" > gpurun_out/r5b/pytest2.log 2>&1; tail -8 gpurun_out/r5b/pytest2.log
SATBA_DETERMINISTIC=1 python tools/tight_metrics.py > gpurun_out/r5b/tight_cm.txt 2>&1; tail -12 gpurun_out/r5b/tight_cm.txt
