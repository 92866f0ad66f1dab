set -x
mkdir -p gpurun_out/r5f
timeout 1200 python -m pytest tests/test_gpu_layout.py tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "beside or tight or many_times or cholesky" > gpurun_out/r5f/pytest.log 2>&1; tail -15 gpurun_out/r5f/pytest.log
tools/gpu.sh r5f_soft bench --cpu-sample-pts 0 --loss soft_l1
SATBA_DEVICE_LOOP=1 tools/gpu.sh r5f_soft_dev bench --cpu-sample-pts 0 --loss soft_l1
SATBA_CHOL_BESIDE=0 tools/gpu.sh r5f_soft_seq bench --cpu-sample-pts 0 --loss soft_l1
tools/gpu.sh r5f_lin bench --cpu-sample-pts 0
