mkdir -p gpurun_out/r5m
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/r5m/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r5m/pytest.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/r5m/pytest.log | head
tools/gpu.sh r5m_C5 bench --cpu-sample-pts 0 --shape C5 --no-e2e
tools/gpu.sh r5m_C5s bench --cpu-sample-pts 0 --shape C5 --loss soft_l1 --no-e2e
tools/gpu.sh r5m_C2 bench --cpu-sample-pts 0 --shape C2 --no-e2e
tools/gpu.sh r5m_C3 bench --cpu-sample-pts 0 --shape C3 --no-e2e
tools/gpu.sh r5m_P3 bench --cpu-sample-pts 0 --shape P3 --no-e2e
