mkdir -p gpurun_out/r5o
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/r5o/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r5o/pytest.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/r5o/pytest.log | head
tools/gpu.sh r5o_C2 bench --cpu-sample-pts 0 --shape C2 --no-e2e --steps 400
SATBA_NO_TAIL_FUSION=1 tools/gpu.sh r5o_C2_nofuse bench --cpu-sample-pts 0 --shape C2 --no-e2e --steps 400
tools/gpu.sh r5o_C2s bench --cpu-sample-pts 0 --shape C2 --no-e2e --steps 400 --loss soft_l1
tools/gpu.sh r5o_C4 bench --cpu-sample-pts 0 --no-e2e
