tools/gpu.sh r5h tests
grep -E "^FAILED|^ERROR" gpurun_out/r5h/pytest_gpu.log | head -30
tools/gpu.sh r5h_lin bench --cpu-sample-pts 0
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5h_lin/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('host_driver','chol_beside','value')}); print(json.dumps(d['e2e'])[:1200])
PY
tools/gpu.sh r5h_soft bench --cpu-sample-pts 0 --loss soft_l1
