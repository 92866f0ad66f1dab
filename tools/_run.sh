mkdir -p gpurun_out/r5q
for s in C2 C3 C5 C4; do python tools/e2e_time.py $s > gpurun_out/r5q/e2e_$s.json 2>/dev/null; done
python tools/e2e_time.py C4 soft_l1 > gpurun_out/r5q/e2e_C4_soft_l1.json 2>/dev/null
python - <<'PY'
import json
for s in ('C2','C3','C5','C4','C4_soft_l1'):
    d=json.load(open('gpurun_out/r5q/e2e_%s.json'%s))
    for c in d['calls']:
        print(s, c['call'], {k: round(v*1e3,2) for k,v in c.items() if k.endswith('_s')}, 'overhead', round(c['host_overhead_frac'],3), 'nfev', c['nfev'])
PY
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x > gpurun_out/r5q/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r5q/pytest.log | tail -1
