#!/bin/bash
# Runs ON the GPU box (via gpurun): GPU parity tests, then the headline bench without the CPU baseline leg.
# usage: tools/gpu_check.sh TAG [bench args...]
tag=$1; shift
mkdir -p gpurun_out/$tag
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1
grep -E "passed|failed|rror" gpurun_out/$tag/pytest.log | tail -5
python bench.py --cpu-sample-pts 0 "$@" > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/$tag/bench.json"))
print(round(d["value"], 1), "it/s", round(d["ms_per_step"], 3), "ms  frac", round(d["roofline"]["frac"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items()})
PY
