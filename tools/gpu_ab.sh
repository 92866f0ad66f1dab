#!/bin/bash
# A/B runs of the headline bench under environment switches: tools/gpu_ab.sh TAG "ENV1=.. ENV2=.." "ENV=.." ...   ("-" = no switch)
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
i=0
for e in "$@"; do
  i=$((i+1))
  if [ "$e" = "-" ]; then e=""; fi
  env $e python bench.py --cpu-sample-pts 0 --steps 80 --warmup 8 $BENCH_ARGS > $out/ab_$i.json 2>> $out/ab.err
  python - <<PY
import json
try:
    d = json.load(open("$out/ab_$i.json"))
    r = d["roofline"]
    print("[$e]", round(d["value"], 1), "it/s", round(d["ms_per_step"], 3), "ms  lin", round(r["ms_per_launch"], 4), repr(d["final_cost"]), {k: round(v, 4) for k, v in d["kernel_ms"].items()})
except Exception as ex:
    print("[$e] failed", ex)
PY
done
