#!/bin/bash
# round 3, first GPU check: determinism tests + tight parity, then the headline bench lines
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_layout.py -m gpu -x -q > $out/pytest_layout.log 2>&1
tail -15 $out/pytest_layout.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "tight or full_size or alternative or linearize or blocks" > $out/pytest_parity.log 2>&1
tail -25 $out/pytest_parity.log
python bench.py --cpu-sample-pts 0 > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err
python bench.py --cpu-sample-pts 0 --steps 100 > $out/bench_b.json 2>> $out/bench.err
python bench.py --cpu-sample-pts 0 --camera-major > $out/bench_camera_major.json 2>> $out/bench.err
python bench.py --cpu-sample-pts 0 --loss soft_l1 > $out/bench_soft_l1.json 2>> $out/bench.err
for s in C2 C3 C5; do python bench.py --shape $s --cpu-sample-pts 0 > $out/bench_$s.json 2>> $out/bench.err; done
python - <<PY
import json
for n in ("bench", "bench_b", "bench_camera_major", "bench_soft_l1", "bench_C2", "bench_C3", "bench_C5"):
    try:
        d = json.load(open("$out/%s.json" % n))
        r = d["roofline"]
        print(n, round(d["value"], 1), "it/s", round(d["ms_per_step"], 3), "ms  lin", round(r["ms_per_launch"], 4), "frac", round(r["frac"], 3), "as_built", round(r["frac_as_built"], 3), d["camera_sums"], d["fixed_point_fallbacks"], repr(d["final_cost"]), {k: round(v, 4) for k, v in d["kernel_ms"].items()})
    except Exception as e:
        print(n, "failed", e)
PY
