#!/bin/bash
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_layout.py -m gpu -x -q > $out/pytest_layout.log 2>&1
tail -12 $out/pytest_layout.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $out/pytest_parity.log 2>&1
tail -6 $out/pytest_parity.log
BENCH_ARGS="" bash tools/gpu_ab.sh $tag - SATBA_HOST_LOOP=1
python bench.py --cpu-sample-pts 0 --driver native-sync --steps 80 --warmup 8 > $out/sync.json 2>>$out/ab.err; python -c "
import json; d=json.load(open('$out/sync.json')); print('native-sync', round(d['value'],1), d['final_cost'], d['accepted_steps'])"
for s in C2 C3 C5; do for drv in native native-sync; do python bench.py --shape $s --cpu-sample-pts 0 --driver $drv > $out/bench_${s}_$drv.json 2>> $out/ab.err; python -c "
import json; d=json.load(open('$out/bench_${s}_$drv.json')); print('$s $drv', round(d['value'],1), 'it/s', d['final_cost'], d['accepted_steps'])"; done; done
python bench.py --cpu-sample-pts 0 --loss soft_l1 > $out/bench_soft_l1.json 2>> $out/ab.err; python -c "
import json; d=json.load(open('$out/bench_soft_l1.json')); print('soft_l1', round(d['value'],1), 'it/s', d['final_cost'], d['accepted_steps'], d['solve_shipped_tolerances'])"
tail -5 $out/ab.err
