#!/bin/bash
# Runs ON the GPU box (through gpurun): the evidence runs of a round, one sub-command each.  Everything lands in gpurun_out/TAG/.
#   tools/gpu.sh TAG tests                       pytest -m gpu + smoke()
#   tools/gpu.sh TAG bench [bench.py args]       python bench.py ... -> bench.json  (repeat with other args / tags for other shapes)
#   tools/gpu.sh TAG benches                     the bench lines of every shape and loss (C4, C2, C3, P3, C5; linear and soft_l1)
#   tools/gpu.sh TAG stats [bench.py args]       rocprofv3 --kernel-trace --stats over bench.py (no CPU leg) -> kernel_stats.txt
#   tools/gpu.sh TAG pmc SHAPE LOSS REGEX        PMC passes (separate runs, counters only) over tools/kernel_times.py -> pmc_summary.txt
#   tools/gpu.sh TAG pmc_loop SHAPE LOSS         FETCH_SIZE / WRITE_SIZE of the kernels inside LM iterations -> pmc_summary.txt
#   tools/gpu.sh TAG chol N...                   tools/chol/chol_bench (dense solve harness) at the given sizes
#   tools/gpu.sh TAG calib                       tools/ubench/fetch_calib.bin: FETCH_SIZE / request counters against known byte counts
# (rounds 1-3 had one script per experiment: tools/gpu_*.sh, 21 of them)
tag=$1; cmd=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
summ() { python3 -c "
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline',{})
        print(f.split('/')[-1], 'it/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'frac', round(r.get('frac',0),3), 'as built', round(r.get('frac_as_built',0) or 0,3), 'nfev', d.get('solve_shipped_tolerances',{}).get('nfev'), {k: round(v,4) for k,v in d.get('kernel_ms',{}).items()})
    except Exception as e: print(f, 'unreadable:', e)
" "$@"; }
case $cmd in
tests)
  timeout 1500 python -m pytest tests -m gpu -q --timeout 300 > $out/pytest_gpu.log 2>&1
  grep -E "passed|failed|rror" $out/pytest_gpu.log | tail -3
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log ;;
bench)
  timeout 900 python bench.py "$@" > $out/bench.json 2> $out/bench.err; summ $out/bench.json ;;
benches)
  timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
  timeout 300 python bench.py --loss soft_l1 --cpu-sample-pts 0 > $out/bench_soft_l1.json 2>> $out/bench.err
  for s in C2 C3 P3 C5; do
    timeout 300 python bench.py --shape $s --cpu-sample-pts 0 > $out/bench_$s.json 2>> $out/bench.err
    timeout 300 python bench.py --shape $s --loss soft_l1 --cpu-sample-pts 0 > $out/bench_${s}_soft_l1.json 2>> $out/bench.err
  done
  summ $out/bench*.json ;;
stats)
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample-pts 0 "$@" > $out/bench_profiled.json 2> $out/prof.log
  python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $out/prof/stats_results.db > $out/kernel_stats.txt
  python3 $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $out/prof/stats_results.db > $out/timeline.txt 2>&1
  head -${HEAD_LINES:-34} $out/kernel_stats.txt; cat $out/timeline.txt
  find $out -name "*.db" -size +2M -delete ;;
pmc|pmc_loop)
  export SATBA_CHOL_BESIDE=0  # (counter collection runs one kernel at a time: the factorisation beside the pair kernel would only time out)
  shape=${1:-C4}; loss=${2:-linear}; pat=${3:-k_linearize}
  cd /tmp && export TMPDIR=/tmp
  if [ $cmd = pmc ]; then
    sets=("SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum")
    args="$shape $loss 4"
  else
    sets=("FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"); pat="k_linearize|k_residual|k_schur_pairs|k_schur_diag"
    args="$shape $loss 9 --loop"
  fi
  i=0
  for set in "${sets[@]}"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace -d $out/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/tools/kernel_times.py $args > $out/pmc$i.log 2>&1
  done
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out "$pat" > $out/pmc_summary.txt
  cat $out/pmc_summary.txt
  find $out -name "*.csv" -size +1M -delete; find $out -name "*.db" -size +8M -delete ;;
chol)
  cd tools/chol && timeout 300 ./chol_bench 10 "$@" > $out/chol.log 2>&1; grep "tiles\|driver\|ok\|FAIL" $out/chol.log | cut -c1-175 ;;
calib)
  cd /tmp && export TMPDIR=/tmp
  $GRAFT_REPO_ROOT/tools/ubench/fetch_calib.bin > $out/plain.txt 2>&1
  i=0
  for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace -d $out/c$i -o c -- $GRAFT_REPO_ROOT/tools/ubench/fetch_calib.bin > $out/c$i.log 2>&1
  done
  python3 $GRAFT_REPO_ROOT/tools/ubench/calib_summary.py $out > $out/summary.txt; cat $out/summary.txt ;;
*) echo "unknown sub-command $cmd"; exit 2 ;;
esac
