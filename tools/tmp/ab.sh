cd $GRAFT_REPO_ROOT
for i in 1 2; do
SATBA_LIB=$GRAFT_REPO_ROOT/tools/tmp/lib_old.so python bench.py --cpu-sample-pts 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', round(d['value'],1), d['kernel_ms'])"
python bench.py --cpu-sample-pts 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', round(d['value'],1), d['kernel_ms'])"
done
