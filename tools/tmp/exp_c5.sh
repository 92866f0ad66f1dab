cd $GRAFT_REPO_ROOT
for v in "SATBA_X=0" "SATBA_SPLIT=0" "SATBA_SPLIT=2" "SATBA_SPLIT=3" "SATBA_SPLIT=2 SATBA_BPC=1" "SATBA_SPLIT=1 SATBA_BPC=1" "SATBA_BPC=3"; do
  env $v timeout 300 python3 tools/kernel_times.py C5 linear 20 2>&1 | tail -1
done
