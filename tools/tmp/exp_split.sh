cd $GRAFT_REPO_ROOT
for v in "SATBA_X=0" "SATBA_SPLIT=1" "SATBA_SPLIT=2" "SATBA_BPC=3" "SATBA_BPC=4" "SATBA_SPLIT=1 SATBA_BPC=4"; do
  env $v timeout 300 python3 tools/kernel_times.py C4 linear 20 2>&1 | tail -1
done
