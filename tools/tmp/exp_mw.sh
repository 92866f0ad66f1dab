cd $GRAFT_REPO_ROOT
for v in "SATBA_X=0" "SATBA_SCHUR_MERGE_W=1" "SATBA_SCHUR_CHUNKS=2" "SATBA_SCHUR_CHUNKS=4"; do
  env $v timeout 300 python3 tools/kernel_times.py C4 soft_l1 20 2>&1 | tail -1
done
