#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6i; mkdir -p $out
echo "== no sum, sleep 0.3 s per message"; NOSUM=1 SLEEP=0.3 python tools/debug_messages2.py 2>&1 | grep -v "amdgpu\|Gloo\|socket.cpp"
echo "== sum"; python tools/debug_messages2.py 2>&1 | grep -v "amdgpu\|Gloo\|socket.cpp"
( cd tools/chol; for i in 1 2; do for b in chol_bench_r5 chol_bench; do echo "== $b"; timeout 120 ./$b 10 250 1000 2>&1 | grep "driver\|FAIL\|fault"; done; done ) | cut -c1-40,100-170
( cd tools/chol; for i in $(seq 1 20); do SATBA_CHECK_ALL=1 timeout 60 ./chol_bench 20 65 66 67 70 96 129 130 192 200 2>&1 | grep "mirror:\|rep \|FAILED\|all ok" | cut -c1-160; done | sort | uniq -c | sort -rn | head -5 )
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -k "messages or two_ranks or rccl or sharded or dense or chol or beside or factoris" 2>&1 | tail -4
