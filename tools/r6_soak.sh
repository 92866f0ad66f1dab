#!/bin/bash
# round 6 soak of the final build: test suite x3 (+ once with the host loop), fuzzers, the dense-solve hand-over hunt, ASan build of the shim
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6soak; mkdir -p $out
{
echo "GPU test suite, three runs in a row and one with SATBA_DEVICE_LOOP=0 (one MI355X box, final build of round 6):"
for i in 1 2 3; do timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed" ; done
SATBA_DEVICE_LOOP=0 timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed"
echo; echo "tools/fuzz_beside.py 100 300 (concurrent front against sequential front, bit for bit):"
timeout 1500 python tools/fuzz_beside.py 100 300 2>&1 | tail -1
echo; echo "tools/fuzz_solve.py 150 500 (device-resident solve against the Python loop on the CPU oracle; 'ok' = cost within 1e-7):"
timeout 2400 python tools/fuzz_solve.py 150 500 > $out/fuzz_solve.log 2>&1; echo "cases ok: $(grep -c ' ok ' $out/fuzz_solve.log)"; grep "DIFF" $out/fuzz_solve.log | head -5
echo; echo "dense solve, hand-over hunt: 40 processes x sizes 65 66 67 70 96 129 130 192 200 250 x 20 repetitions, every solve and mirror checked:"
( cd tools/chol; for i in $(seq 1 40); do SATBA_CHECK_ALL=1 timeout 60 ./chol_bench 20 65 66 67 70 96 129 130 192 200 250 2>&1 | grep "mirror:\|rep \|FAILED\|all ok" | cut -c1-160; done | sort | uniq -c | sort -rn | head -5 )
echo; echo "host-side AddressSanitizer build of the shim + tests/asan/abi_driver.c on the GPU box (live handle: begin / fetch, copy lanes):"
( cd sat-bundleadjust_amd/csrc && make asan_check 2>&1 | tail -1 )
} 2>&1 | grep -v amdgpu.ids | tee $out/soak.txt
