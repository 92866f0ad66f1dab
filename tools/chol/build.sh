#!/bin/bash
# builds tools/chol/chol_bench for gfx950 and prints the register / scratch use of the tile kernel's functions
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Wall -Wno-unused-function -I ../../sat-bundleadjust_amd/csrc $EXTRA chol_bench.hip -o ${OUT:-chol_bench} -save-temps=obj 2>&1 | grep -i "error\|warning" -A3 | head -30 || true
S=chol_bench-hip-amdgcn-amd-amdhsa-gfx950.s
for f in c3_chain_diag c3_chain_rider c3_chain_aux c3_chain_inv c3_chain_dnext c3_mirror_task c3_owner k_chol_tiles; do
  awk "/^_ZN5satba[0-9]*${f}.*:/,/; ScratchSize/" $S > /tmp/_f.s
  echo "$f: vgpr $(grep '; NumVgprs' /tmp/_f.s | tail -1 | awk '{print $3}') scratch $(grep '; ScratchSize' /tmp/_f.s | tail -1 | awk '{print $3}') code $(grep 'codeLenInByte' /tmp/_f.s | tail -1 | awk '{print $4}') spill_st $(grep -c scratch_store /tmp/_f.s) spill_ld $(grep -c scratch_load /tmp/_f.s)"
done
rm -f chol_bench-hip-* chol_bench-host-* chol_bench.hip-hip-*
