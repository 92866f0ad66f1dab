#!/bin/bash
# launch parameters re-swept on the final build (C4 linear unless said)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6s; mkdir -p $out
{
python3 tools/sweep_env.py r6s SATBA_CHOL_BESIDE_WGS -,24,28,32,36,40,-
python3 tools/sweep_env.py r6s SATBA_BPC -,1,2,3,4
python3 tools/sweep_env.py r6s SATBA_SPLIT -,0,1
python3 tools/sweep_env.py r6s SATBA_LIN_REP -,1,2
python3 tools/sweep_env.py r6s SATBA_CM_CHUNKS -,8,16
python3 tools/sweep_env.py r6s SATBA_BPC -,1,2,3,4 --shape C3
python3 tools/sweep_env.py r6s SATBA_SPLIT -,0,1,2 --shape C3
python3 tools/sweep_env.py r6s SATBA_SPLIT -,2,3 --shape C2
python3 tools/sweep_env.py r6s SATBA_BPC -,1,2,4 --shape C2
} 2>&1 | grep -v amdgpu.ids | tee $out/sweeps.txt
