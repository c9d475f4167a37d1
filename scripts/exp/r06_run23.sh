#!/bin/bash
cd $GRAFT_REPO_ROOT
python - > gpurun_out/r06_cfg29_check.txt 2>&1 <<'PY'
import sys; sys.argv=['x','9,29']
sys.path.insert(0,'scripts')
import gemm_p_bench as g
print('bad', g.correctness())
PY
grep -v amdgpu.ids gpurun_out/r06_cfg29_check.txt | tail -14
CFGS=9,29,31,30 python scripts/exp/r06_small_m.py > gpurun_out/r06_small_m2.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06_small_m2.txt
