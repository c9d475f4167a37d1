#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_mini_matrix.txt
: > $O
for j in 0 2 3; do python scripts/exp/r06_mini_matrix.py $j >> $O 2>&1; done
grep -v amdgpu.ids $O
