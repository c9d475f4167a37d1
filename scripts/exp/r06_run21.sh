#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/exp/r06_small_m.py > gpurun_out/r06_small_m.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06_small_m.txt
