#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
T=scripts/exp/hazard_tree
for v in ${VARIANTS:-0 32}; do
  cp $T/point_sam_amd/csrc/libpointsam_hip_haz$v.so $T/point_sam_amd/csrc/libpointsam_hip.so
  echo -n "variant $v: "
  timeout 200 python $T/scripts/exp/race_short.py 2>&1 | grep -v "amdgpu\|Warning" | tail -3
done | tee gpurun_out/r04_hazard_race_short.log
