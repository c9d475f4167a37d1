#!/bin/bash
# the tight eager reproducer of round 3 (fc2 with fixed inputs beside another stream's GEMMs) on variant libraries of the pre-fix tree
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
T=scripts/exp/hazard_tree
for v in ${VARIANTS:-0 32}; do
  cp $T/point_sam_amd/csrc/libpointsam_hip_haz$v.so $T/point_sam_amd/csrc/libpointsam_hip.so
  echo "=== variant $v"
  timeout 300 python $T/scripts/exp/r03_race.py 2>&1 | grep -v "amdgpu\|Warning" | tail -${TAILN:-60}
done | tee gpurun_out/r04_hazard_race.log
