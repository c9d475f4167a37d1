#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in ${VARIANTS:-0 3 1 2 4 5}; do
  for rep in 1 2; do timeout 240 python scripts/exp/r04_hazard_repro.py $v ${STEPS:-30} 2>&1 | grep -E "variant|Error|error" | tail -3; done
done | tee gpurun_out/r04_hazard_variants.log
