#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python scripts/exp/r05_mfma_power.py > gpurun_out/r05_mfma_power.txt 2>&1; echo "exit $?"
cat gpurun_out/r05_mfma_power.txt
