#!/bin/bash
# Round 5, GPU call 6: cycles and sustained clock (GRBM_GUI_ACTIVE / duration) + matrix-pipe busy of cfg 21 against the continuous kernel (cfg 94 / 95)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
GEMM_CFGS=21,94,95 bash scripts/exp/gemm_clock.sh > gpurun_out/r05_gemm_clock.log 2>&1
cp gpurun_out/gemm_clock.txt gpurun_out/r05_gemm_clock.txt 2>/dev/null
cat gpurun_out/r05_gemm_clock.txt; tail -5 gpurun_out/r05_gemm_clock.log
