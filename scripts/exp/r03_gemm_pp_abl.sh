#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
ABL_BASE=200 PSAM_HIP_LIB=$PWD/scripts/exp/libpointsam_abl.so timeout 600 python scripts/exp/gemm_abl.py > gpurun_out/r03_gemm_pp_abl.log 2>&1; echo "abl exit $?" >> gpurun_out/r03_gemm_pp_abl.log
head -45 gpurun_out/r03_gemm_pp_abl.log
