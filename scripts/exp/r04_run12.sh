#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
bash scripts/gpu_profile.sh r04 5 > gpurun_out/gpu_profile_r04.log 2>&1; echo "profile exit $?"
bash scripts/gpu_pmc.sh r04 > gpurun_out/gpu_pmc_r04.log 2>&1; echo "pmc exit $?"
python scripts/pmc_to_traffic.py r04 gpurun_out/r04_traffic.json > gpurun_out/r04_traffic.log 2>&1; echo "traffic exit $?"
bash scripts/exp/gemm_clock.sh > gpurun_out/gemm_clock_r04.log 2>&1; echo "clock exit $?"; cp gpurun_out/gemm_clock.txt gpurun_out/r04_gemm_counters.txt
f=$(find gpurun_out/prof_r04 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r04_rocprofv3_kernel_stats.csv; head -12 "$f" | cut -c1-160
cat gpurun_out/prof_r04.bench.json | cut -c1-400
cat gpurun_out/r04_gemm_counters.txt
head -c 1500 gpurun_out/r04_traffic.json
