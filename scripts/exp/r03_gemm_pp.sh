#!/bin/bash
# GPU box: ping-pong GEMM (gemm_f16x3pp.hip) against the lock-step ring kernel: GEMM kernel tests, correctness vs fp64, per-shape timing, layer
# loops on 1/2/3 streams, then the ablation split of the ping-pong kernel.  Logs -> gpurun_out/r03_gemm_pp*.log
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
CFGS=${1:-21,50,51,59,60,61,55,57,58}
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "gemm or linear or mlp" > gpurun_out/r03_gemm_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/r03_gemm_tests.log
tail -5 gpurun_out/r03_gemm_tests.log
fi
timeout 900 python scripts/gemm_p_bench.py $CFGS > gpurun_out/r03_gemm_pp.log 2>&1; echo "bench exit $?" >> gpurun_out/r03_gemm_pp.log
grep -v "^check.*ok$" gpurun_out/r03_gemm_pp.log | tail -60
if [ -f scripts/exp/libpointsam_abl.so ]; then
  ABL_BASE=200 PSAM_HIP_LIB=$PWD/scripts/exp/libpointsam_abl.so timeout 600 python scripts/exp/gemm_abl.py > gpurun_out/r03_gemm_pp_abl.log 2>&1; echo "abl exit $?" >> gpurun_out/r03_gemm_pp_abl.log
  head -32 gpurun_out/r03_gemm_pp_abl.log
fi
