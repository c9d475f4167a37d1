#!/bin/bash
# the new interp3 stress test on the build BEFORE the fix (must fail) and on the shipped library (must pass); then the GPU suite
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== before the fix"; PSAM_LIB_PATH=$PWD/scripts/exp/libpointsam_before_interp3_fix.so timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=line -p no:cacheprovider -k "interp3_bitwise_stable" 2>&1 | tail -5
echo "== shipped"; timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=line -p no:cacheprovider -k "interp3_bitwise_stable or interp" 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $O/r04_pytest_gpu_full2.log 2>&1; echo "pytest exit $?"; tail -4 $O/r04_pytest_gpu_full2.log
