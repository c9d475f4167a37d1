#!/bin/bash
# the experiments variant of the library (scripts/exp/build_experiments_lib.sh): the tests that skip on the production build
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
export PSAM_LIB_PATH=$PWD/scripts/exp/libpointsam_experiments.so
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -rs -k "continuous or row_ln or fork or token or gemm_f16x3 or twoway" > gpurun_out/pytest_experiments.log 2>&1; echo "pytest exit $?"
tail -15 gpurun_out/pytest_experiments.log
