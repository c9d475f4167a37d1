#!/bin/bash
# Runs on the GPU box (via gpurun): the whole -m gpu suite in one pytest process (as the driver does) with timing, then the default bench.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider --durations=15 ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
grep -E "^\[|max\|err\||gap|passed|failed|Error|error" gpurun_out/pytest_gpu.log | tail -60
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"
cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
