#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "skinny or mlp3 or attention_small or scale_pack or g8_packing or linear_ln256 or rows_multi" > gpurun_out/pytest_new.log 2>&1; echo "pytest kernels exit $?"; tail -12 gpurun_out/pytest_new.log
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_variants.py -m gpu -q -s --tb=short -p no:cacheprovider -x > gpurun_out/pytest_e2e.log 2>&1; echo "pytest e2e exit $?"; grep -E "^\[two-way|^\[coarse|^\[cfg5|passed|failed|Error" gpurun_out/pytest_e2e.log | tail -30
for mode in 2 1 0; do
  cd /tmp
  PSAM_TWOWAY_FAST=$mode PSAM_ROWS_MULTI=$([ $mode = 2 ] && echo 1 || echo 0) timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/click$mode -o t -- python $R/scripts/exp/r05_click_trace.py run > $R/gpurun_out/click.log 2>&1; echo "trace exit $?"
  cd $R
  python scripts/exp/r05_click_trace.py report "gpurun_out/click$mode/**/t_kernel_trace.csv" > gpurun_out/r05_click_kernels_mode$mode.txt; tail -2 gpurun_out/r05_click_kernels_mode$mode.txt
  find gpurun_out/click$mode -name "*.csv" -size +20M -delete
done
cat gpurun_out/r05_click_kernels_mode2.txt
