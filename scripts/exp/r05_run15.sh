#!/bin/bash
# experiments library: the fork test again; then the mask encoder's 128x512 row-LayerNorm tile at batch 1 (one replayed click, PSAM_GEMM_ROWLN512 = 0 / 1)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
export PSAM_LIB_PATH=$PWD/scripts/exp/libpointsam_experiments.so
timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -k "fork" 2>&1 | tail -3
for ln in 0 1; do
  cd /tmp
  PSAM_GEMM_ROWLN512=$ln timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/clickln$ln -o t -- python $R/scripts/exp/r05_click_trace.py run > $R/gpurun_out/click.log 2>&1; echo "trace exit $?"
  cd $R
  python scripts/exp/r05_click_trace.py report "gpurun_out/clickln$ln/**/t_kernel_trace.csv" > gpurun_out/r05_click_kernels_rowln$ln.txt; sed -n 3,14p gpurun_out/r05_click_kernels_rowln$ln.txt | cut -c1-110; tail -1 gpurun_out/r05_click_kernels_rowln$ln.txt
  rm -rf gpurun_out/clickln$ln
done
