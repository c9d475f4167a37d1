#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/click -o t -- python $R/scripts/exp/r05_click_trace.py run > $R/gpurun_out/click.log 2>&1; echo "trace exit $?"
cd $R
python scripts/exp/r05_click_trace.py report "gpurun_out/click/**/t_kernel_trace.csv" > gpurun_out/r05_click_kernels.txt; tail -5 gpurun_out/r05_click_kernels.txt
find gpurun_out/click -name "*.csv" -size +20M -delete
ATTN_L=512 timeout 200 python scripts/attn_bench.py 2>&1 | tail -8
