#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; mkdir -p gpurun_out/pf16; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf16 -o p -- python $R/bench.py --no-cpu-baseline --no-gemm-profile --no-pipeline --steps 5 --precision ${1:-f16x3} > /dev/null 2>&1)
f=$(find gpurun_out/pf16 -name '*kernel_stats.csv' | head -1); head -24 $f | cut -c1-170
