#!/bin/bash
# HBM traffic counters of the dominant kernel (separate --pmc passes, as the microarch guide prescribes). GPU box.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02}
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch_$TAG -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-profile --no-graphs --no-stage-times --sustained-steps 0 > /dev/null 2>$GRAFT_REPO_ROOT/gpurun_out/pmc_fetch_$TAG.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write_$TAG -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-profile --no-graphs --no-stage-times --sustained-steps 0 > /dev/null 2>$GRAFT_REPO_ROOT/gpurun_out/pmc_write_$TAG.err
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/pmc_*_$TAG/*kernel_trace.csv
ls -la gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG
