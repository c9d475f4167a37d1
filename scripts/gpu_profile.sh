#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command (GPU box).  Output: gpurun_out/prof_*/ ; summary copied by hand to profiles/.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02}
STEPS=${2:-3}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-gemm-profile --no-stage-times --sustained-steps 0 ${BENCH_EXTRA:-} > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.err
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_$TAG -name "*stats*" | head; 
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f"
# keep only the small summaries (the raw trace can be large)
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +30M -delete
cat gpurun_out/prof_$TAG.bench.json
