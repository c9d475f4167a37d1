#!/bin/bash
# GPU box: rocprofv3 kernel stats of one BASELINE config's un-pipelined passes (scripts/stage_times.py): STAGE_CFGS=cfg5 bash scripts/exp/r03_prof_cfg.sh
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; C=${STAGE_CFGS:-cfg5}
cd /tmp
STAGE_CFGS=$C rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$C -o st -- python $R/scripts/stage_times.py > $R/gpurun_out/prof_$C.log 2>&1
cd $R
f=$(find gpurun_out/prof_$C -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" | cut -c1-230
find gpurun_out/prof_$C -name "*kernel_trace.csv" -delete
