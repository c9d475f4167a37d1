#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_stream_sets.txt
: > $O
python scripts/exp/r06_stream_sets.py cfg5 >> $O 2>&1
grep -vE "Warning|amdgpu.ids" $O | cut -c1-300
# ROCclr's own log: stream (software queue) -> hardware queue
AMD_LOG_LEVEL=4 python scripts/exp/r06_stream_alias.py 10 > /tmp/clr.log 2>&1
grep -iE "hardware queue|hw queue|acquire|HWq|created.*queue|queue.*priority" /tmp/clr.log | sed -E 's/^:[0-9]+:[^:]*:[ 0-9]*: *[0-9]* *us:? *//' | cut -c1-160 | sort | uniq -c | sort -rn | head -60 > gpurun_out/r06_clr_queue_log.txt
head -c 3000 /tmp/clr.log > gpurun_out/r06_clr_head.txt
wc -l /tmp/clr.log
cat gpurun_out/r06_clr_queue_log.txt | head -60
