#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_inproc3.txt
: > $O
P="python scripts/exp/r06_inproc.py"
echo "== private streams: cfg5,cfg5,cfg5" >> $O; PSAM_PRIVATE_STREAMS=1 $P cfg5,cfg5,cfg5 >> $O 2>&1
echo "== private streams Q8: cfg5,cfg5,cfg5" >> $O; GPU_MAX_HW_QUEUES=8 PSAM_PRIVATE_STREAMS=1 $P cfg5,cfg5,cfg5 >> $O 2>&1
echo "== private streams Q2: cfg5,cfg5" >> $O; GPU_MAX_HW_QUEUES=2 PSAM_PRIVATE_STREAMS=1 $P cfg5,cfg5 >> $O 2>&1
echo "== probed shared pool: cfg2,cfg5,cfg3" >> $O; $P cfg2,cfg5,cfg3 >> $O 2>&1
echo "== probed shared pool Q2: cfg2,cfg5" >> $O; GPU_MAX_HW_QUEUES=2 $P cfg2,cfg5 >> $O 2>&1
grep -vE "Warning|amdgpu.ids" $O | cut -c1-330
