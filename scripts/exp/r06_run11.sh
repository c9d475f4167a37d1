#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_inproc5.txt
: > $O
P="python scripts/exp/r06_inproc.py"
echo "== probed pool: cfg2,cfg5,cfg3" >> $O; $P cfg2,cfg5,cfg3 >> $O 2>&1
for j in 1 2 3 5 6; do
echo "== probed pool, $j+2 application streams first: cfg5" >> $O; $P cfg5 --junk $j >> $O 2>&1
done
echo "== private, 2+2 application streams first: cfg5" >> $O; PSAM_PRIVATE_STREAMS=1 $P cfg5 --junk 2 >> $O 2>&1
echo "== probed pool Q8, 2+2 application streams first: cfg5" >> $O; GPU_MAX_HW_QUEUES=8 $P cfg5 --junk 2 >> $O 2>&1
grep -E "^==|SUMMARY|POOL|Error|error" $O | cut -c1-300
