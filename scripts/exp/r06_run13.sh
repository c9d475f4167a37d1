#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_inproc6.txt
: > $O
P="python scripts/exp/r06_inproc.py"
for j in 0 1 2 3 4 5 6 7 9; do
echo "== probed pool, $j+2 application streams first: cfg5" >> $O; $P cfg5 --junk $j --repeat 1 >> $O 2>&1
done
echo "== probed pool: cfg2,cfg5,cfg3" >> $O; $P cfg2,cfg5,cfg3 >> $O 2>&1
grep -E "^==|SUMMARY|POOL|Error|error" $O | cut -c1-300
