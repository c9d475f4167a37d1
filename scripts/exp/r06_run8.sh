#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_inproc4.txt
: > $O
P="python scripts/exp/r06_inproc.py"
echo "== probed pool: cfg2,cfg5,cfg3" >> $O; $P cfg2,cfg5,cfg3 >> $O 2>&1
echo "== probed pool, 5+2 application streams first: cfg5,cfg2" >> $O; $P cfg5,cfg2 --junk 5 >> $O 2>&1
echo "== probed pool, 2+2 application streams first: cfg5" >> $O; $P cfg5 --junk 2 >> $O 2>&1
echo "== private, 2+2 application streams first: cfg5" >> $O; PSAM_PRIVATE_STREAMS=1 $P cfg5 --junk 2 >> $O 2>&1
echo "== normal-priority tokenizer stream: cfg2,cfg5,cfg3" >> $O; PSAM_TOK_PRIORITY=0 $P cfg2,cfg5,cfg3 >> $O 2>&1
echo "== probed pool again: cfg2,cfg5,cfg3" >> $O; $P cfg2,cfg5,cfg3 >> $O 2>&1
grep -E "^==|SUMMARY|POOL|Error|error" $O | cut -c1-300
