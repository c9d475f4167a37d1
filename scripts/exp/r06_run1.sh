#!/bin/bash
# round 6 call 1: in-process slowdown discriminators
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_inproc.txt
: > $O
P="python scripts/exp/r06_inproc.py"
echo "== standalone cfg5" >> $O; $P cfg5 >> $O 2>&1
echo "== standalone cfg3" >> $O; $P cfg3 >> $O 2>&1
echo "== cfg2,cfg5,cfg3 keep" >> $O; $P cfg2,cfg5,cfg3 >> $O 2>&1
echo "== cfg2,cfg5,cfg3 teardown" >> $O; $P cfg2,cfg5,cfg3 --teardown >> $O 2>&1
echo "== Q8 cfg2,cfg5,cfg3 keep" >> $O; GPU_MAX_HW_QUEUES=8 $P cfg2,cfg5,cfg3 >> $O 2>&1
echo "== Q2 cfg2,cfg5,cfg3 keep" >> $O; GPU_MAX_HW_QUEUES=2 $P cfg2,cfg5,cfg3 >> $O 2>&1
echo "== cfg5,cfg2 keep" >> $O; $P cfg5,cfg2 >> $O 2>&1
echo "== cfg5,cfg5,cfg5 keep" >> $O; $P cfg5,cfg5,cfg5 >> $O 2>&1
echo "== cfg2 extra, cfg5" >> $O; $P cfg2,cfg5 --extra >> $O 2>&1
grep -E "^==|SUMMARY" $O
