#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_host_gate.txt
: > $O
P="python scripts/exp/r06_inproc.py"
for i in 1 2; do
echo "== gpu-side waits: cfg2,cfg5,cfg3" >> $O; $P cfg2,cfg5,cfg3 --steps 40 >> $O 2>&1
echo "== host-gated next(): cfg2,cfg5,cfg3" >> $O; PSAM_HOST_GATE=1 $P cfg2,cfg5,cfg3 --steps 40 >> $O 2>&1
done
grep -E "^==|SUMMARY|Error|error" $O | cut -c1-300
