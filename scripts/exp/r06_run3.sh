#!/bin/bash
# round 6 call 3: per-launch penalty or lost concurrency?  shared stream pool as the fix?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_inproc2.txt
: > $O
P="python scripts/exp/r06_inproc.py"
echo "== private streams: cfg5,cfg5 recheck" >> $O; PSAM_PRIVATE_STREAMS=1 $P cfg5,cfg5 --recheck >> $O 2>&1
echo "== shared pool: cfg5,cfg5 recheck" >> $O; $P cfg5,cfg5 --recheck >> $O 2>&1
echo "== shared pool: cfg2,cfg5,cfg3 recheck" >> $O; $P cfg2,cfg5,cfg3 --recheck >> $O 2>&1
echo "== shared pool: cfg2,cfg5,cfg3 teardown" >> $O; $P cfg2,cfg5,cfg3 --teardown >> $O 2>&1
grep -vE "Warning|amdgpu.ids" $O | cut -c1-400
