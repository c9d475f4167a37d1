#!/bin/bash
# round 6 call 2: which HSA queue does each stream's work go through (first vs second harness of one process)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_queue_map.txt
: > $O
rocprofv3 --kernel-trace --output-format csv -d /tmp/qm -o qm -- python scripts/exp/r06_inproc.py cfg5,cfg5 --steps 6 --repeat 1 >> $O 2>&1
f=$(find /tmp/qm -name '*kernel_trace.csv' | head -1)
python scripts/exp/r06_queue_map.py $f >> $O 2>&1
tail -40 $O
