#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_stream_alias.txt
: > $O
python scripts/exp/r06_stream_alias.py >> $O 2>&1
GPU_MAX_HW_QUEUES=8 python scripts/exp/r06_stream_alias.py >> $O 2>&1
GPU_MAX_HW_QUEUES=2 python scripts/exp/r06_stream_alias.py 6 >> $O 2>&1
cat $O
