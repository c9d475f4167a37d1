#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_queue_truth.txt
: > $O
for j in 0 1 2 3 5 6; do
  echo "== junk $j" >> $O
  AMD_LOG_LEVEL=4 python scripts/exp/r06_queue_truth.py $j > /tmp/q.log 2>&1
  grep -E "^POOL|^TAG" /tmp/q.log >> $O
  grep -E "Created SWq" /tmp/q.log | sed -E 's/.*Created/Created/' >> $O
  grep -E "grid=\[(256256|256768|257280|257792)," /tmp/q.log | sed -E 's/.*(SWq=[^,]*, HWq=[^,]*, id=[0-9]+).*(grid=\[[0-9]+).*/\1 \2/' >> $O
done
cat $O
