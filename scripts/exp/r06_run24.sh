#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_pipe_keysplit.txt
: > $O
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-mfma-probe --no-stage-times --no-gemm-profile --sustained-steps 0 --no-other-workloads"
for rep in 1 2; do
for ks in 1 2 4; do
  echo "== cfg5 PIPE_KEYSPLIT=$ks" >> $O
  PSAM_PIPE_KEYSPLIT=$ks $B --workload cfg5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O
done
done
cat $O
