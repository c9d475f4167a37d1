#!/bin/bash
# Round 6, VERDICT r05 item 4: is a dense-path kernel's time "refilled" by the other stream?  Experiments library, PSAM_ABLATE_REPEAT launches an idempotent
# kernel twice; the step time it adds, against the kernel's own time x launches per step (profiles/r06/r06_steady_state_*), is its exposed share.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_refill.txt
: > $O
export PSAM_LIB_PATH=$PWD/scripts/exp/libpointsam_experiments.so
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-mfma-probe --no-stage-times --no-gemm-profile --sustained-steps 0 --no-other-workloads"
for w in cfg2 cfg3; do
for rep in 1 2; do
for m in 0 1 2 4; do
  echo "== $w PSAM_ABLATE_REPEAT=$m" >> $O
  PSAM_ABLATE_REPEAT=$m $B --workload $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O
done
done
done
cat $O
