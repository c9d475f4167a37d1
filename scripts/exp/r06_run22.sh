#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_splitk_fill.txt
: > $O
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-mfma-probe --sustained-steps 0 --no-other-workloads"
for rep in 1 2; do
for pct in 55 30; do
  echo "== cfg5 SPLITK_MAX_FILL_PCT=$pct" >> $O
  PSAM_GEMM_SPLITK_MAX_FILL_PCT=$pct $B --workload cfg5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'avg_launch_ms', d['roofline']['avg_launch_ms'], 'stage', d['stage_ms'].get('vit_blocks'), d['stage_ms'].get('encode_total'), 'click', d.get('ms_per_additional_click'))" >> $O
done
done
cat $O
