#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 300 python scripts/exp/r04_attn.py 2>&1 | grep -v amdgpu.ids | tee $O/r04_attn.log
for rep in 1 2; do for v in 0 1; do
  PSAM_ATTN_VARIANT=$v timeout 300 python bench.py --no-cpu-baseline --sustained-steps 100 > $O/r04_bench8_v${v}_$rep.json 2> $O/r04_bench8_v${v}_$rep.err; echo "bench v$v exit $?"
done; done
for f in $O/r04_bench8_v*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], "sustained", d.get("sustained",{}).get("value"), "stage", d.get("stage_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
