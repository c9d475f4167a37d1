#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/r04_pytest_gpu_a.log 2>&1; echo "pytest exit $?"
tail -15 $O/r04_pytest_gpu_a.log
for m in 0 1; do
  PSAM_GEMM_TR=$m timeout 300 python bench.py --no-cpu-baseline --sustained-steps 100 > $O/r04_bench3_tr${m}.json 2> $O/r04_bench3_tr${m}.err; echo "bench tr$m exit $?"
done
for f in $O/r04_bench3_tr*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], "sustained", d.get("sustained",{}).get("value"), "frac", d["roofline"]["frac"], "stage", d.get("stage_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
