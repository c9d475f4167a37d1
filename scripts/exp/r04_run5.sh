#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 400 python scripts/exp/r04_q.py > $O/r04_q.log 2>&1; echo "q exit $?"
timeout 500 python scripts/gemm_p_bench.py 21,80,81,82,83,84 > $O/r04_gemm_p_q.log 2>&1; echo "gemm_p exit $?"
for q in 0 80 84; do
  PSAM_GEMM_Q=$q timeout 300 python bench.py --no-cpu-baseline --sustained-steps 100 > $O/r04_bench5_q${q}.json 2> $O/r04_bench5_q${q}.err; echo "bench q$q exit $?"
done
grep -v amdgpu.ids $O/r04_q.log | tail -22
grep -v "^check" $O/r04_gemm_p_q.log | tail -20
for f in $O/r04_bench5_q*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], "sustained", d.get("sustained",{}).get("value"), "frac", d["roofline"]["frac"], "stage", d.get("stage_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
