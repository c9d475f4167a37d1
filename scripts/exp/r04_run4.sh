#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "${PYTEST_K:-patch or golden or c_eva or against_oracle or register_epilogue or properties or pipelines}" > $O/r04_pytest_gpu_b.log 2>&1; echo "pytest exit $?"
tail -25 $O/r04_pytest_gpu_b.log
for rep in 1 2; do for m in 0 1; do
  PSAM_GEMM_TR=$m timeout 300 python bench.py --no-cpu-baseline --sustained-steps 100 > $O/r04_bench4_tr${m}_$rep.json 2> $O/r04_bench4_tr${m}_$rep.err; echo "bench tr$m exit $?"
done; done
for f in $O/r04_bench4_tr*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], "sustained", d.get("sustained",{}).get("value"), "frac", d["roofline"]["frac"], "stage", d.get("stage_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
