#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider -k "c_eva_gelu or cfg5 or attention_packed or c_eva_block" > $O/r04_pytest_gpu_d.log 2>&1; echo "pytest exit $?"
grep -E "^\[|passed|failed|Error|error" $O/r04_pytest_gpu_d.log | tail -20
timeout 900 python bench.py --workload cfg5 --sustained-steps 100 --no-cpu-baseline > $O/r04_bench_cfg5_c.json 2> $O/r04_bench_cfg5_c.err; echo "bench cfg5 exit $?"; tail -2 $O/r04_bench_cfg5_c.err
python - $O/r04_bench_cfg5_c.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("stage_ms"), d.get("roofline",{}).get("frac"))
PY
