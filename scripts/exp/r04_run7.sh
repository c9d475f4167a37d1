#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "click_session or variants_through or row_ln_512 or register_epilogue or graph_pipeline" > $O/r04_pytest_gpu_c.log 2>&1; echo "pytest exit $?"
tail -25 $O/r04_pytest_gpu_c.log
timeout 900 python bench.py --workload cfg3 --sustained-steps 50 --no-stage-times > $O/r04_bench_cfg3.json 2> $O/r04_bench_cfg3.err; echo "bench cfg3 exit $?"; tail -3 $O/r04_bench_cfg3.err
python - $O/r04_bench_cfg3.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("parity"), d.get("cpu_baseline"))
PY
