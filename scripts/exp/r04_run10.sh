#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "e2e or variants" > $O/r04_pytest_gpu_e.log 2>&1; echo "pytest exit $?"
tail -12 $O/r04_pytest_gpu_e.log
for w in cfg2 cfg5; do
timeout 600 python bench.py --workload $w --no-cpu-baseline --sustained-steps 100 > $O/r04_bench10_$w.json 2> $O/r04_bench10_$w.err; echo "bench $w exit $?"
python - $O/r04_bench10_$w.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("stage_ms"))
PY
done
