#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/r04_pytest_gpu_full.log 2>&1; echo "pytest exit $?"
tail -8 $O/r04_pytest_gpu_full.log
for rep in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --sustained-steps 100 > $O/r04_bench13_$rep.json 2> /dev/null; echo "bench exit $?"
python - $O/r04_bench13_$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], "sustained", d["sustained"]["value"], "frac", d["roofline"]["frac"], d.get("stage_ms"))
PY
done
